#!/bin/bash
# usage: tools/prof.sh <tag> [bench args...]   (run on the GPU box from the repo root)
set -o pipefail
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras "$@" > $out/trace.log 2>&1
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE GRBM_COUNT"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $out/pmc_$n -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras "$@" > $out/pmc_$n.log 2>&1 || echo "pmc $n failed"
done
python3 tools/prof_summary.py $out
