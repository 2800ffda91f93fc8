#!/bin/bash
# usage: tools/prof_any.sh <tag> <kernel-name-filter (regex)> <python script> [args...]   (run on the GPU box from the repo root)
# One kernel-trace pass and the PMC passes of MI355X_MICROARCH.md (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one;
# never combined with a trace domain), then tools/prof_any_summary.py -> gpurun_out/prof_<tag>/summary.txt + summary.json.
# The program after `--` is python3 itself (no env / bash -c hop under rocprofv3).
set -o pipefail
tag=$1; pat=$2; shift 2
out=$PWD/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 "$@" > $out/trace.log 2>&1 || echo "trace pass failed"
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE GRBM_COUNT"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $out/pmc_$n -- python3 "$@" > $out/pmc_$n.log 2>&1 || echo "pmc $n failed"
  echo "pass $n done"
done
python3 tools/prof_any_summary.py $out "$pat" | tee $out/summary.txt
