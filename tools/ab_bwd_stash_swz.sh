#!/bin/bash
# In-call A/B of the extra stash swizzle bit of k_fused_bwd_i8 (QN_BWD_STASH_SWZ, csrc/qn_fused_bwd_i8.hip): the cfg2 gradient step
# with the default library and with the flagged build, alternating, two rounds.  Build the flagged library first (here, CPU):
#   QN_HIPCC_FLAGS=-DQN_BWD_STASH_SWZ=1 python -c "import quinn_amd._lib as l; print(l.build())"     -> quinn_amd/lib/libquinn_amd_<key>.so
# then on the GPU box:  bash tools/ab_bwd_stash_swz.sh quinn_amd/lib/libquinn_amd_<key>.so
set -o pipefail
OTHER=${1:?path of the flagged library}
for i in 1 2; do
  for lib in quinn_amd/lib/libquinn_amd.so "$OTHER"; do
    QUINN_AMD_LIB=$PWD/$lib python3 bench.py --steps 300 --warmup 20 --kind grad --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json; r = json.loads(sys.stdin.read()); print('$lib', round(r['value']), 'evals/s  step', r['ms_per_step'], 'ms  kernel', r['roofline']['kernel_ms'], 'ms  frac', round(r['roofline']['frac'], 4))" || exit 1
  done
done
