#!/bin/bash
# A/B of library builds for both evaluation kinds (log-posterior and gradient) and optionally the float64-MFMA forward,
# inside ONE call: tools/ab_libs_kinds.sh <rounds> <name|base> ...
rounds=$1; shift
for i in $(seq $rounds); do
  for v in "$@"; do
    if [ $v == base ]; then unset QUINN_AMD_LIB; else export QUINN_AMD_LIB=$PWD/quinn_amd/lib/libquinn_amd_$v.so; fi
    for k in "--kind logpost" "--kind grad" "--kind logpost --path fused_dp"; do
      python3 bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline $k 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$v', '$k', round(d['value']), 'kernel_ms', round(r['kernel_ms'], 5), 'min', round(r['kernel_ms_min'], 5), 'frac', round(r['frac'], 4), 'traffic', r.get('traffic'), flush=True)"
    done
  done
done
