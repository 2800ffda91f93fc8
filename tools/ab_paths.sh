#!/bin/bash
# A/B of bench.py kernel families inside ONE call (boxes differ by ~10 %): tools/ab_paths.sh [rounds] [path ...]
rounds=${1:-3}; shift
paths=${@:-auto fused_dp}
for i in $(seq $rounds); do
  for p in $paths; do
    python3 bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline --path $p 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$p', round(d['value']), 'kernel_ms', round(r['kernel_ms'], 5), 'min', round(r['kernel_ms_min'], 5), 'frac', round(r['frac'], 4), flush=True)"
  done
done
