#!/bin/bash
# A/B of library builds on the operator throughput at the BASELINE config shapes: tools/ab_cfg.sh <name|base> ...
for v in "$@"; do
  if [ $v == base ]; then unset QUINN_AMD_LIB; else export QUINN_AMD_LIB=$PWD/quinn_amd/lib/libquinn_amd_$v.so; fi
  echo "== $v"; python3 tools/bench_configs.py 2>/dev/null | grep -v "cfg1\|cfg2" | python3 -c "
import sys, json
for ln in sys.stdin:
    name, js = ln.split(' {', 1); d = json.loads('{' + js)
    print('  ', name, 'fwd %.1f TF  grad %.1f TF' % (d['fwd_tflops'], d['grad_tflops']), flush=True)"
done
