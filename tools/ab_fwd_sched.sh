#!/bin/bash
# In-call A/B of compiler scheduling strategies for qn_fused_i8.hip (variants built by tools/ab_build3.py <name> qn_fused_i8.hip -mllvm ...):
# headline bench line per variant, two alternating rounds.  Run on the GPU box from the repo root.
for r in 1 2; do
for v in base igrp maxilp iter; do
  if [ $v == base ]; then unset QUINN_AMD_LIB; else export QUINN_AMD_LIB=$PWD/quinn_amd/lib/libquinn_amd_$v.so; fi
  python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(sys.argv[1], round(r['value']), r['roofline']['kernel_ms'])" $v
done; done
