#!/bin/bash
# HBM traffic of the headline kernels from PMC counters, in separate passes as the guide prescribes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass).  Run on the GPU box from the repo root.
set -o pipefail
out=$PWD/gpurun_out/prof_traffic
mkdir -p $out
export TMPDIR=/tmp
for kind in logpost grad; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $out/${kind}_$c -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --kind $kind > $out/${kind}_$c.log 2>&1 || echo "pass $kind $c failed"
  done
done
python3 tools/prof_traffic.py $out
