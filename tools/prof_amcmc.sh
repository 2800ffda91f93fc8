#!/bin/bash
# kernel-trace of the device AMCMC engine at cfg2 (run on the GPU box from the repo root)
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_amcmc
mkdir -p $out
NMCMC=${NMCMC:-3000} USE_GRAPH=${USE_GRAPH:-0} rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/bench_amcmc_device.py > $out/trace.log 2>&1
tail -1 $out/trace.log
python3 tools/prof_summary.py $out | head -24
