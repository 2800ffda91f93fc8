#!/bin/bash
# Round-4 profile set (run on the GPU box from the repo root): rocprofv3 --kernel-trace --stats + three PMC passes for
#   fwd  : the headline kernel k_fused_fwd_i8 with the in-kernel SSE sum (bench.py default)
#   grad : k_fused_bwd_i8 + the float64 second pass + k_grad_reduce (bench.py --kind grad)
# the HBM traffic passes, the busy fractions bench.py quotes (gpurun_out/pmc_busy.json -> profiles/pmc_busy.json) and the
# dispatch-order trace that shows the clock ramp of a fresh process.  Every profiled program goes directly after `--`.
set -o pipefail
bash tools/prof.sh r04_fwd > gpurun_out/prof_r04_fwd.txt 2>&1; echo "fwd done"
bash tools/prof.sh r04_grad --kind grad > gpurun_out/prof_r04_grad.txt 2>&1; echo "grad done"
bash tools/prof_traffic.sh > gpurun_out/prof_r04_traffic.txt 2>&1; echo "traffic done"
python3 tools/pmc_busy.py gpurun_out/prof_r04_fwd k_fused_fwd_i8 k_fused_fwd_i8 r04_fused_fwd_i8_rocprofv3.txt
python3 tools/pmc_busy.py gpurun_out/prof_r04_grad k_fused_bwd_i8 k_fused_bwd_i8 r04_fused_bwd_i8_rocprofv3.txt
python3 tools/trace_order.py gpurun_out/prof_r04_fwd/trace k_fused_fwd_i8 > gpurun_out/r04_clock_ramp_kernel_trace.txt
python3 bench.py --steps 200 --warmup 20 > gpurun_out/bench_r04_default.json 2> gpurun_out/bench_r04_default.err; echo "bench done"
python3 bench.py --steps 200 --warmup 20 --kind grad --no-extras --no-cpu-baseline > gpurun_out/bench_r04_grad.json 2> gpurun_out/bench_r04_grad.err; echo "bench grad done"
