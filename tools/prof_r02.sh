#!/bin/bash
# Round-2 profile set (run on the GPU box from the repo root): rocprofv3 --kernel-trace --stats + three PMC passes for
#   fwd  : the headline kernel (bench.py default: sliced int8-product forward)
#   fwd_dp: the float64-MFMA forward (bench.py --path fused_dp)
#   grad : k_fused_bwd_f64 (bench.py --kind grad)
#   cfg4 : k_gemm64 FWD / DA / DW at the cfg4 layer shape (tools/prof_cfg.sh)
# and the HBM traffic passes.  Every profiled program goes directly after `--`.
set -o pipefail
bash tools/prof.sh r02_fwd > gpurun_out/prof_r02_fwd.txt 2>&1; echo "fwd done"
bash tools/prof.sh r02_fwd_dp --path fused_dp > gpurun_out/prof_r02_fwd_dp.txt 2>&1; echo "fwd_dp done"
bash tools/prof.sh r02_grad --kind grad > gpurun_out/prof_r02_grad.txt 2>&1; echo "grad done"
bash tools/prof_cfg.sh > gpurun_out/prof_r02_cfg4.txt 2>&1; echo "cfg4 done"
bash tools/prof_traffic.sh > gpurun_out/prof_r02_traffic.txt 2>&1; echo "traffic done"
python3 bench.py --steps 200 --warmup 20 > gpurun_out/bench_r02_default.json 2> gpurun_out/bench_r02_default.err; echo "bench done"
