#!/bin/bash
# round-2 first GPU call: i8-pipe microbenchmark, bench (1 rank; 2-rank rehearsal weak + strong), GPU test suite
set -o pipefail
mkdir -p gpurun_out
./tools/ubench_i8 > gpurun_out/ubench_i8.txt 2>&1 || echo "ubench_i8 failed"
python3 bench.py --steps 200 --warmup 20 > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err || echo "bench n1 failed"
python3 bench.py --gpus 2 --steps 100 --warmup 10 > gpurun_out/bench_n2_weak.json 2> gpurun_out/bench_n2_weak.err || echo "bench n2 weak failed"
python3 bench.py --gpus 2 --scaling strong --steps 100 --warmup 10 > gpurun_out/bench_n2_strong.json 2> gpurun_out/bench_n2_strong.err || echo "bench n2 strong failed"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/pytest_gpu.txt
