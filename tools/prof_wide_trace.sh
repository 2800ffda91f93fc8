#!/bin/bash
# Kernel trace of one gradient + one forward evaluation at the cfg4 (4x256, N = 16384, 64 members) and cfg3 (3x128, N = 8192,
# 128 samples) shapes after a 0.3 s warm-up each (settled clock): per-kernel median durations.  GPU box, repo root.
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_wide_trace
rm -rf $out; mkdir -p $out
cat > /tmp/run_wide_trace.py <<'PY'
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from quinn_amd.ops import MLPArch, BatchedMLP
for dims, N, B in (((1, 256, 256, 256, 256, 1), 16384, 64), ((2, 128, 128, 128, 1), 8192, 128)):
    arch = MLPArch(dims, "tanh"); rs = np.random.RandomState(0)
    x = rs.rand(N, dims[0]) * 6 - 3; y = np.sin(x).sum(axis=1, keepdims=True)
    op = BatchedMLP(arch, x, y); W = op.weights(0.1 * rs.randn(B, arch.nparams))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        op.sse_grad(W); torch.cuda.synchronize()
    for _ in range(10): op.sse_grad(W)
    for _ in range(10): op.sse(W)
    torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 /tmp/run_wide_trace.py > $out/trace.log 2>&1
python3 tools/prof_summary.py $out | grep -v "^== pmc" | head -60
