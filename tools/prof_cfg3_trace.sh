#!/bin/bash
# kernel trace of one gradient + one forward evaluation at the cfg3 shape (128 samples, 3x128, N=8192, d=2); GPU box, repo root
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_cfg3
mkdir -p $out
cat > /tmp/run_cfg3.py <<'PY'
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from quinn_amd.ops import MLPArch, BatchedMLP
dims, N, B = (2, 128, 128, 128, 1), 8192, 128
arch = MLPArch(dims, "tanh"); rs = np.random.RandomState(0)
x = rs.rand(N, 2) * 6 - 3; y = np.sin(x).sum(axis=1, keepdims=True)
op = BatchedMLP(arch, x, y); W = op.weights(0.1 * rs.randn(B, arch.nparams))
for _ in range(4): op.sse_grad(W)
for _ in range(4): op.sse(W)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 /tmp/run_cfg3.py > $out/trace.log 2>&1
python3 tools/prof_summary.py $out | head -40
