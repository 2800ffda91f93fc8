#!/bin/bash
# kernel-trace only of the layer-wise path at the cfg4 shape (run on the GPU box from the repo root)
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_cfg
mkdir -p $out
cat > /tmp/run_cfg.py <<'PY'
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from quinn_amd.ops import MLPArch, BatchedMLP
dims, N, B = (1, 256, 256, 256, 256, 1), 16384, 64
arch = MLPArch(dims, "tanh"); rs = np.random.RandomState(0)
x = rs.rand(N, 1) * 6 - 3; y = np.sin(x)
op = BatchedMLP(arch, x, y); W = op.weights(0.1 * rs.randn(B, arch.nparams))
for _ in range(3): op.sse(W)
for _ in range(3): op.sse_grad(W)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 /tmp/run_cfg.py > $out/trace.log 2>&1
python3 tools/prof_summary.py $out | head -30
