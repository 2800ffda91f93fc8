import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from quinn_amd.ops import MLPArch, BatchedMLP
for dims, N, B in (((1, 256, 256, 256, 256, 1), 16384, 32), ((2, 128, 128, 128, 1), 8192, 128)):
    arch = MLPArch(dims, "tanh"); rs = np.random.RandomState(0)
    x = rs.rand(N, dims[0]) * 6 - 3; y = np.sin(x).sum(axis=1, keepdims=True)
    op = BatchedMLP(arch, x, y); W = op.weights(0.1 * rs.randn(B, arch.nparams))
    op.sse(W); torch.cuda.synchronize()
    op.sse(W); torch.cuda.synchronize()
    op.sse_grad(W); torch.cuda.synchronize()
