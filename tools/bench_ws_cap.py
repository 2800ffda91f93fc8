import sys, os, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from quinn_amd.ops import MLPArch, BatchedMLP
dims, N = (1, 256, 256, 256, 256, 1), 32768
arch = MLPArch(dims, "tanh"); rs = np.random.RandomState(0)
x = rs.rand(N, 1) * 6 - 3; y = np.sin(x)
for cap in (16, 48, 96):
    op = BatchedMLP(arch, x, y, max_workspace_bytes=cap << 30)
    for B in (64, 256):
        W = op.weights(0.1 * rs.randn(B, arch.nparams))
        op.sse_grad(W); torch.cuda.synchronize()
        t0 = time.perf_counter(); op.sse_grad(W); op.sse_grad(W); torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 2
        print("cap", cap, "B", B, "chunk", op._chunk(B, N, True), "grad TFLOP/s", B * arch.flops_fwdbwd(N) / el / 1e12, flush=True)
    del op; torch.cuda.empty_cache()
