#!/usr/bin/env python3
"""Forward / gradient rates at the cfg3 and cfg4 shapes in one JSON line (for tools/ab_run.sh)."""
import sys, os, time, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quinn_amd.ops import MLPArch, BatchedMLP
def timeit(fn, n):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
out = {}
rs = np.random.RandomState(0)
for name, dims, N, B in (("cfg3", (2, 128, 128, 128, 1), 8192, 128), ("cfg4", (1, 256, 256, 256, 256, 1), 16384, 64)):
    arch = MLPArch(dims, "tanh")
    x = rs.rand(N, dims[0]) * 6 - 3; y = np.sin(x).sum(axis=1, keepdims=True)
    op = BatchedMLP(arch, x, y)
    W = op.weights(0.1 * rs.randn(B, arch.nparams))
    out[name + "_fwd_tflops"] = round(B * arch.flops_fwd(N) / timeit(lambda: op.sse(W), 8) / 1e12, 2)
    out[name + "_grad_tflops"] = round(B * arch.flops_fwdbwd(N) / timeit(lambda: op.sse_grad(W), 5) / 1e12, 2)
print(json.dumps(out))
