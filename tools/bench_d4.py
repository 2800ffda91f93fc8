#!/usr/bin/env python3
"""Forward rate of a 4-input 3x64 network (the DP = 4 instance of the fused forward kernel) in one JSON line."""
import sys, os, time, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quinn_amd.ops import MLPArch, BatchedMLP
def timeit(fn, n):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
rs = np.random.RandomState(0)
out = {}
for dims in ((4, 64, 64, 64, 1), (3, 32, 32, 2), (8, 64, 64, 64, 1), (16, 64, 64, 64, 16)):
    arch = MLPArch(dims, "tanh")
    N, B = 4096, 64
    x = rs.rand(N, dims[0]) * 6 - 3; y = np.sin(x).sum(axis=1, keepdims=True) * np.ones((1, dims[-1]))
    op = BatchedMLP(arch, x, y)
    W = op.weights(0.1 * rs.randn(B, arch.nparams))
    out[str(dims)] = round(B / timeit(lambda: op.sse(W), 50))
print(json.dumps(out))
