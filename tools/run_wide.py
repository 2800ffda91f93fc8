#!/usr/bin/env python3
"""Forward and gradient calls at the shapes of BASELINE configs[2..4] (cfg3 / cfg4 / cfg5), for profiler passes
(tools/prof_any.sh wide "k_i8|k_dW|k_gemm|k_fwd|k_bwd|k_first|k_last|k_sse" tools/run_wide.py [cfg3,cfg4,cfg5] [reps])."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quinn_amd.ops import MLPArch, BatchedMLP
CFG = {"cfg3": ((2, 128, 128, 128, 1), 8192, 128), "cfg4": ((1, 256, 256, 256, 256, 1), 16384, 64),
       "cfg5": ((1, 256, 256, 256, 256, 1), 32768, 32)}
which = (sys.argv[1] if len(sys.argv) > 1 else "cfg3,cfg4").split(",")
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for name in which:
    dims, N, B = CFG[name]
    arch = MLPArch(dims, "tanh")
    rs = np.random.RandomState(0)
    x = rs.rand(N, dims[0]) * 6 - 3
    y = np.sin(x).sum(axis=1, keepdims=True)
    op = BatchedMLP(arch, x, y)
    W = op.weights(0.1 * rs.randn(B, arch.nparams))
    for fn, fl, kind in ((lambda: op.sse(W), arch.flops_fwd(N), "fwd"), (lambda: op.sse_grad(W), arch.flops_fwdbwd(N), "grad")):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / reps
        print(f"{name} {kind}: {1e3 * t:.3f} ms  {B * fl / t / 1e12:.1f} TFLOP/s", flush=True)
    del op, W
    torch.cuda.empty_cache()
