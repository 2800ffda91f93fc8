#!/usr/bin/env python3
"""Networks with 5..16 inputs or outputs: gradient and forward launch times on the layer-wise kernels (QN_PATH_GENERIC: where such gradients ran
until round 4) against the default dispatch (the fused float64-MFMA kernels' DP = 8 instances, qn_fused_d8.hip)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quinn_amd import _lib
from quinn_amd.ops import MLPArch, BatchedMLP


def rate(f):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50


for dims, act, N, B in (((6, 64, 64, 64, 1), "tanh", 4096, 64), ((8, 64, 64, 64, 1), "relu", 4096, 64), ((5, 32, 32, 1), "tanh", 4096, 64),
                        ((8, 16, 16, 1), "tanh", 256, 64), ((6, 11, 11, 11, 1), "tanh", 1000, 256), ((12, 32, 32, 1), "tanh", 4096, 64),
                        ((16, 20, 20, 20, 1), "relu", 1000, 256), ((10, 64, 64, 64, 1), "relu", 4096, 64), ((12, 64, 64, 1), "tanh", 4096, 64), ((2, 32, 32, 8), "tanh", 4096, 64),
                        ((8, 20, 20, 10), "relu", 1000, 256), ((1, 64, 64, 64, 6), "tanh", 4096, 64), ((10, 64, 64, 12), "relu", 4096, 64)):
    arch = MLPArch(dims, act); rs = np.random.RandomState(0)
    x = rs.rand(N, dims[0]) * 6 - 3; y = np.sin(x).sum(axis=1, keepdims=True) * np.ones((1, dims[-1]))
    op = BatchedMLP(arch, x, y); W = op.weights(0.3 * rs.randn(B, arch.nparams))
    res = {}
    for name, path in (("layerwise", _lib.PATH_GENERIC), ("default", _lib.PATH_AUTO)):
        op.set_path(path)
        res[name] = (rate(lambda: op.sse(W)), rate(lambda: op.sse_grad(W)), op.path(B, N, False), op.path(B, N, True))
    fl_f, fl_g = B * arch.flops_fwd(N) / 1e9, B * arch.flops_fwdbwd(N) / 1e9
    a, b = res["layerwise"], res["default"]
    print(f"{dims} {act} N={N} B={B}: forward {a[0]:.4f} -> {b[0]:.4f} ms ({fl_f / b[0]:.1f} TFLOP/s), gradient {a[1]:.4f} -> {b[1]:.4f} ms "
          f"({fl_g / b[1]:.1f} TFLOP/s)  paths fwd {a[2]}->{b[2]} grad {a[3]}->{b[3]}", flush=True)
