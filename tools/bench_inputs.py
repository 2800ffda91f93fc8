#!/usr/bin/env python3
"""Gradient launch time of the 64-chain 3x64 network at N = 4096 with 1..4 inputs: the sliced int8 backward kernel (QN_PATH_FUSED;
d = 3, 4 since the end of round 3) against the float64-MFMA fused kernel (QN_PATH_FUSED_DP)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quinn_amd import _lib
from quinn_amd.ops import MLPArch, BatchedMLP
for d in (1, 2, 3, 4):
    dims, N, B = (d, 64, 64, 64, 1), 4096, 64
    arch = MLPArch(dims, "tanh"); rs = np.random.RandomState(0)
    x = rs.rand(N, d) * 6 - 3; y = np.sin(x).sum(axis=1, keepdims=True)
    op = BatchedMLP(arch, x, y); W = op.weights(0.3 * rs.randn(B, arch.nparams))
    res = {}
    for name, path in (("int8", _lib.PATH_FUSED), ("f64", _lib.PATH_FUSED_DP)):
        op.set_path(path)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3: op.sse_grad(W); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): op.sse_grad(W)
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 100
    print("d", d, "ms per gradient launch: int8 %.4f  f64 %.4f" % (res["int8"], res["f64"]), flush=True)
