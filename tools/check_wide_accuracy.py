#!/usr/bin/env python3
"""Accuracy of the int8-slice kernels for wide networks against the exact float64 layer-wise kernels (QN_PATH_GENERIC):
relative error of SSE, predictions and gradients at the cfg3 / cfg4 network shapes (reduced row / member counts)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quinn_amd import _lib
from quinn_amd.ops import MLPArch, BatchedMLP
for dims, N, B, ws in (((2, 128, 128, 128, 1), 8192, 16, 0.1), ((1, 256, 256, 256, 256, 1), 16384, 8, 0.1), ((1, 256, 256, 256, 256, 1), 4096, 8, 1.0)):
    arch = MLPArch(dims, "tanh"); rs = np.random.RandomState(0)
    x = rs.rand(N, dims[0]) * 2 * np.pi - np.pi; y = np.sin(x).sum(axis=1, keepdims=True) + 0.02 * rs.randn(N, 1)
    op = BatchedMLP(arch, x, y); W = op.weights(ws * rs.randn(B, arch.nparams) / (1.0 if ws < 1 else np.sqrt(dims[1])))
    out = {}
    for path in (_lib.PATH_AUTO, _lib.PATH_GENERIC):
        op.set_path(path)
        s, g = op.sse_grad(W); s2, pr = op.sse_pred(W)
        out[path] = [t.double().cpu().numpy() for t in (s, g, pr)]
    a, r = out[_lib.PATH_AUTO], out[_lib.PATH_GENERIC]
    print(dims, "N", N, "B", B, "wscale", ws, "| sse rel %.2e | pred max/scale %.2e | grad max/max|g| %.2e" % (
        np.abs(a[0] / r[0] - 1).max(), np.abs(a[2] - r[2]).max() / np.abs(r[2]).max(), np.abs(a[1] - r[1]).max() / np.abs(r[1]).max()), flush=True)
