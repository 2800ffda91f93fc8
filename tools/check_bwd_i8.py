#!/usr/bin/env python3
"""Development check of k_fused_bwd_i8 (csrc/qn_fused_bwd_i8.hip): gradient / SSE of 64-wide tanh networks on the int8-slice
path against the float64-MFMA kernels (QN_PATH_FUSED_DP) and the layer-wise kernels (QN_PATH_GENERIC), then timings of both
paths in one process (A/B on one box).  usage: tools/check_bwd_i8.py [reps]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quinn_amd import _lib
from quinn_amd.ops import MLPArch, BatchedMLP

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rs = np.random.RandomState(0)
ok = True
for dims, N, B, ws in (((1, 64, 64, 64, 1), 4096, 64, 0.1), ((1, 64, 64, 64, 1), 100, 3, 1.0), ((2, 64, 64, 1), 333, 5, 0.5),
                       ((1, 64, 64, 64, 1), 64, 1, 0.3), ((2, 64, 64, 64, 1), 1000, 7, 2.0), ((1, 50, 50, 50, 1), 500, 4, 0.5)):
    arch = MLPArch(dims, "tanh")
    x = rs.rand(N, dims[0]) * 2 * np.pi - np.pi
    y = np.sin(x).sum(axis=1, keepdims=True) + 0.02 * rs.randn(N, 1)
    parts = []
    for a_, b_ in zip(dims[:-1], dims[1:]):
        parts.append(ws * rs.randn(B, b_ * a_) / np.sqrt(a_) * (3.0 if a_ > 2 else 1.0))
        parts.append(ws * rs.randn(B, b_))
    W = np.concatenate(parts, axis=1)
    res = {}
    for name, path in (("auto", _lib.PATH_AUTO), ("fused_dp", _lib.PATH_FUSED_DP), ("generic", _lib.PATH_GENERIC)):
        op = BatchedMLP(arch, x, y)
        op.set_path(path)
        s, g = op.sse_grad(W)
        torch.cuda.synchronize()
        res[name] = (s.cpu().numpy(), g.cpu().numpy())
    gmax = np.abs(res["generic"][1]).max(axis=1, keepdims=True)
    for other in ("fused_dp", "generic"):
        es = np.abs(res["auto"][0] / res[other][0] - 1).max()
        eg = (np.abs(res["auto"][1] - res[other][1]) / gmax).max()
        flag = "ok  " if es < 1e-11 and eg < 1e-10 else "FAIL"
        ok &= flag == "ok  "
        print(flag, dims, "N", N, "B", B, "ws", ws, "auto vs", other, "sse %.1e grad %.1e" % (es, eg), flush=True)
    e2 = (np.abs(res["fused_dp"][1] - res["generic"][1]) / gmax).max()
    print("     fused_dp vs generic grad %.1e" % e2)
# timing at the cfg2 shape
arch = MLPArch((1, 64, 64, 64, 1), "tanh")
N, B = 4096, 64
x = rs.rand(N, 1) * 2 * np.pi - np.pi
y = np.sin(x) + 0.02 * rs.randn(N, 1)
for name, path in (("auto (int8 slices)", _lib.PATH_AUTO), ("fused_dp (float64 MFMA)", _lib.PATH_FUSED_DP), ("auto (int8 slices)", _lib.PATH_AUTO)):
    op = BatchedMLP(arch, x, y)
    op.set_path(path)
    Wt = op.weights(0.1 * rs.randn(B, arch.nparams))
    for _ in range(5):
        op.sse_grad(Wt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        op.sse_grad(Wt)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    print("%-26s %.1f us per call, %.1f k gradient evals/s, %.1f TFLOP/s" % (name, el * 1e6, B / el / 1e3, B * arch.flops_fwdbwd(N) / el / 1e12), flush=True)
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
