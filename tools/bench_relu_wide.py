#!/usr/bin/env python3
"""Forward / gradient rates of 128- and 256-wide networks by activation at the cfg3 / cfg4 shapes: the int8-slice kernels (tanh;
relu / identity since round 4: k_i8_wide_fwd_u, k_i8_wide_bwd<.., false>) against the float64 layer-wise kernels
(QN_PATH_GENERIC forced), and tanh as the yardstick.  Method of bench.py's graph_rate."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from quinn_amd import _lib
from quinn_amd.ops import MLPArch, BatchedMLP
dev = torch.device("cuda")
RAGGED = int(os.environ.get("RAGGED", "0"))        # rows taken off each size: an odd / ragged row count (the int8 weight gradient + float64 tail)
for dims, N, B in (((2, 128, 128, 128, 1), 8192 - RAGGED, 128), ((1, 256, 256, 256, 256, 1), 16384 - RAGGED, 64)):
    x, y = bench.synthetic(N, dims[0])
    for act in ("tanh", "relu", "identity"):
        arch = MLPArch(dims, act)
        op = BatchedMLP(arch, x, y)
        W = op.weights(np.random.RandomState(1).randn(B, arch.nparams) / np.sqrt(dims[1]))
        res = {}
        for name, path in (("auto", _lib.PATH_AUTO), ("f64", _lib.PATH_GENERIC)):
            op.set_path(path)
            tf, _, _ = bench.graph_rate(lambda: op.sse(W), dev)
            tg, _, _ = bench.graph_rate(lambda: op.sse_grad(W), dev)
            res[name] = (B * arch.flops_fwd(N) / tf / 1e12, B * arch.flops_fwdbwd(N) / tg / 1e12, 1e3 * tf, 1e3 * tg,
                         op.arith(B, N, False), op.arith(B, N, True))
        op.set_path(_lib.PATH_AUTO); a = op.sse(W); ga = op.sse_grad(W)[1]
        op.set_path(_lib.PATH_GENERIC); b = op.sse(W); gb = op.sse_grad(W)[1]
        err = float(((a - b).abs() / b.abs()).max())
        gerr = float(((ga - gb).abs().amax(dim=1) / gb.abs().amax(dim=1)).max())
        ra, rf = res["auto"], res["f64"]
        print(f"{dims[1]}x{len(dims) - 2} N={N} B={B} {act:8s}: forward auto (arith {ra[4]}) {ra[0]:6.1f} TFLOP/s = {ra[0] / 78.6:.3f} {ra[2]:.3f} ms | "
              f"layer-wise f64 {rf[0]:6.1f} = {rf[0] / 78.6:.3f} | gradient auto (arith {ra[5]}) {ra[1]:6.1f} TFLOP/s = {ra[1] / 78.6:.3f} {ra[3]:.3f} ms | "
              f"layer-wise f64 {rf[1]:6.1f} = {rf[1] / 78.6:.3f} | max rel diff SSE {err:.2e}, max |dg| / max |g| {gerr:.2e}", flush=True)
        del op
        torch.cuda.empty_cache()
