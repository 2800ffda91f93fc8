#!/usr/bin/env python3
"""Forward (log-posterior) rate of 64-wide networks by activation: the int8-slice kernel (tanh; relu / identity since round 4)
against the float64-MFMA fused kernel (QN_PATH_FUSED_DP), cfg2 sizes (64 chains, N = 4096).  Method of bench.py's graph_rate."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from quinn_amd import _lib
from quinn_amd.ops import MLPArch, BatchedMLP
N, B = 4096, 64
x, y = bench.synthetic(N, 1)
dev = torch.device("cuda")
for hid in ((64, 64, 64), (64, 64)):
    for act in ("tanh", "relu", "identity"):
        arch = MLPArch((1,) + hid + (1,), act)
        op = BatchedMLP(arch, x, y)
        W = op.weights(np.random.RandomState(1).randn(B, arch.nparams) * (0.1 if act == "tanh" else 0.2))
        out = {}
        for name, path in (("auto", _lib.PATH_AUTO), ("fused_dp", _lib.PATH_FUSED_DP)):
            op.set_path(path)
            t, tmin, tmax = bench.graph_rate(lambda: op.sse(W), dev)
            out[name] = (B * arch.flops_fwd(N) / t / 1e12, op.arith(B, N, False), 1e6 * t)
        gout = {}
        for name, path in (("auto", _lib.PATH_AUTO), ("fused_dp", _lib.PATH_FUSED_DP)):
            op.set_path(path)
            t, tmin, tmax = bench.graph_rate(lambda: op.sse_grad(W), dev)
            gout[name] = (B * arch.flops_fwdbwd(N) / t / 1e12, op.arith(B, N, True), 1e6 * t)
        op.set_path(_lib.PATH_AUTO); a = op.sse(W); ga = op.sse_grad(W)[1]; op.set_path(_lib.PATH_FUSED_DP); b = op.sse(W); gb = op.sse_grad(W)[1]
        err = float(((a - b).abs() / b.abs()).max())
        gerr = float(((ga - gb).abs().amax(dim=1) / gb.abs().amax(dim=1)).max())
        print(f"{len(hid)}x64 {act:8s} gradient: auto (arith {gout['auto'][1]}) {gout['auto'][0]:6.1f} TFLOP/s = {gout['auto'][0] / 78.6:.3f}  {gout['auto'][2]:.1f} us | "
              f"f64 MFMA {gout['fused_dp'][0]:6.1f} TFLOP/s = {gout['fused_dp'][0] / 78.6:.3f} | max |dg| / max|g| {gerr:.2e}", flush=True)
        print(f"{len(hid)}x64 {act:8s}: auto (arith {out['auto'][1]}) {out['auto'][0]:6.1f} TFLOP/s = {out['auto'][0] / 78.6:.3f}  {out['auto'][2]:.1f} us | "
              f"f64 MFMA {out['fused_dp'][0]:6.1f} TFLOP/s = {out['fused_dp'][0] / 78.6:.3f} | max rel diff of SSE {err:.2e}", flush=True)
