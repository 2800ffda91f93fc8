#!/usr/bin/env python3
"""Randomised parity sweep of the int8-slice kernels (QN_PATH_AUTO: the fused 64-wide forward, the 128 / 256-wide forward,
activation-gradient and weight-gradient kernels, the layer-wise int8 forward for d > 4) against the float64 kernels
(QN_PATH_GENERIC): random widths, depths, input counts, row counts,
vector counts, bias on / off, row subsets, weight scales from 1e-3 (tiny activations) to 4 (saturated, chaotic).
usage: tools/fuzz_wide.py [ncases] [seed] [big]"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quinn_amd import _lib
from quinn_amd.ops import MLPArch, BatchedMLP
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
big = len(sys.argv) > 3 and sys.argv[3] == "big"
worst = [0.0, 0.0, 0.0]
for case in range(ncases):
    h = int(rs.choice([64, 128, 256])); nhid = int(rs.randint(2, 6 if h <= 128 else 5)); d = int(rs.choice([1, 2, 3, 4, 4, 6]))
    N = int(rs.choice([rs.randint(1, 70), rs.randint(70, 700), rs.randint(700, 3000)])); B = int(rs.choice([1, 2, rs.randint(3, 40)]))
    if big:                                         # many rows / many vectors: other row splits, grids beyond one wave of workgroups
        N = int(rs.randint(3000, 40000)); B = int(rs.randint(1, max(2, 1500000 // N)))
        if rs.rand() < 0.3: N, B = int(rs.randint(1, 300)), int(rs.randint(300, 3000))
    if h >= 128 and rs.rand() < 0.5: N = max(64, N // 64 * 64)      # whole 64-row chunks: the group-scale weight-gradient kernel
    bias = bool(rs.rand() < 0.8); wscale = float(rs.choice([1e-3, 0.1, 1.0, 4.0])) / np.sqrt(h)
    dims = (d,) + (h,) * nhid + (1,)
    arch = MLPArch(dims, "tanh", bias=bias)
    x = rs.rand(N, d) * 2 * np.pi - np.pi; y = np.sin(x).sum(axis=1, keepdims=True) + 0.02 * rs.randn(N, 1)
    W = wscale * rs.randn(B, arch.nparams) * np.sqrt(h)
    idx = rs.randint(0, N, size=(B, int(rs.randint(1, N + 1)))) if rs.rand() < 0.4 else None
    op = BatchedMLP(arch, x, y)
    out = {}
    for path in (_lib.PATH_AUTO, _lib.PATH_GENERIC):
        op.set_path(path)
        s, g = op.sse_grad(W, row_idx=idx); s2, pr = op.sse_pred(W, row_idx=idx)
        out[path] = [t.double().cpu().numpy() for t in (s, g, s2, pr)]
    a, r = out[_lib.PATH_AUTO], out[_lib.PATH_GENERIC]
    e = [np.abs(a[0] / r[0] - 1).max(), np.abs(a[1] - r[1]).max() / max(np.abs(r[1]).max(), 1e-300), np.abs(a[3] - r[3]).max() / max(np.abs(r[3]).max(), 1e-300)]
    # weights ~ N(0, 16): deep saturated networks amplify rounding differences (the float64 kernels differ from the oracle
    # by up to 1e-11 there; the 47-bit operands of the int8-slice kernels by ~64 times that: tests/check_chaotic_regime.py)
    f = 30.0 if wscale * np.sqrt(h) >= 4 else 1.0
    ok = e[0] <= 1e-11 * f and e[1] <= 1e-10 * f and e[2] <= 1e-11 * f and np.abs(a[2] / r[2] - 1).max() <= 1e-11 * f
    worst = [max(u, v) for u, v in zip(worst, e)]
    print(("ok  " if ok else "FAIL"), dims, "N", N, "B", B, "bias", bias, "rows", None if idx is None else idx.shape[1], "wscale %.3g" % (wscale * np.sqrt(h)),
          "| sse %.1e grad %.1e pred %.1e" % tuple(e), flush=True)
    del op
    torch.cuda.empty_cache()
print("worst: sse %.2e grad %.2e pred %.2e" % tuple(worst))
