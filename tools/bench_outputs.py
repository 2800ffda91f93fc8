#!/usr/bin/env python3
"""Forward launch time of the 64-chain 3x64 network at N = 4096 with 1 / 2 / 4 outputs (the int8-slice forward kernel takes up to four
outputs since round 3; before, o > 1 ran the float64-MFMA kernel: 0.109 / 0.115 ms at o = 2 / 4, now 0.086 / 0.091 ms)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from quinn_amd.ops import MLPArch, BatchedMLP
for o in (1, 2, 4):
    dims, N, B = (1, 64, 64, 64, o), 4096, 64
    arch = MLPArch(dims, "tanh"); rs = np.random.RandomState(0)
    x = rs.rand(N, 1) * 6 - 3; y = np.sin(x) * np.ones((1, o))
    op = BatchedMLP(arch, x, y); W = op.weights(0.3 * rs.randn(B, arch.nparams))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3: op.sse(W); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): op.sse(W)
    e1.record(); torch.cuda.synchronize()
    print("o", o, "ms per launch %.4f" % (e0.elapsed_time(e1) / 200), flush=True)
