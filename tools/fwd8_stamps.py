#!/usr/bin/env python3
"""Diagnostic: per-workgroup timeline of k_fused_fwd_i8 (library variant built with -DQN_FWD8_STAMPS by
`tools/ab_build2.py stampsf qn_fused_i8.hip -DQN_FWD8_STAMPS`): s_memrealtime (100 MHz) at kernel entry, after the staging,
after the row loop, at exit -- for every workgroup, relative to the earliest entry of the launch.  usage: tools/fwd8_stamps.py [B]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["QUINN_AMD_LIB"] = os.path.join(ROOT, "quinn_amd", "lib", "libquinn_amd_stampsf.so")
from quinn_amd.ops import MLPArch, BatchedMLP
arch = MLPArch((1, 64, 64, 64, 1), "tanh")
rs = np.random.RandomState(0)
x = rs.rand(4096, 1) * 6 - 3; y = np.sin(x)
for B in ([int(sys.argv[1])] if len(sys.argv) > 1 else [64, 8]):
    W = 0.1 * rs.randn(B, arch.nparams)
    op = BatchedMLP(arch, x, y)
    for _ in range(200):
        s, p = op.sse_pred(W)
    torch.cuda.synchronize()
    nwg = ((B + 7) // 8) * 8 * (512 // B if B < 64 else 8)
    t = p.reshape(-1)[-8 * nwg:].cpu().numpy().reshape(nwg, 8)
    t = t[(t > 1e9).all(axis=1)]                        # (slots a later workgroup overwrote with predictions are dropped)
    nwg = len(t)
    t = (t - t[:, 0].min()) / 100.0                      # us
    print(f"B = {B}: {nwg} workgroups; us relative to the first entry (median / max over workgroups)")
    for k, n in enumerate(["entry", "staging done", "row loop done", "exit"]):
        print(f"  {n:14s} {np.median(t[:, k]):7.2f} {t[:, k].max():7.2f}")
    for k, n in enumerate(["loads issued", "table + thin arrived", "matrix 1 sliced", "matrix 2 sliced"]):
        print(f"    staging: {n:22s} {np.median(t[:, 4 + k] - t[:, 0]):6.2f} us after entry")
    print(f"  per workgroup: staging {np.median(t[:, 1] - t[:, 0]):.2f} us, rows {np.median(t[:, 2] - t[:, 1]):.2f} us, tail {np.median(t[:, 3] - t[:, 2]):.2f} us")
