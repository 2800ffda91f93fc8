#!/usr/bin/env python3
"""Diagnostic: does a third workgroup (wave) per CU (SIMD) speed up k_fused_fwd_i8?  A 2x64 network needs 44 KB of LDS per
workgroup (3 fit a CU); QN_DEBUG_LDS_PAD (library built with -DQN_DEBUG_LDS_PAD) pads the request so that only 2 fit.
192 chains x 8 row splits = 1536 workgroups: whole rounds either way."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["QUINN_AMD_LIB"] = os.path.join(ROOT, "quinn_amd", "lib", "libquinn_amd_ldspad.so")
from quinn_amd.ops import MLPArch, BatchedMLP
dims, N, B = (1, 64, 64, 1), 4096, int(os.environ.get("B", "192"))
arch = MLPArch(dims, "tanh"); rs = np.random.RandomState(0)
x = rs.rand(N, 1) * 6 - 3; y = np.sin(x)
op = BatchedMLP(arch, x, y); W = op.weights(0.3 * rs.randn(B, arch.nparams))
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:
    op.sse(W); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): op.sse(W)
e1.record(); torch.cuda.synchronize()
print("pad", os.environ.get("QN_DEBUG_LDS_PAD", "0"), "B", B, "ms per launch %.4f" % (e0.elapsed_time(e1) / 200))
