#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the fused backward kernel (QN_BWD_STAMPS build).
Builds a SEPARATE library (never the shipped one), runs cfg2 once, prints cycles per iteration."""
import ctypes, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quinn_amd import _lib
so = os.path.join(ROOT, "gpurun_out", "libquinn_amd_stamps.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
if True:
    srcs = [os.path.join(_lib.CSRC, s) for s in _lib.SOURCES]
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-DQN_BWD_STAMPS", "-o", so] + srcs, check=True)
_lib.LIBPATH = so
from quinn_amd.ops import MLPArch, BatchedMLP
arch = MLPArch((1, 64, 64, 64, 1), "tanh")
rs = np.random.RandomState(0)
x = rs.rand(4096, 1) * 6 - 3; y = np.sin(x)
W = 0.1 * rs.randn(64, arch.nparams)
op = BatchedMLP(arch, x, y)
for _ in range(3):
    s, g = op.sse_grad(W)
torch.cuda.synchronize()
ws = op._ws
need = op.workspace_bytes(64, 4096, True) - 1024
st = ws[need:need + 96].cpu().numpy().view(np.int64).reshape(1, 12).astype(np.float64)
iters = 16
names = ["loop top+x load", "forward layers", "last layer+resid", "barrier A(last)", "stash+barrier B(last)", "colsum last + dz", "hidden: barriers+stash", "hidden: dW MFMA", "hidden: db colsum", "hidden: dA MFMA+dz", "first-layer stage", "-"]
tot = st.mean(axis=0)
print("cycles per iteration (wave 0 of split 0, mean over chains); total %.0f" % (tot.sum() / iters))
for n, v in zip(names, tot):
    print("  %-26s %8.0f  %5.1f%%" % (n, v / iters, 100 * v / tot.sum()))
