#!/usr/bin/env python3
"""Diagnostic: per-phase cycles of k_fused_bwd_i8 (library variant built with -DQN_BWD8_STAMPS by
`tools/ab_build2.py stamps8 qn_fused_bwd_i8.hip -DQN_BWD8_STAMPS`; never the shipped library)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["QUINN_AMD_LIB"] = os.path.join(ROOT, "quinn_amd", "lib", "libquinn_amd_stamps8.so")
from quinn_amd.ops import MLPArch, BatchedMLP
arch = MLPArch((1, 64, 64, 64, 1), "tanh")
rs = np.random.RandomState(0)
x = rs.rand(4096, 1) * 6 - 3; y = np.sin(x)
W = 0.1 * rs.randn(64, arch.nparams)
op = BatchedMLP(arch, x, y)
for _ in range(3):
    s, g = op.sse_grad(W)
torch.cuda.synchronize()
need = op.workspace_bytes(64, 4096, True) - 1024
st4 = op._ws[need:need + 4 * 96].cpu().numpy().view(np.int64).astype(np.float64).reshape(4, 12)
st = st4[0]
iters = 16
names = ["loop top + x load", "forward: first layer", "forward: hidden layers", "last layer, residual, dz_NH", "exponents + barrier A", "slice dz, transposes, stash writes",
         "barrier B", "dW + db products", "dA products + dz", "first layer backward", "staging (once per workgroup; x iters here)", "-"]
print("cycles per iteration (wave 0 of workgroup 0; s_memtime ticks = 100 MHz x ... see total); total %.0f" % (st.sum() / iters))
for n, v in zip(names, st):
    print("  %-36s %8.0f  %5.1f%%" % (n, v / iters, 100 * v / st.sum()))
print("per wave (cycles per iteration):")
for k, n in enumerate(names[:11]):
    print("  %-36s " % n + " ".join("%7.0f" % (st4[w][k] / iters) for w in range(4)))
