#!/usr/bin/env python3
"""Diagnostic: in-kernel stamps of k_i8_wide_fwd / k_i8_wide_bwd (library variant built with -DQN_WIDE_STAMPS by
`tools/ab_build2.py widest qn_wide_i8.hip -DQN_WIDE_STAMPS`): gradient calls at the cfg4 and cfg3 shapes."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["QUINN_AMD_LIB"] = os.path.join(ROOT, "quinn_amd", "lib", "libquinn_amd_widest.so")
from quinn_amd.ops import MLPArch, BatchedMLP
for dims, N, B in (((1, 256, 256, 256, 256, 1), 16384, 64), ((2, 128, 128, 128, 1), 8192, 128)):
    arch = MLPArch(dims, "tanh"); rs = np.random.RandomState(0)
    x = rs.rand(N, dims[0]) * 6 - 3; y = np.sin(x).sum(axis=1, keepdims=True)
    op = BatchedMLP(arch, x, y); W = op.weights(0.1 * rs.randn(B, arch.nparams))
    for _ in range(12): op.sse_grad(W)
    torch.cuda.synchronize()
    print("----", dims, flush=True)
