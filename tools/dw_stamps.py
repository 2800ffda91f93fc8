#!/usr/bin/env python3
"""Diagnostic: in-kernel stamps of k_i8_dw_g / k_i8_dw (library variant built by `python tools/ab_build2.py dwstamps qn_dw_i8.hip -DQN_DW_STAMPS`): one cfg4-shape gradient call."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["QUINN_AMD_LIB"] = os.path.join(ROOT, "quinn_amd", "lib", "libquinn_amd_dwstamps.so")
from quinn_amd.ops import MLPArch, BatchedMLP
dims, N, B = (1, 256, 256, 256, 256, 1), 16384, 32
arch = MLPArch(dims, "tanh"); rs = np.random.RandomState(0)
x = rs.rand(N, 1) * 6 - 3; y = np.sin(x)
op = BatchedMLP(arch, x, y); W = op.weights(0.1 * rs.randn(B, arch.nparams))
for _ in range(30): op.sse_grad(W)
torch.cuda.synchronize()
