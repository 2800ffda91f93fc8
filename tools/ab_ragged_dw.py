#!/usr/bin/env python3
"""Gradient time at ragged / odd row counts with the library named by QUINN_AMD_LIB (in-call A/B: the per-chunk weight-gradient kernel
k_i8_dw against k_i8_dw_g on the whole chunks + the float64 tail; build the old dispatch with QN_HIPCC_FLAGS=-DQN_DW_RAGGED_OLD)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from quinn_amd.ops import MLPArch, BatchedMLP
dev = torch.device("cuda")
for dims, N, B, act in (((2, 128, 128, 128, 1), 8191, 128, "tanh"), ((2, 128, 128, 128, 1), 8136, 128, "tanh"), ((2, 128, 128, 128, 1), 8191, 128, "relu"),
                        ((1, 256, 256, 256, 256, 1), 13107, 128, "tanh"), ((1, 256, 256, 256, 256, 1), 13107, 128, "relu")):
    x, y = bench.synthetic(N, dims[0])
    arch = MLPArch(dims, act)
    op = BatchedMLP(arch, x, y)
    W = op.weights(np.random.RandomState(7).randn(B, arch.nparams) / np.sqrt(dims[1]))
    t, tmin, tmax = bench.graph_rate(lambda: op.sse_grad(W), dev)
    print(os.path.basename(os.environ.get("QUINN_AMD_LIB", "default")), dims[1], act, N, f"gradient {1e3 * t:.4f} ms = {B * arch.flops_fwdbwd(N) / t / 1e12:.2f} TFLOP/s", flush=True)
    del op, W
    torch.cuda.empty_cache()
