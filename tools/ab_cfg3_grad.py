#!/usr/bin/env python3
"""cfg3 / cfg4-shape gradient time with the library named by QUINN_AMD_LIB (in-call A/B of kernel variants: run it once per library)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from quinn_amd.ops import MLPArch, BatchedMLP
dev = torch.device("cuda")
for dims, N, B in (((2, 128, 128, 128, 1), 8192, 128), ((1, 256, 256, 256, 256, 1), 16384, 64)):
    x, y = bench.synthetic(N, dims[0])
    arch = MLPArch(dims, "tanh")
    op = BatchedMLP(arch, x, y)
    W = op.weights(0.1 * np.random.RandomState(7).randn(B, arch.nparams))
    t, tmin, tmax = bench.graph_rate(lambda: op.sse_grad(W), dev)
    print(os.path.basename(os.environ.get("QUINN_AMD_LIB", "default")), dims[1], f"gradient {1e3 * t:.4f} ms = {B * arch.flops_fwdbwd(N) / t / 1e12:.2f} TFLOP/s", flush=True)
