#!/usr/bin/env python3
"""Device HMC / MALA at cfg2: direct launches against HIP-graph replay of step pairs (DeviceHMC(use_graph=True)), alternating in one call."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quinn_amd.ops import MLPArch, BatchedMLP
from quinn_amd.mcmc.device_hmc import DeviceHMC
from quinn_amd.mcmc.device_mala import DeviceMALA
N, C = 4096, 64
rs = np.random.RandomState(0)
x = rs.rand(N, 1) * 2 * np.pi - np.pi
y = 0.02 * rs.randn(N, 1) + np.sin(x)
arch = MLPArch((1, 64, 64, 64, 1), "tanh")
op = BatchedMLP(arch, x, y)
ini = np.stack([0.1 * np.random.RandomState(1000 + c).randn(arch.nparams) for c in range(C)])
for name, mk in (("hmc L=3", lambda g: DeviceHMC(op, 0.02, epsilon=3.5e-5, L=3, seed=1, use_graph=g)),
                 ("mala", lambda g: DeviceMALA(op, 0.02, epsilon=7.9e-5, seed=1, use_graph=g))):
    for g in (False, True, False, True):
        eng = mk(g)
        eng.run(60, ini, store_chain=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = eng.run(600, ini, store_chain=False)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        print(name, "graph" if g else "direct", f"{600 / el:.1f} steps/s  acc {float(r['accrate'].mean()):.3f}", flush=True)
