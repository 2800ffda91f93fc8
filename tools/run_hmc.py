#!/usr/bin/env python3
"""Device HMC at cfg2 (64 chains, 3x64 tanh, N = 4096, L = 3, step size at acceptance ~0.6) for profiler passes:
tools/prof_any.sh hmc "k_hmc|k_accept|k_fused_bwd|k_grad_reduce" tools/run_hmc.py [steps]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quinn_amd.ops import MLPArch, BatchedMLP
from quinn_amd.mcmc.device_hmc import DeviceHMC
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rs = np.random.RandomState(0)
x = rs.rand(4096, 1) * 2 * np.pi - np.pi
y = 0.02 * rs.randn(4096, 1) + np.sin(x)
arch = MLPArch((1, 64, 64, 64, 1), "tanh")
op = BatchedMLP(arch, x, y)
ini = np.stack([0.1 * np.random.RandomState(1000 + c).randn(arch.nparams) for c in range(64)])
eng = DeviceHMC(op, 0.02, epsilon=3.5e-5, L=3, seed=1)
eng.run(30, ini, store_chain=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
r = eng.run(nsteps, ini, store_chain=False)
torch.cuda.synchronize(); el = time.perf_counter() - t0
print(f"{nsteps / el:.1f} steps/s, acceptance {float(r['accrate'].mean()):.2f}", flush=True)
