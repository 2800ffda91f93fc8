"""Device AMCMC before the first adaptation at cfg2: time the enqueuing thread needs per step against the time the GPU needs
(1 and 2 chain groups)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quinn_amd.ops import MLPArch, BatchedMLP
from quinn_amd.mcmc.device_amcmc import DeviceAMCMC
C, N = 64, 4096
arch = MLPArch((1, 64, 64, 64, 1), "tanh")
rs = np.random.RandomState(0)
x = rs.rand(N, 1) * 2 * np.pi - np.pi
y = 0.02 * rs.randn(N, 1) + np.sin(x)
op = BatchedMLP(arch, x, y)
ini = np.stack([np.random.RandomState(1000 + c).rand(arch.nparams) for c in range(C)])
for G in (1, 2):
    eng = DeviceAMCMC(op, 0.02, gamma=0.01, t0=100, tadapt=100000, seed=1, groups=G)
    eng.run(200, ini, store_chain=False)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.run(4000, ini, store_chain=False)
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(G, "enqueue %.1f us/step  total %.1f us/step  -> %.0f steps/s" % ((t1 - t0) / 4000 * 1e6, (t2 - t0) / 4000 * 1e6, 4000 / (t2 - t0)), flush=True)
