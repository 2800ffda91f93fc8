#!/usr/bin/env python3
"""Device AMCMC at cfg2 with the 64 chains split into G groups on G HIP streams (DeviceAMCMC(groups=G)): one
group's small kernels (accept, apply-delta, partial sums: a fifth of a step, one workgroup per chain) overlap the
other groups' forward kernels.  Chains are keyed by their global index; only the summation order of a chain's SSE
changes with the batch size (row splits per chain), so chains agree in distribution, not bit for bit."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quinn_amd.ops import MLPArch, BatchedMLP
from quinn_amd.mcmc.device_amcmc import DeviceAMCMC

C, N = 64, 4096
NMCMC = int(os.environ.get("NMCMC", "3000"))
arch = MLPArch((1, 64, 64, 64, 1), "tanh")
rs = np.random.RandomState(0)
x = rs.rand(N, 1) * 2 * np.pi - np.pi
y = 0.02 * rs.randn(N, 1) + np.sin(x)
ini = np.stack([np.random.RandomState(1000 + c).rand(arch.nparams) for c in range(C)])
out = {"nmcmc": NMCMC}
op = BatchedMLP(arch, x, y)
for G in [int(g) for g in os.environ.get("NGROUPS", "1,2,4").split(",")]:
    eng = DeviceAMCMC(op, 0.02, gamma=0.01, t0=100, tadapt=1000, seed=1, groups=G, use_graph=os.environ.get("USE_GRAPH", "0") == "1")
    eng.run(80, ini, store_chain=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = eng.run(NMCMC, ini, store_chain=False)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    out[f"groups{G}_steps_per_s"] = round(NMCMC / el, 1)
    out[f"groups{G}_accrate_mean"] = round(float(r['accrate'].mean()), 4)
    out[f"groups{G}_maxpost_mean"] = float(r['maxpost'].mean())
    del eng, r
    torch.cuda.empty_cache()
print(json.dumps(out))
