#!/usr/bin/env python3
"""configs[0]: ONE adaptive-Metropolis chain, 2x16 tanh MLP, 256 points -- the reference's own
CPU-runnable case -- through NN_MCMC.fit on the host (bit-exact) engine: full-loop steps/s."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quinn_amd.nns.mlp import MLP
from quinn_amd.solvers.nn_mcmc import NN_MCMC
rs = np.random.RandomState(0)
x = rs.rand(256, 1) * 2 * np.pi - np.pi
y = np.sin(x) + 0.02 * rs.randn(256, 1)
torch.manual_seed(0)
out = {}
for engine, nsteps in (('host', 1500), ('device', 6000)):
    for nch in (1, 64):
        uq = NN_MCMC(MLP(1, 1, (16, 16), activ='tanh'), verbose=False)
        kw = dict(zflag=False, datanoise=0.02, sampler='amcmc', sampler_params={'gamma': 0.01}, engine=engine)
        if nch > 1 or engine == 'device':
            kw['seeds'] = list(range(nch))
        np.random.seed(0)
        uq.fit(x, y, nmcmc=nsteps, **kw)              # warm-up of the same size (allocator)
        np.random.seed(0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        uq.fit(x, y, nmcmc=nsteps, **kw)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        key = f"{nch}_chains" if engine == 'host' else f"{nch}_chains_device_engine"
        out[key] = {"steps_per_s": nsteps / el, "chain_steps_per_s": nsteps * nch / el,
                    "accrate": float(np.mean(np.asarray(uq.mcmc_results['accrate'])))}
print(json.dumps(out))
