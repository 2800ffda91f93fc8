#!/usr/bin/env python3
"""Two ranks (gloo collectives, both on the one GPU of the box) run NN_MCMC.fit(engine='device') on 6 chains; every rank
ends with all 6 chains, equal to a single-process run of the same seeds up to the summation order of the SSE
(the row split per chain depends on the batch size).  python -m torch.distributed.run --nproc-per-node 2 tools/check_device_2rank.py"""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from quinn_amd.nns.mlp import MLP
from quinn_amd.solvers.nn_mcmc import NN_MCMC

torch.set_default_dtype(torch.double)
world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    dist.init_process_group("gloo")
rs = np.random.RandomState(0)
x = rs.rand(200, 1) * 4 - 2
y = np.sin(2 * x) + 0.1 * rs.randn(200, 1)
torch.manual_seed(0)
net = MLP(1, 1, (16, 16), activ='tanh')
uq = NN_MCMC(net, verbose=False)
uq.fit(x, y, zflag=False, datanoise=0.1, nmcmc=600, sampler='amcmc', sampler_params={'gamma': 0.1, 't0': 100, 'tadapt': 200},
       seeds=range(6), engine='device')
out = {"rank": int(os.environ.get("RANK", "0")), "world": world, "chains": list(uq.samples.shape),
       "accrate": [round(float(a), 4) for a in uq.mcmc_results['accrate']],
       "maxpost": [round(float(a), 6) for a in uq.mcmc_results['maxpost']]}
print(json.dumps(out), flush=True)
if world > 1:
    dist.barrier(); dist.destroy_process_group()
