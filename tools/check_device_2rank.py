#!/usr/bin/env python3
"""NN_MCMC.fit(engine='device') on 6 chains under 1 or 2 ranks (gloo collectives; on a 1-GPU box both ranks use cuda:0):
prints one JSON line per rank with what the API returned.  Chains are keyed by their GLOBAL id, so the 2-rank run must
equal the single-process run up to the summation order of a chain's SSE (the row split depends on the batch size).
    python tools/check_device_2rank.py [amcmc|hmc] [all|root|none]            (one process)
    python -m torch.distributed.run --nproc-per-node 2 tools/check_device_2rank.py hmc root
tests/test_gpu_00_launch.py starts both as children (before its own process touches the GPU) and compares them."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from quinn_amd.nns.mlp import MLP
from quinn_amd.solvers.nn_mcmc import NN_MCMC

sampler = sys.argv[1] if len(sys.argv) > 1 else "amcmc"
gather = sys.argv[2] if len(sys.argv) > 2 else "all"
torch.set_default_dtype(torch.double)
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
if world > 1:
    dist.init_process_group("gloo")
rs = np.random.RandomState(0)
x = rs.rand(200, 1) * 4 - 2
y = np.sin(2 * x) + 0.1 * rs.randn(200, 1)
torch.manual_seed(0)
net = MLP(1, 1, (16, 16), activ='tanh')
uq = NN_MCMC(net, verbose=False)
if sampler == "amcmc":
    uq.fit(x, y, zflag=False, datanoise=0.1, nmcmc=600, sampler='amcmc', sampler_params={'gamma': 0.1, 't0': 100, 'tadapt': 200},
           seeds=range(6), engine='device', gather=gather)
else:
    uq.fit(x, y, zflag=False, datanoise=0.1, nmcmc=60, sampler='hmc', sampler_params={'epsilon': 0.002, 'L': 3},
           seeds=range(6), engine='device', gather=gather)
r = uq.mcmc_results
out = {"rank": rank, "world": world, "sampler": sampler, "gather": gather, "chains": list(uq.samples.shape),
       "accrate": [round(float(a), 4) for a in r['accrate']], "maxpost": [round(float(a), 5) for a in r['maxpost']],
       "last_logpost": [round(float(a), 5) for a in r['logpost'][:, -1]],
       "chain_checksum": [round(float(a), 6) for a in r['chain'].sum(axis=(1, 2))]}
print(json.dumps(out), flush=True)
if world > 1:
    dist.barrier(); dist.destroy_process_group()
