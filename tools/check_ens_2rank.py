#!/usr/bin/env python3
"""NN_Ens.fit / NN_RMS.fit on 5 members under 1 or 2 ranks (gloo collectives; on a 1-GPU box both ranks use cuda:0): members
shard over ranks, every rank consumes the random streams of ALL members in the reference's order, ONE all_gather returns
the results -- so every rank must hold what the single process holds (up to the summation order of a batched reduction).
    python tools/check_ens_2rank.py [ens|rms]                                  (one process)
    python -m torch.distributed.run --nproc-per-node 2 tools/check_ens_2rank.py rms
tests/test_gpu_00_launch.py starts both as children and compares them."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from quinn_amd.nns.mlp import MLP
from quinn_amd.solvers.nn_ens import NN_Ens
from quinn_amd.solvers.nn_rms import NN_RMS

kind = sys.argv[1] if len(sys.argv) > 1 else "ens"
torch.set_default_dtype(torch.double)
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
if world > 1:
    dist.init_process_group("gloo")
rs = np.random.RandomState(0)
x = rs.rand(90, 2) * 4 - 2
y = np.sin(x.sum(axis=1, keepdims=True)) + 0.1 * rs.randn(90, 1)
xv = rs.rand(30, 2) * 4 - 2
yv = np.sin(xv.sum(axis=1, keepdims=True))
torch.manual_seed(3)
np.random.seed(4)
net = MLP(2, 1, (16, 16), activ='tanh')
if kind == "ens":
    uq = NN_Ens(net, nens=5, dfrac=0.8, verbose=False)
    uq.fit(x, y, val=[xv, yv], lrate=0.01, batch_size=20, nepochs=15, freq_out=100000)
else:
    uq = NN_RMS(net, nens=5, dfrac=0.8, datanoise=0.1, priorsigma=1.0, verbose=False)
    uq.fit(x, y, val=[xv, yv], lrate=0.01, batch_size=20, nepochs=15, freq_out=100000)
r = uq.fit_results
np.random.seed(7)
pred = uq.predict_ens(xv)
out = {"rank": rank, "world": world, "kind": kind, "members": int(r['best_w'].shape[0]),
       "best_w_checksum": [round(float(a), 8) for a in np.asarray(r['best_w']).sum(axis=1)],
       "final_w_checksum": [round(float(a), 8) for a in np.asarray(r['final_w']).sum(axis=1)],
       "best_loss": [round(float(np.min(np.asarray(h)[:, 3])), 8) for h in r['history']],
       "pred_checksum": [round(float(a), 8) for a in np.asarray(pred).reshape(pred.shape[0], -1).sum(axis=1)]}
print(json.dumps(out), flush=True)
if world > 1:
    dist.barrier(); dist.destroy_process_group()
