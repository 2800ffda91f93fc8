#!/usr/bin/env python3
"""Update rates of the VI and ensemble trainers at BASELINE configs 3 and 4 (device random draws):
cfg3: one optimiser step of NN_VI (3 ELBO evaluations of 128 MC samples + 1 backward + Adam),
cfg4: one optimiser step of a 512-member ensemble, full batch (fwd+bwd on each member's 80 % subset,
validation forward, Adam)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quinn_amd.nns.mlp import MLP
from quinn_amd.solvers.nn_vi import NN_VI
from quinn_amd.solvers.nn_ens import NN_Ens
import quinn_amd.nns.nnfit as nf
DT = os.environ.get("QN_DTYPE", "float64")      # float32: the layer-wise kernels in single precision (master weights stay float64)
out = {"dtype": DT}
rs = np.random.RandomState(0)
# ---- cfg3
N = 8192
x = rs.rand(N, 2) * 2 * np.pi - np.pi
y = np.sin(x).sum(axis=1, keepdims=True) + 0.02 * rs.randn(N, 1)
torch.manual_seed(0)
vi = NN_VI(MLP(2, 1, (128, 128, 128), activ='tanh'), rng='device', dtype=DT)
vi.fit(x, y, val=[x[:1024], y[:1024]], datanoise=0.02, lrate=0.01, nsam=128, nepochs=1, freq_out=1000)
def vi_timed(ne):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    vi.fit(x, y, val=[x[:1024], y[:1024]], datanoise=0.02, lrate=0.01, nsam=128, nepochs=ne, freq_out=100000)
    torch.cuda.synchronize(); return time.perf_counter() - t0
# per optimiser step: difference of a 44-epoch and a 4-epoch fit, the smallest of three timings each (the fixed part of a fit --
# module copies, uploads, the two printed epochs' read-backs -- is not a step; round 3 / early round 4 divided a 5-epoch fit by 5)
el = (min(vi_timed(44) for _ in range(3)) - min(vi_timed(4) for _ in range(3))) / 40
out["cfg3_vi_step_s"] = el
out["cfg3_vi_mc_sample_evals_per_s"] = (128 * 2 + 128 * 1024 / N) / el     # train (fwd+bwd) + full (fwd) + val on 1/8 of the rows
# ---- cfg4
N = 16384
x = rs.rand(N, 1) * 2 * np.pi - np.pi
y = np.sin(x) + 0.02 * rs.randn(N, 1)
ens = NN_Ens(MLP(1, 1, (256, 256, 256, 256), activ='tanh'), nens=512, dfrac=0.8, dtype=DT)
ens.fit(x, y, val=[x[:2048], y[:2048]], lrate=0.01, nepochs=1, perm_mode='device', freq_out=1000)
def timed(ne):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ens.fit(x, y, val=[x[:2048], y[:2048]], lrate=0.01, nepochs=ne, perm_mode='device', freq_out=1000)
    torch.cuda.synchronize(); return time.perf_counter() - t0
# per optimiser step, without the per-learner host bookkeeping around the run: difference of an 18-epoch and a 2-epoch fit, the
# smaller of two timings each (round 3 took (t5 - t1) / 4 of single timings: the fixed part scatters by +-0.3 s between calls,
# which moved the quotient between 0.09 and 0.20 s)
t1 = min(timed(2), timed(2))
t9 = min(timed(18), timed(18))
el = (t9 - t1) / 16
out["cfg4_fit_fixed_overhead_s"] = t1 - el
out["cfg4_ens_step_s"] = el
out["cfg4_member_updates_per_s"] = 512 / el
print(json.dumps(out))
