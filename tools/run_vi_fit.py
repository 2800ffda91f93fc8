#!/usr/bin/env python3
"""A cfg3 NN_VI fit of a few epochs, for profiler passes (rocprofv3 --kernel-trace --stats -- python3 tools/run_vi_fit.py [epochs])."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quinn_amd.nns.mlp import MLP
from quinn_amd.solvers.nn_vi import NN_VI
rs = np.random.RandomState(0)
N = 8192
x = rs.rand(N, 2) * 2 * np.pi - np.pi
y = np.sin(x).sum(axis=1, keepdims=True) + 0.02 * rs.randn(N, 1)
torch.manual_seed(0)
vi = NN_VI(MLP(2, 1, (128, 128, 128), activ='tanh'), rng='device')
vi.fit(x, y, val=[x[:1024], y[:1024]], datanoise=0.02, lrate=0.01, nsam=128, nepochs=int(sys.argv[1]) if len(sys.argv) > 1 else 30, freq_out=100000)
torch.cuda.synchronize()
