#!/usr/bin/env python3
"""Residual network of examples/ex_ufit.py (RNet(3, 3, Poly(0)), 22 parameters): operator throughput for
many chains, and full-loop adaptive-Metropolis steps/s of NN_MCMC.fit (host engine) on the example's data size."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_default_dtype(torch.double)
from quinn_amd.nns.rnet import RNet, Poly
from quinn_amd.ops import MLPArch, BatchedMLP
from quinn_amd.solvers.nn_mcmc import NN_MCMC


def net(r=3, L=3, wp=None):
    return RNet(r, L, wp_function=wp or Poly(0), indim=1, outdim=1, layer_pre=True, layer_post=True)


def timeit(fn, n):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


out = {}
rs = np.random.RandomState(0)
for (r, L, B, N) in [(3, 3, 1, 13), (3, 3, 64, 4096), (3, 3, 4096, 4096), (16, 7, 256, 4096), (64, 3, 64, 4096)]:
    x = rs.rand(N, 1) * 2 * np.pi - np.pi
    y = np.sin(x) + 0.02 * rs.randn(N, 1)
    arch = MLPArch.from_module(net(r, L))
    op = BatchedMLP(arch, x, y)
    W = torch.as_tensor(0.3 * rs.randn(B, arch.nparams), device=op.device)
    tf = timeit(lambda: op.sse(W), 50)
    tg = timeit(lambda: op.sse_grad(W), 50)
    out[f"r{r}_L{L}_B{B}_N{N}"] = {"fwd_ms": 1e3 * tf, "fwd_evals_per_s": B / tf, "fwd_GFLOPs": arch.flops_fwd(N) * B / tf / 1e9,
                                   "grad_ms": 1e3 * tg, "grad_evals_per_s": B / tg,
                                   "grad_GFLOPs": arch.flops_fwdbwd(N) * B / tg / 1e9}
x = rs.rand(13, 1) * 2 * np.pi - np.pi
y = np.sin(x) + 0.02 * rs.randn(13, 1)
for nch in (1, 64):
    torch.manual_seed(0)
    uq = NN_MCMC(net(), verbose=False)
    kw = dict(zflag=False, datanoise=0.02, sampler='amcmc', sampler_params={'gamma': 0.01})
    if nch > 1:
        kw['seeds'] = list(range(nch))
    np.random.seed(0)
    uq.fit(x, y, nmcmc=50, **kw)
    np.random.seed(0)
    t0 = time.perf_counter()
    uq.fit(x, y, nmcmc=1500, **kw)
    el = time.perf_counter() - t0
    out[f"ex_ufit_amcmc_{nch}_chains"] = {"steps_per_s": 1500 / el, "chain_steps_per_s": 1500 * nch / el}
print(json.dumps(out, indent=1))
