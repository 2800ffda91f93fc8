#!/usr/bin/env python3
"""End-to-end HMC on the device engine: cfg2 shape (64 chains, 3x64, N=4096, L=3) and cfg5
(256 chains, 4x256, N=32768, L=10): steps/s and gradient evals/s; host engine next to it at cfg2."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quinn_amd.ops import MLPArch, BatchedMLP
from quinn_amd.mcmc.device_hmc import DeviceHMC
from quinn_amd.mcmc.hmc import HMC
from quinn_amd.ops import neg_log_post_from_sse

def data(N):
    rs = np.random.RandomState(0)
    x = rs.rand(N, 1) * 2 * np.pi - np.pi
    return x, 0.02 * rs.randn(N, 1) + np.sin(x)

out = {}
for name, dims, N, C, L, nsteps in [("cfg2", (1, 64, 64, 64, 1), 4096, 64, 3, 100), ("cfg5", (1, 256, 256, 256, 256, 1), 32768, 256, 10, 2)]:
    arch = MLPArch(dims, "tanh")
    x, y = data(N)
    op = BatchedMLP(arch, x, y)
    ini = np.stack([0.1 * np.random.RandomState(1000 + c).randn(arch.nparams) for c in range(C)])
    # cfg5: a step size at which the leapfrog is stable (acceptance ~1).  At 1e-4 (round-1 / early round-2 records) every
    # trajectory diverges: acceptance 0 and weights beyond 2^100, which the int8-slice kernels of the wide networks hand to
    # their plain-float64 rows (5x slower backward) -- a property of that step size, not of the sampler.
    eng = DeviceHMC(op, 0.02, epsilon=0.0005 if name == "cfg2" else 2e-6, L=L, seed=1)
    eng.run(2 if name == "cfg2" else 1, ini, store_chain=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = eng.run(nsteps, ini, store_chain=False)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    out[name] = {"chains": C, "L": L, "steps_per_s": nsteps / el, "grad_evals_per_s": nsteps * L * C / el,
                 "grad_tflops": nsteps * L * C * arch.flops_fwdbwd(N) / el / 1e12, "accrate": float(r["accrate"].mean())}
    if name == "cfg2":
        sig = 0.02
        lp = lambda W: -neg_log_post_from_sse(op.sse(W).cpu().numpy(), N, sig)
        lg = lambda W: -(0.5 * op.sse_grad(W)[1].double().cpu().numpy() / sig ** 2)
        mc = HMC(epsilon=0.0005, L=L)
        mc.setLogPostBatch(lp, lg)
        rngs = [np.random.RandomState(c) for c in range(C)]
        mc.run(2, ini, rngs=rngs, verbose=False)
        t0 = time.perf_counter()
        mc.run(20, ini, rngs=rngs, verbose=False)
        elh = time.perf_counter() - t0
        out["cfg2_host_engine"] = {"steps_per_s": 20 / elh, "grad_evals_per_s": 20 * (L + 1) * C / elh}
    del op
    torch.cuda.empty_cache()
print(json.dumps(out))
