#!/usr/bin/env python3
"""End-to-end HMC / MALA on the device engines: cfg2 shape (64 chains, 3x64, N=4096, HMC L=3 and MALA) and cfg5
(256 chains, 4x256, N=32768, L=10): steps/s and gradient evals/s; host engine next to it at cfg2.  The cfg2 step size is
searched (30-step runs) for an acceptance rate inside (0.2, 0.9): a chain that rejects every step never runs the
accept-side copies of state and gradient rows."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quinn_amd.ops import MLPArch, BatchedMLP
from quinn_amd.mcmc.device_hmc import DeviceHMC
from quinn_amd.mcmc.device_mala import DeviceMALA
from quinn_amd.mcmc.hmc import HMC
from quinn_amd.ops import neg_log_post_from_sse

def data(N):
    rs = np.random.RandomState(0)
    x = rs.rand(N, 1) * 2 * np.pi - np.pi
    return x, 0.02 * rs.randn(N, 1) + np.sin(x)

NG = int(os.environ.get("NGROUPS", "0")) or None          # chain groups on their own HIP streams (default: the engines')
out = {"groups": NG}
for name, dims, N, C, L, nsteps in [("cfg2", (1, 64, 64, 64, 1), 4096, 64, 3, 300), ("cfg5", (1, 256, 256, 256, 256, 1), 32768, 256, 10, 2)][:1 if os.environ.get("CFG2_ONLY") else 2]:
    arch = MLPArch(dims, "tanh")
    x, y = data(N)
    op = BatchedMLP(arch, x, y)
    ini = np.stack([0.1 * np.random.RandomState(1000 + c).randn(arch.nparams) for c in range(C)])
    # cfg5: a step size at which the leapfrog is stable (acceptance ~1).  At 1e-4 (round-1 / early round-2 records) every
    # trajectory diverges: acceptance 0 and weights beyond 2^100, which the int8-slice kernels of the wide networks hand to
    # their plain-float64 rows (5x slower backward) -- a property of that step size, not of the sampler.
    eps = 2e-6
    if name == "cfg2":
        for k in range(16):                                                   # largest step size with acceptance in (0.3, 0.85)
            eps = 4e-4 / 1.5 ** k
            acc = float(DeviceHMC(op, 0.02, epsilon=eps, L=L, seed=1).run(150, ini, store_chain=False)["accrate"].mean())
            if 0.3 < acc < 0.85:
                break
    eng = DeviceHMC(op, 0.02, epsilon=eps, L=L, seed=1, groups=NG)
    eng.run(60 if name == "cfg2" else 1, ini, store_chain=False)           # (warm-up; cfg2: lets the clock settle as well)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = eng.run(nsteps, ini, store_chain=False)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    out[name] = {"chains": C, "L": L, "epsilon": eps, "steps_per_s": nsteps / el, "grad_evals_per_s": nsteps * L * C / el,
                 "grad_tflops": nsteps * L * C * arch.flops_fwdbwd(N) / el / 1e12, "accrate": float(r["accrate"].mean())}
    if name == "cfg2":
        for k in range(20):
            epm = 4e-4 / 1.5 ** k
            acc = float(DeviceMALA(op, 0.02, epsilon=epm, seed=1).run(300, ini, store_chain=False)["accrate"].mean())
            if 0.3 < acc < 0.85:
                break
        em = DeviceMALA(op, 0.02, epsilon=epm, seed=1, groups=NG)
        em.run(60, ini, store_chain=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rm = em.run(3 * nsteps, ini, store_chain=False)
        torch.cuda.synchronize(); elm = time.perf_counter() - t0
        out["cfg2_mala"] = {"chains": C, "epsilon": epm, "steps_per_s": 3 * nsteps / elm, "grad_evals_per_s": 3 * nsteps * C / elm,
                            "grad_tflops": 3 * nsteps * C * arch.flops_fwdbwd(N) / elm / 1e12, "accrate": float(rm["accrate"].mean())}
    if name == "cfg2":
        sig = 0.02
        lp = lambda W: -neg_log_post_from_sse(op.sse(W).cpu().numpy(), N, sig)
        lg = lambda W: -(0.5 * op.sse_grad(W)[1].double().cpu().numpy() / sig ** 2)
        mc = HMC(epsilon=eps, L=L)
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
