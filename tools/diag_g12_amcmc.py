"""Diagnostic (round 4): where does the G12 adaptive-Metropolis chain (p = 1761, rank-deficient adapted covariance) of the build
leave the reference's fixture on THIS host, and does the CPU oracle stepped on this host leave it at the same step?"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, spec_of           # noqa: E402
from oracle import mcmc_ref, mlp_ref                 # noqa: E402

g = load_golden("g12_amcmc.npz")
spec = spec_of(g)
mod = mlp_ref.build_module(spec)
yd = [v for v in g["y"]]
n = int(g["nmcmc"])
t = time.time()
ref = mcmc_ref.run_chain(lambda w: mlp_ref.logpost(mod, w, g["x"], yd, float(g["sigma"])),
                         mcmc_ref.AmcmcState(cov_ini=float(g["cov_ini_diag"]) * np.eye(spec.nparams), gamma=float(g["gamma"]),
                                             t0=int(g["t0"]), tadapt=int(g["tadapt"])), n, g["param_ini"],
                         np.random.RandomState(int(g["seed"])))
print("oracle on this host: %.0f s" % (time.time() - t), flush=True)
facc = (g["chain"][1:] != g["chain"][:-1]).any(axis=1)
d = np.abs(ref["chain"] - g["chain"]).max(axis=1)
print("oracle vs fixture: acceptance differs at", np.flatnonzero(ref["accepted"] != facc), " first state difference at step",
      (np.flatnonzero(d > 0)[:1]), " max |dx| per step (20..26):", d[20:27])
if "--cpu" in sys.argv:
    sys.exit(0)
from quinn_amd.nns.mlp import MLP                    # noqa: E402
from quinn_amd.solvers.nn_mcmc import NN_MCMC        # noqa: E402
for kernels in ("auto", "float64"):
    solver = NN_MCMC(MLP(1, 1, (40, 40), activ="tanh"), verbose=False, kernels=kernels)
    solver.fit(g["x"], g["y"], zflag=False, datanoise=float(g["sigma"]), nmcmc=n, sampler='amcmc', param_ini=g["param_ini"],
               seeds=[int(g["seed"])], sampler_params={'cov_ini': float(g["cov_ini_diag"]) * np.eye(spec.nparams),
                                                       'gamma': float(g["gamma"]), 't0': int(g["t0"]), 'tadapt': int(g["tadapt"])})
    chain = solver.samples[0]
    acc = (chain[1:] != chain[:-1]).any(axis=1)
    print(kernels, "build vs fixture: acceptance differs at", np.flatnonzero(acc != facc), "| build vs oracle-on-this-host: acceptance differs at",
          np.flatnonzero(acc != ref["accepted"]), "states equal bitwise:", np.array_equal(chain, ref["chain"]),
          "max |logpost rel diff|", np.max(np.abs(solver.mcmc_results["logpost"][0] / ref["logpost"] - 1)))
