import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from conftest import load_golden, spec_of
from oracle import mlp_ref
from quinn_amd.nns.mlp import MLP
from quinn_amd.solvers.nn_mcmc import NN_MCMC
g = load_golden("g2_amcmc_cfg1.npz")
dims=[int(v) for v in g["dims"]]
solver = NN_MCMC(MLP(dims[0],dims[-1],tuple(dims[1:-1]),activ=str(g["activ"])), verbose=False)
np.random.seed(int(g["seed"]))
solver.fit(g["x"], g["y"], zflag=False, datanoise=float(g["sigma"]), nmcmc=int(g["nmcmc"]), sampler='amcmc',
           sampler_params={'gamma': float(g["gamma"]), 't0': int(g["t0"]), 'tadapt': int(g["tadapt"])})
ch=solver.samples
bad=np.where((ch!=g["chain"]).any(axis=1))[0]
print("first differing step", bad[:5])
r=solver.mcmc_results
i=bad[0]
print("alphas mine/ref", r["alphas"][i-1:i+2], g["alphas"][i-1:i+2])
print("logpost mine/ref", r["logpost"][i-1:i+2], g["logpost"][i-1:i+2])
print("max rel lp diff before", np.max(np.abs(r["logpost"][:i]-g["logpost"][:i])/np.abs(g["logpost"][:i])))
print("maxabs state diff at i", np.abs(ch[i]-g["chain"][i]).max())
