"""The reference's own formula-level known answers, re-asserted on the oracle
(reference tests cited per test; relative to /root/reference/tests)."""
import math

import numpy as np
import torch

from oracle import mlp_ref, mcmc_ref, vi_ref
from oracle.mlp_ref import MLPSpec


def test_numpar_is_21():                      # test_mlp.py:45-54  MLP(2,1,(5,)).numpar()==21
    assert MLPSpec((2, 5, 1)).nparams == 21
    assert sum(p.numel() for p in mlp_ref.build_module(MLPSpec((2, 5, 1))).parameters()) == 21


def test_flatten_unflatten_roundtrip():       # test_nnwrap.py:22-38
    spec = MLPSpec((2, 8, 8, 1), "tanh")
    mod = mlp_ref.build_module(spec)
    w = np.concatenate([p.detach().flatten().numpy() for p in mod.parameters()])
    x = np.random.RandomState(0).rand(7, 2)
    y0 = mod(torch.tensor(x)).detach().numpy()
    mlp_ref.load_flat(mod, w)
    assert np.array_equal(mod(torch.tensor(x)).detach().numpy(), y0)
    assert np.array_equal(mlp_ref.forward_flat(mod, w, x), y0)


def test_gaussian_and_gmm_closed_forms():     # test_rvar.py:42-66, 89-102
    spec = MLPSpec((1, 1), "tanh", bias=False)      # a single parameter
    mu, rho = np.array([0.3]), np.array([math.log(0.7)])
    eps = np.array([[0.0]])
    x, y = np.zeros((1, 1)), np.zeros((1, 1))
    r = vi_ref.viloss(spec, mu, rho, eps, x, y, 1.0, 1, want_grad=False, pi=0.25, sigma1=0.5, sigma2=2.0)
    w = 0.3
    assert abs(r["log_q"] - (-0.5 * math.log(2 * math.pi) - math.log(0.7))) < 1e-12
    n = lambda v, s: math.exp(-v * v / (2 * s * s)) / (s * math.sqrt(2 * math.pi))
    assert abs(r["log_prior"] - math.log(0.25 * n(w, 0.5) + 0.75 * n(w, 2.0))) < 1e-12


def test_chain_shapes_alphas_and_maxpost():   # test_mcmc.py:73-90, 129-143
    mean = np.array([1.0, -1.0])
    lp = lambda x: -0.5 * float(np.sum((x - mean) ** 2))
    rng = np.random.RandomState(42)
    res = mcmc_ref.run_chain(lp, mcmc_ref.AmcmcState(gamma=0.5), 500, np.zeros(2), rng)
    assert res["chain"].shape == (501, 2) and res["logpost"].shape == (501,) and res["alphas"].shape == (501,)
    assert res["alphas"][0] == 0.0
    assert res["maxpost"] >= np.max(res["logpost"]) - 1e-15
    lp1 = lambda x: -0.5 * float(x[0] * x[0])   # test_mcmc.py:56-70: 1-D unit Gaussian, 2000 steps
    r1 = mcmc_ref.run_chain(lp1, mcmc_ref.AmcmcState(gamma=0.5), 2000, np.array([0.0]), np.random.RandomState(42))
    assert 0.05 < r1["accrate"] < 0.95


def test_map_near_gaussian_mean():            # test_mcmc.py:33-53, 93-109
    mean = np.array([1.0, -1.0])
    lp = lambda x: -0.5 * float(np.sum((x - mean) ** 2))
    lg = lambda x: -(x - mean)
    res = mcmc_ref.run_chain(lp, mcmc_ref.AmcmcState(gamma=0.5, t0=50, tadapt=100), 3000, np.zeros(2), np.random.RandomState(42))
    assert np.all(np.abs(res["mapparams"] - mean) < 0.5)
    res = mcmc_ref.run_chain(lp, mcmc_ref.HmcState(epsilon=0.1, L=5), 1000, np.zeros(2),
                             np.random.RandomState(42), logpostgrad=lg)
    assert np.all(np.abs(res["mapparams"] - mean) < 0.5)


def test_logpost_perfect_beats_bad():         # test_losses.py:10-93 (perfect target < bad target)
    spec = MLPSpec((1, 4, 1), "tanh")
    mod = mlp_ref.build_module(spec)
    w = 0.3 * np.random.RandomState(1).randn(spec.nparams)
    x = np.linspace(-1, 1, 9)[:, None]
    y = mlp_ref.forward_flat(mod, w, x)
    good = mlp_ref.logpost(mod, w, x, [v for v in y], 0.1)
    bad = mlp_ref.logpost(mod, w, x, [v + 1.0 for v in y], 0.1)
    assert np.isfinite(good) and good > bad
    assert abs(good - (-(9 / 2) * math.log(2 * math.pi) - 9 * math.log(0.1))) < 1e-9
