"""GPU: NN_MCMC through the HIP path reproduces the reference's chains.  Acceptance
sequences and chain states must be bit-exact in float64 for AMCMC (proposals do not depend
on the log-posterior value); log-posteriors agree to 1e-11; HMC/MALA states, which do depend
on device gradients, agree to 1e-9 over the fixture horizon with identical acceptance."""
import numpy as np
import pytest

from conftest import load_golden, spec_of
from quinn_amd.nns.mlp import MLP
from quinn_amd.solvers.nn_mcmc import NN_MCMC

pytestmark = pytest.mark.gpu


def _net(g):
    dims = [int(v) for v in g["dims"]]
    return MLP(dims[0], dims[-1], tuple(dims[1:-1]), activ=str(g["activ"]))


@pytest.mark.parametrize("name", ["g2_amcmc_0.npz", "g2_amcmc_1.npz", "g2_amcmc_cfg1.npz"])
def test_amcmc_chain_bit_exact(name):
    g = load_golden(name)
    solver = NN_MCMC(_net(g), verbose=False)
    np.random.seed(int(g["seed"]))
    solver.fit(g["x"], g["y"], zflag=False, datanoise=float(g["sigma"]), nmcmc=int(g["nmcmc"]), sampler='amcmc',
               sampler_params={'gamma': float(g["gamma"]), 't0': int(g["t0"]), 'tadapt': int(g["tadapt"])})
    # acceptance indices: bit-exact against the reference's run
    acc = (solver.samples[1:] != solver.samples[:-1]).any(axis=1)
    assert np.array_equal(acc, (g["chain"][1:] != g["chain"][:-1]).any(axis=1))
    assert solver.mcmc_results["accrate"] == float(g["accrate"])
    # chain states: the proposal uses the HOST's LAPACK SVD (numpy multivariate_normal), whose last
    # bits depend on the CPU the fixture was made on -> 1e-9 against the fixture, and bit-exact
    # against the oracle stepping the same chain on THIS host.
    np.testing.assert_allclose(solver.samples, g["chain"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(solver.mcmc_results["logpost"], g["logpost"], rtol=1e-9)
    from oracle import mlp_ref, mcmc_ref
    from conftest import spec_of
    spec = spec_of(g)
    mod = mlp_ref.build_module(spec)
    yd = [v for v in g["y"]]
    rng = np.random.RandomState(int(g["seed"]))
    ini = rng.rand(spec.nparams)
    ref = mcmc_ref.run_chain(lambda w: mlp_ref.logpost(mod, w, g["x"], yd, float(g["sigma"])),
                             mcmc_ref.AmcmcState(gamma=float(g["gamma"]), t0=int(g["t0"]), tadapt=int(g["tadapt"])),
                             int(g["nmcmc"]), ini, rng)
    assert np.array_equal(solver.samples, ref["chain"])
    assert np.array_equal(acc, ref["accepted"])
    assert np.array_equal(solver.cmode, ref["mapparams"])
    np.testing.assert_allclose(solver.mcmc_results["logpost"], ref["logpost"], rtol=1e-11)
    fin = np.isfinite(g["alphas"]) & (g["alphas"] < 1e300)
    np.testing.assert_allclose(solver.mcmc_results["alphas"][fin], g["alphas"][fin], rtol=1e-7, atol=1e-300)


def test_multichain_lockstep_equals_sequential_reference_runs():
    g = load_golden("g8_multichain.npz")
    C = int(g["nchains"])
    solver = NN_MCMC(_net(g), verbose=False)
    solver.fit(g["x"], g["y"], zflag=False, datanoise=float(g["sigma"]), nmcmc=int(g["nmcmc"]), sampler='amcmc',
               sampler_params={'gamma': float(g["gamma"]), 't0': int(g["t0"]), 'tadapt': int(g["tadapt"])},
               seeds=[int(g["seed0"]) + c for c in range(C)])
    assert solver.samples.shape == g["chain"].shape
    acc = (solver.samples[:, 1:] != solver.samples[:, :-1]).any(axis=2)
    assert np.array_equal(acc, (g["chain"][:, 1:] != g["chain"][:, :-1]).any(axis=2))
    np.testing.assert_allclose(solver.samples, g["chain"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(solver.mcmc_results["logpost"], g["logpost"], rtol=1e-9)
    assert np.array_equal(solver.mcmc_results["accrate"], g["accrate"])


@pytest.mark.parametrize("name,sampler", [("g3_hmc_0.npz", "hmc"), ("g3_hmc_1.npz", "hmc"), ("g3_mala.npz", "mala")])
def test_gradient_samplers(name, sampler):
    g = load_golden(name)
    solver = NN_MCMC(_net(g), verbose=False)
    sp = {'epsilon': float(g["epsilon"])}
    if sampler == "hmc":
        sp['L'] = int(g["L"])
    solver.fit(g["x"], g["y"], zflag=False, datanoise=float(g["sigma"]), nmcmc=int(g["nmcmc"]), sampler=sampler,
               sampler_params=sp, seeds=[int(g["seed"])])
    chain = solver.samples[0]
    acc = (chain[1:] != chain[:-1]).any(axis=1)
    assert np.array_equal(acc, (g["chain"][1:] != g["chain"][:-1]).any(axis=1))   # acceptance indices
    np.testing.assert_allclose(chain, g["chain"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(solver.mcmc_results["logpost"][0], g["logpost"], rtol=1e-9)


def test_predictions_from_chain():
    g = load_golden("g7_predict.npz")
    solver = NN_MCMC(_net(g), verbose=False)
    solver.samples = g["chain"]
    solver.cmode = g["chain"][-1]
    yens = solver.predict_ens(g["xg"], nens=int(g["nens"]), nburn=int(g["nburn"]))
    np.testing.assert_allclose(yens, g["yens"], rtol=1e-12, atol=1e-13)
    import functools
    solver.predict_ens = functools.partial(solver.predict_ens, nburn=int(g["nburn"]))
    solver._predict_ens_dev = functools.partial(solver._predict_ens_dev, nburn=int(g["nburn"]))   # (the moments stay on the device)
    ymean, yvar, ycov = solver.predict_mom_sample(g["xg"], msc=2, nsam=int(g["nens"]))
    np.testing.assert_allclose(ymean, g["ymean"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(yvar, g["yvar"], rtol=1e-9, atol=1e-15)
    np.testing.assert_allclose(ycov, g["ycov"], rtol=1e-9, atol=1e-15)
    m1, v1, c1 = solver.predict_mom_sample(g["xg"], msc=1, nsam=int(g["nens"]))                 # qn_pred_moments: mean + variance
    assert c1 is None
    np.testing.assert_allclose(m1, np.mean(g["yens"], axis=0), rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(v1, np.var(g["yens"], axis=0, ddof=1), rtol=1e-11, atol=1e-18)
    m0, v0, c0 = solver.predict_mom_sample(g["xg"], msc=0, nsam=int(g["nens"]))
    assert v0 is None and c0 is None and np.array_equal(m0, m1)


def test_zflag_bfgs_and_single_vector_api():
    """zflag=True path (scipy BFGS on the device log-posterior) and the reference's
    single-vector logpost / logpostgrad signatures."""
    g = load_golden("g1_logpost_0.npz")
    solver = NN_MCMC(_net(g), verbose=False)
    lpinfo = {'xd': g["x"], 'yd': [v for v in g["y"]], 'ltype': 'classical', 'lparams': {'sigma': float(g["sigma"])}}
    assert abs(solver.logpost(g["W"][0], lpinfo) - g["logpost"][0]) <= 1e-11 * abs(g["logpost"][0])
    gr = solver.logpostgrad(g["W"][0], lpinfo)
    assert gr.shape == (solver.pdim,)
    np.testing.assert_allclose(gr, g["grad"][0], rtol=1e-8, atol=1e-10 * np.abs(g["grad"][0]).max())
    from quinn_amd.nns.mlp import MLP
    small = NN_MCMC(MLP(1, 1, (5,), activ='tanh'), verbose=False)     # test_solvers.py:25-100 shape
    x = np.linspace(-1, 1, 20)[:, None]
    # the BFGS pre-fit starts the chain near a mode, far above a random start: with scipy's finite differences of the
    # device log-posterior (default, as the reference does) and with the device gradient as the jacobian (opt-in)
    for jac in (None, 'device'):
        np.random.seed(42)
        small.fit(x, np.sin(3 * x), zflag=True, datanoise=0.1, nmcmc=200, sampler='amcmc', sampler_params={}, bfgs_jac=jac)
        assert small.samples.shape == (201, small.pdim) and small.cmode.shape == (small.pdim,)
        lp0 = small.mcmc_results['logpost'][0]
        rnd = np.mean([small.logpost(np.random.RandomState(s).rand(small.pdim), small.lpinfo) for s in range(5)])
        assert lp0 > rnd + 10.0, (jac, lp0, rnd)
    # the operator cache follows the dataset OBJECTS: a new lpinfo with other data is evaluated on that data
    lp_a = small.logpost(small.cmode, small.lpinfo)
    other = {'xd': x.copy(), 'yd': [v for v in np.cos(3 * x)], 'ltype': 'classical', 'lparams': {'sigma': 0.1}}
    lp_b = small.logpost(small.cmode, other)
    assert lp_a != lp_b and small.logpost(small.cmode, small.lpinfo) == lp_a
    assert small.predict_ens(x, nens=5, nburn=100).shape == (5, 20, 1)
    assert small.predict_MAP(x).shape == (20, 1)
