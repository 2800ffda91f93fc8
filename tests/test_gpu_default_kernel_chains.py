"""GPU: reference-generated chains (fixtures G12, tests/golden/gen_golden.py::g12) whose log-posterior / gradient evaluations
run through the DEFAULT kernels of their shape -- the sliced int8-product kernels (k_fused_fwd_i8 / k_fused_bwd_i8 for the 64-wide
network, the zero-padded 64-wide twin for the 40-wide one, k_i8_wide_* / k_i8_dw for the 128-wide one) -- and, as the second arm,
through the plain float64 kernels (`kernels='float64'`).  Every case first asserts WHICH arithmetic the operator dispatches
(qn_mlp_arith), so a fixture cannot silently fall to another kernel family.

Bars: acceptance indices bit-exact against the reference's run (quinn/mcmc/mcmc.py:65-85 accept test; proposals of
quinn/mcmc/hmc.py:27-70, mala.py:24-53, admcmc.py:38-74); HMC / MALA states 1e-9 (they integrate device gradients), AMCMC states
bit-exact against the oracle stepped on this host (its proposals do not depend on log-posterior values); log-posteriors 1e-9.
Plus one live-oracle HMC chain at the full configs[1] size (N = 4096)."""
import numpy as np
import pytest

from conftest import load_golden, spec_of
from quinn_amd import _lib
from quinn_amd.nns.mlp import MLP
from quinn_amd.solvers.nn_mcmc import NN_MCMC

pytestmark = pytest.mark.gpu

ARMS = [("auto", _lib.ARITH_I8_FUSED), ("float64", _lib.ARITH_PLAIN)]


def _net(g):
    dims = [int(v) for v in g["dims"]]
    return MLP(dims[0], dims[-1], tuple(dims[1:-1]), activ=str(g["activ"]))


def _assert_arith(solver, want, grad):
    op = solver._operator(solver.lpinfo)
    assert op.path(1, op.N, False) == _lib.PATH_FUSED and op.arith(1, op.N, False) == want
    if grad:
        assert op.path(1, op.N, True) == _lib.PATH_FUSED and op.arith(1, op.N, True) == want


@pytest.mark.parametrize("kernels,arith", ARMS)
@pytest.mark.parametrize("name", ["g12_hmc_0.npz", "g12_hmc_1.npz", "g12_mala.npz", "g13_relu_hmc.npz", "g13_relu_mala.npz"])
def test_g12_gradient_chains_on_3x64(name, kernels, arith):
    """(g13_relu_*: the reference's default activation; int8 slices with per-row activation scales since round 4)"""
    g = load_golden(name)
    assert str(g["activ"]) == ("relu" if "relu" in name else "tanh")
    assert tuple(int(v) for v in g["dims"]) == (1, 64, 64, 64, 1)
    solver = NN_MCMC(_net(g), verbose=False, kernels=kernels)
    sampler = "mala" if "mala" in name else "hmc"
    sp = {'epsilon': float(g["epsilon"])}
    if sampler == "hmc":
        sp['L'] = int(g["L"])
    n = int(g["nmcmc"])
    solver.fit(g["x"], g["y"], zflag=False, datanoise=float(g["sigma"]), nmcmc=n, sampler=sampler, sampler_params=sp,
               param_ini=g["param_ini"], seeds=[int(g["seed"])])
    _assert_arith(solver, arith, grad=True)
    chain = solver.samples[0]
    acc = (chain[1:] != chain[:-1]).any(axis=1)
    assert 0 < g["accepted"].sum() < n                                         # the fixture holds both outcomes
    assert np.array_equal(acc, g["accepted"]), np.flatnonzero(acc != g["accepted"])      # acceptance indices: bit-exact
    assert solver.mcmc_results["accrate"][0] == float(g["accrate"])
    np.testing.assert_allclose(chain[:, g["cols"]], g["chain_cols"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(chain[n // 2], g["chain_mid"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(chain[-1], g["chain_final"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(solver.mcmc_results["logpost"][0], g["logpost"], rtol=1e-9)
    np.testing.assert_allclose(solver.cmode[0], g["mapparams"], rtol=1e-9, atol=1e-9)
    # how close each accept test came to flipping: margin = |log u - log mh| of the reference's run, and the build's own mh
    fin = np.isfinite(g["alphas"][1:]) & (g["alphas"][1:] > 0)
    margin = np.abs(np.log(g["uniforms"][fin]) - np.log(g["alphas"][1:][fin]))
    mine = solver.mcmc_results["alphas"][0][1:][fin]
    drift = np.abs(np.log(mine) - np.log(g["alphas"][1:][fin]))
    print(f"{name} kernels={kernels}: closest accept test {margin.min():.3e} (log units), largest |dlog mh| of the build {drift.max():.3e}")
    assert drift.max() < 0.01 * margin.min() or drift.max() < 1e-6


_ORACLE_AMCMC = {}


def _oracle_amcmc_on_this_host(g, spec):
    """The CPU oracle stepping the G12 adaptive-Metropolis chain on THIS host (once per session: ~70 SVDs of 1761 x 1761)."""
    if "res" not in _ORACLE_AMCMC:
        from oracle import mcmc_ref, mlp_ref
        mod = mlp_ref.build_module(spec)
        yd = [v for v in g["y"]]
        _ORACLE_AMCMC["res"] = mcmc_ref.run_chain(
            lambda w: mlp_ref.logpost(mod, w, g["x"], yd, float(g["sigma"])),
            mcmc_ref.AmcmcState(cov_ini=float(g["cov_ini_diag"]) * np.eye(spec.nparams), gamma=float(g["gamma"]), t0=int(g["t0"]),
                                tadapt=int(g["tadapt"])), int(g["nmcmc"]), g["param_ini"], np.random.RandomState(int(g["seed"])))
    return _ORACLE_AMCMC["res"]


@pytest.mark.parametrize("kernels,arith", ARMS)
def test_g12_amcmc_on_padded_40_wide(kernels, arith):
    """p = 1761, adaptation at steps 20, 40, 60.  Before the first adaptation the proposals are host-independent (cov_ini is a
    multiple of the identity): acceptance indices and states are held against the REFERENCE's fixture bit for bit / to 1e-9.
    From the first adapted proposal on, the reference draws through numpy's SVD of a RANK-DEFICIENT covariance (<= 21 distinct
    states + 1e-8 I in 1761 dimensions, admcmc.py:66-70): the basis LAPACK returns for the 1740-dimensional degenerate subspace
    is implementation-defined, and the reference's own chain differs between hosts from there (measured: the oracle stepped on
    the GPU box's EPYC 9575F leaves the fixture made on the build container's CPU at step 22, |dx| = 5e-5, acceptance index
    24 -- tools/diag_g12_amcmc.py, profiles/r04_diag_g12_amcmc.txt).  So the whole chain -- all three adaptations -- is held
    bit for bit against the oracle stepped on THIS host (which test_oracle_golden.py pins bit for bit to the fixture on the
    host the fixture was made on), and against the fixture for as long as this host's oracle itself stays on it."""
    from threadpoolctl import threadpool_limits
    g = load_golden("g12_amcmc.npz")
    spec = spec_of(g)
    assert spec.dims == (1, 40, 40, 1)
    n, tadapt = int(g["nmcmc"]), int(g["tadapt"])
    with threadpool_limits(limits=16):                # (same LAPACK threading for the build's host engine and the oracle)
        solver = NN_MCMC(_net(g), verbose=False, kernels=kernels)
        solver.fit(g["x"], g["y"], zflag=False, datanoise=float(g["sigma"]), nmcmc=n, sampler='amcmc', param_ini=g["param_ini"],
                   seeds=[int(g["seed"])], sampler_params={'cov_ini': float(g["cov_ini_diag"]) * np.eye(spec.nparams),
                                                           'gamma': float(g["gamma"]), 't0': int(g["t0"]), 'tadapt': tadapt})
        ref = _oracle_amcmc_on_this_host(g, spec)
    _assert_arith(solver, arith, grad=False)
    chain = solver.samples[0]
    acc = (chain[1:] != chain[:-1]).any(axis=1)
    fix_acc = (g["chain"][1:] != g["chain"][:-1]).any(axis=1)
    assert 0 < fix_acc[:tadapt].sum() < tadapt and 0 < fix_acc[tadapt + 1:].sum() < n - tadapt - 1   # both outcomes, before and after
    # (1) the whole chain against the oracle on this host: acceptance indices and states bit for bit
    assert np.array_equal(acc, ref["accepted"]), np.flatnonzero(acc != ref["accepted"])
    assert np.array_equal(chain, ref["chain"])
    assert np.array_equal(solver.cmode[0], ref["mapparams"])
    np.testing.assert_allclose(solver.mcmc_results["logpost"][0], ref["logpost"], rtol=1e-11)
    assert solver.mcmc_results["accrate"][0] == ref["accrate"]
    # (2) against the reference's fixture, for as long as this host's LAPACK reproduces the fixture's proposals
    same = np.abs(ref["chain"] - g["chain"]).max(axis=1) <= 1e-9 * (1 + np.abs(g["chain"]).max(axis=1))
    upto = n + 1 if same.all() else int(np.flatnonzero(~same)[0])           # first state where the host's oracle left the fixture
    assert upto > tadapt, f"the oracle on this host leaves the fixture at state {upto}, before the first adaptation"
    print(f"g12_amcmc kernels={kernels}: this host's oracle reproduces the reference fixture for states 0..{upto - 1} of {n}")
    assert np.array_equal(acc[:upto - 1], fix_acc[:upto - 1])               # acceptance indices vs the reference: bit-exact
    np.testing.assert_allclose(chain[:upto], g["chain"][:upto], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(solver.mcmc_results["logpost"][0][:upto], g["logpost"][:upto], rtol=1e-9)
    if upto == n + 1:
        assert solver.mcmc_results["accrate"][0] == float(g["accrate"])


@pytest.mark.parametrize("kernels,arith", ARMS)
def test_g13_relu_amcmc_on_padded_40_wide(kernels, arith):
    """Adaptive Metropolis before its first adaptation on MLP(1,1,(40,40),'relu') (zero-padded 64-wide twin; relu on the
    int8-slice forward): cov_ini is a multiple of the identity, so the proposals are host-independent -- acceptance indices
    bit-exact against the reference's fixture, states bit for bit (they are the proposals), log-posteriors 1e-9."""
    g = load_golden("g13_relu_amcmc.npz")
    spec = spec_of(g)
    assert spec.dims == (1, 40, 40, 1) and str(g["activ"]) == "relu"
    n = int(g["nmcmc"])
    solver = NN_MCMC(_net(g), verbose=False, kernels=kernels)
    solver.fit(g["x"], g["y"], zflag=False, datanoise=float(g["sigma"]), nmcmc=n, sampler='amcmc', param_ini=g["param_ini"],
               seeds=[int(g["seed"])], sampler_params={'cov_ini': float(g["cov_ini_diag"]) * np.eye(spec.nparams),
                                                       'gamma': float(g["gamma"]), 't0': int(g["t0"]), 'tadapt': int(g["tadapt"])})
    _assert_arith(solver, arith, grad=False)
    chain = solver.samples[0]
    acc = (chain[1:] != chain[:-1]).any(axis=1)
    assert 0 < g["accepted"].sum() < n
    assert np.array_equal(acc, g["accepted"]), np.flatnonzero(acc != g["accepted"])
    assert solver.mcmc_results["accrate"][0] == float(g["accrate"])
    np.testing.assert_allclose(chain[:, g["cols"]], g["chain_cols"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(chain[-1], g["chain_final"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(solver.mcmc_results["logpost"][0], g["logpost"], rtol=1e-9)


@pytest.mark.parametrize("name", ["g12_viloss.npz", "g13_relu_viloss.npz"])
def test_g12_viloss_on_2x128(name):
    import torch
    from quinn_amd.vi.bnet import BNet
    g = load_golden(name)
    assert str(g["activ"]) == ("relu" if "relu" in name else "tanh")
    for kernels, want in (("auto", _lib.ARITH_I8_WIDE), ("float64", _lib.ARITH_PLAIN)):
        bm = BNet(_net(g), pi=float(g["prior"][0]), sigma1=float(g["prior"][1]), sigma2=float(g["prior"][2]))
        with torch.no_grad():
            bm.theta.copy_(torch.as_tensor(np.concatenate([g["mu"], g["rho"]]), device=bm.theta.device))
        S = int(g["nsam"])
        feed = [g["eps_elbo"], g["eps_loss"]]
        bm._draw_eps = lambda n: torch.as_tensor(feed.pop(0), device=bm.device)
        if kernels == "float64":
            bm.op.use_exact_float64()
        lp, lq, nll = bm.sample_elbo(g["x"], g["y"], S, likparams=[float(g["datanoise"])])
        assert bm.op.N == len(g["x"]) and bm.op.arith(S, bm.op.N, True) == want and bm.op.arith(S, bm.op.N, False) == want
        assert abs(lp.item() - float(g["elbo_log_prior"])) <= 1e-12 * abs(float(g["elbo_log_prior"]))
        assert abs(lq.item() - float(g["elbo_log_q"])) <= 1e-12 * abs(float(g["elbo_log_q"]))
        assert abs(nll.item() - float(g["elbo_nll"])) <= 1e-11 * abs(float(g["elbo_nll"]))
        bm.loss_params = [float(g["datanoise"]), S, int(g["num_batches"])]
        loss = bm.viloss(g["x"], g["y"])
        assert abs(loss.item() - float(g["loss"])) <= 1e-11 * abs(float(g["loss"]))
        loss.backward()
        gr = bm.theta.grad.cpu().numpy()
        p = bm.p
        sc = max(np.abs(g["dmu"]).max(), np.abs(g["drho"]).max())
        e = max(np.abs(gr[:p] - g["dmu"]).max(), np.abs(gr[p:] - g["drho"]).max()) / sc
        print(f"{name} kernels={kernels}: max gradient error / max|g| = {e:.2e}")
        assert e <= 1e-10


@pytest.mark.parametrize("kernels,arith", ARMS)
def test_live_oracle_hmc_chain_at_full_cfg2_size(kernels, arith):
    """configs[1] network and data size (3x64 tanh, N = 4096): 4 HMC chains x 20 steps (L = 3) by the build (host engine, device
    log-posterior / gradient kernels) against oracle/mcmc_ref.run_chain stepping the same chains on this host's CPU in float64
    (quinn/mcmc/mcmc.py:65-85, hmc.py:27-70): acceptance indices bit-exact, states 1e-9, log-posteriors 1e-10."""
    from oracle import mcmc_ref, mlp_ref
    N, C, nmcmc, L, eps, sigma = 4096, 4, 20, 3, 4e-4, 0.1
    x, y = mlp_ref.synthetic_data(N, 1, 0.02, seed=0)
    dims = (1, 64, 64, 64, 1)
    solver = NN_MCMC(MLP(1, 1, (64, 64, 64), activ='tanh'), verbose=False, kernels=kernels)
    seeds = [300 + c for c in range(C)]
    inis = np.stack([0.1 * np.random.RandomState(1000 + c).randn(solver.pdim) for c in range(C)])
    solver.fit(x, y, zflag=False, datanoise=sigma, nmcmc=nmcmc, sampler='hmc', sampler_params={'epsilon': eps, 'L': L},
               param_ini=inis, seeds=seeds)
    _assert_arith(solver, arith, grad=True)
    mod = mlp_ref.build_module(mlp_ref.MLPSpec(dims, "tanh"))
    yd = [v for v in y]
    nacc = 0
    for c in range(C):
        ref = mcmc_ref.run_chain(lambda w: mlp_ref.logpost(mod, w, x, yd, sigma), mcmc_ref.HmcState(epsilon=eps, L=L), nmcmc,
                                 inis[c], np.random.RandomState(seeds[c]),
                                 logpostgrad=lambda w: mlp_ref.logpostgrad(mod, w, x, yd, sigma))
        chain = solver.samples[c]
        acc = (chain[1:] != chain[:-1]).any(axis=1)
        assert np.array_equal(acc, ref["accepted"]), (c, np.flatnonzero(acc != ref["accepted"]))
        np.testing.assert_allclose(chain, ref["chain"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(solver.mcmc_results["logpost"][c], ref["logpost"], rtol=1e-10)
        nacc += int(ref["accepted"].sum())
    print(f"live-oracle HMC at cfg2 size, kernels={kernels}: {nacc} of {C * nmcmc} proposals accepted")
    assert 0 < nacc < C * nmcmc
