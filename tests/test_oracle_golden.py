"""Pin the CPU oracle (oracle/) against fixtures produced by the imported reference
(tests/golden/gen_golden.py).  CPU only; no reference needed at run time."""
import numpy as np
import pytest
import torch

from conftest import load_golden, spec_of, assert_chain_matches_fixture
from oracle import mlp_ref, mcmc_ref, vi_ref, fit_ref


@pytest.mark.parametrize("ci", range(5))
def test_g1_logpost_and_grad_bitwise(ci):
    g = load_golden(f"g1_logpost_{ci}.npz")
    spec = spec_of(g)
    mod = mlp_ref.build_module(spec)
    assert spec.nparams == g["W"].shape[1]
    yd = [yy for yy in g["y"]]
    sigma = float(g["sigma"])
    for k, w in enumerate(g["W"]):
        assert mlp_ref.logpost(mod, w, g["x"], yd, sigma) == g["logpost"][k]
        assert np.array_equal(mlp_ref.logpostgrad(mod, w, g["x"], yd, sigma), g["grad"][k])
        assert np.array_equal(mlp_ref.forward_flat(mod, w, g["x"]), g["pred"][k])
        # the kernel-facing decomposition: logpost == scalar tail applied to the SSE
        s = mlp_ref.sse(mod, w, g["x"], g["y"])
        assert mlp_ref.logpost_from_sse(s, len(yd), sigma) == g["logpost"][k]


def _closures(g):
    spec = spec_of(g)
    mod = mlp_ref.build_module(spec)
    yd = [yy for yy in g["y"]]
    sigma = float(g["sigma"])
    lp = lambda w: mlp_ref.logpost(mod, w, g["x"], yd, sigma)
    lg = lambda w: mlp_ref.logpostgrad(mod, w, g["x"], yd, sigma)
    return spec, lp, lg


def _check_chain(res, g):
    assert_chain_matches_fixture(res, g)
    assert res["alphas"][0] == 0.0 and len(res["alphas"]) == int(g["nmcmc"]) + 1
    if not np.array_equal(res["chain"], g["chain"]):
        import warnings
        warnings.warn("chain equals the reference fixture to 1e-9 but not bitwise (different host LAPACK)")


@pytest.mark.parametrize("name", ["g2_amcmc_0.npz", "g2_amcmc_1.npz", "g2_amcmc_cfg1.npz"])
def test_g2_amcmc_chain_bitwise(name):
    g = load_golden(name)
    spec, lp, _ = _closures(g)
    rng = np.random.RandomState(int(g["seed"]))
    ini = rng.rand(spec.nparams)
    prop = mcmc_ref.AmcmcState(gamma=float(g["gamma"]), t0=int(g["t0"]), tadapt=int(g["tadapt"]))
    res = mcmc_ref.run_chain(lp, prop, int(g["nmcmc"]), ini, rng, record_uniforms=True)
    _check_chain(res, g)
    if "uniforms" in g:
        assert np.array_equal(res["uniforms"], g["uniforms"])
        acc = g["chain"][1:] != g["chain"][:-1]
        assert np.array_equal(res["accepted"], acc.any(axis=1))


@pytest.mark.parametrize("name", ["g3_hmc_0.npz", "g3_hmc_1.npz"])
def test_g3_hmc_chain_bitwise(name):
    g = load_golden(name)
    spec, lp, lg = _closures(g)
    rng = np.random.RandomState(int(g["seed"]))
    ini = rng.rand(spec.nparams)
    prop = mcmc_ref.HmcState(epsilon=float(g["epsilon"]), L=int(g["L"]))
    _check_chain(mcmc_ref.run_chain(lp, prop, int(g["nmcmc"]), ini, rng, logpostgrad=lg), g)


def test_g3_mala_chain_bitwise():
    g = load_golden("g3_mala.npz")
    spec, lp, lg = _closures(g)
    rng = np.random.RandomState(int(g["seed"]))
    ini = rng.rand(spec.nparams)
    prop = mcmc_ref.MalaState(epsilon=float(g["epsilon"]))
    _check_chain(mcmc_ref.run_chain(lp, prop, int(g["nmcmc"]), ini, rng, logpostgrad=lg), g)


def test_g8_multichain_definition():
    g = load_golden("g8_multichain.npz")
    spec, lp, _ = _closures(g)
    C = int(g["nchains"])
    res = mcmc_ref.run_multichain(
        lambda: lp, lambda: mcmc_ref.AmcmcState(gamma=float(g["gamma"]), t0=int(g["t0"]), tadapt=int(g["tadapt"])),
        int(g["nmcmc"]), spec.nparams, [int(g["seed0"]) + c for c in range(C)])
    for c in range(C):
        assert_chain_matches_fixture(res, g, c)


@pytest.mark.parametrize("ci", range(3))
def test_g4_viloss(ci):
    g = load_golden(f"g4_viloss_{ci}.npz")
    spec = spec_of(g)
    pr = dict(pi=float(g["prior"][0]), sigma1=float(g["prior"][1]), sigma2=float(g["prior"][2]))
    # init draws: mu ~ U, rho ~ U tensor by tensor from the seeded generator -- but the reference
    # first builds the MLP (consuming the generator); so only check the eps replay + values here
    r = vi_ref.viloss(spec, g["mu"], g["rho"], g["eps_elbo"], g["x"], g["y"], float(g["datanoise"]),
                      int(g["num_batches"]), want_grad=False, **pr)
    assert r["log_prior"] == float(g["elbo_log_prior"])
    assert r["log_q"] == float(g["elbo_log_q"])
    assert r["nll"] == float(g["elbo_nll"])
    r = vi_ref.viloss(spec, g["mu"], g["rho"], g["eps_loss"], g["x"], g["y"], float(g["datanoise"]),
                      int(g["num_batches"]), **pr)
    assert r["loss"] == float(g["loss"])
    # gradient accumulation order differs (flat leaf vs one Parameter per tensor): rounding only
    np.testing.assert_allclose(r["dmu"], g["dmu"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(r["drho"], g["drho"], rtol=1e-12, atol=1e-12)


def test_g5_vifit_trajectory():
    g = load_golden("g5_vifit.npz")
    spec = spec_of(g)
    gen = torch.Generator()
    gen.set_state(torch.from_numpy(g["gen_state"]))
    info = fit_ref.fit_vi(spec, g["mu0"], g["rho0"], g["x"], g["y"], g["xval"], g["yval"], int(g["nepochs"]),
                          int(g["batch_size"]), float(g["lrate"]), int(g["nsam"]), float(g["datanoise"]), gen)
    np.testing.assert_allclose(info["history"], g["history"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(info["final"][0], g["mu_final"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(info["final"][1], g["rho_final"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(info["best"][0], g["mu_best"], rtol=1e-9, atol=1e-11)
    assert info["best_epoch"] == int(g["best_epoch"])


def test_g5_variational_init_draw_order():
    g = load_golden("g5_vifit.npz")
    spec = spec_of(g)
    # reference: manual_seed -> MLP() init (consumes draws) -> BNet init.  Replay: seed, build the
    # same MLP through torch (same draws), then draw (mu, rho) in the oracle's order.
    gen = torch.Generator(); gen.manual_seed(int(g["torch_seed"]))
    torch.manual_seed(int(g["torch_seed"]))
    mod = mlp_ref.build_module(spec)          # consumes the global generator like MLP()
    gen.set_state(torch.get_rng_state())
    mu, rho = vi_ref.init_variational(spec, gen)
    assert np.array_equal(mu, g["mu0"]) and np.array_equal(rho, g["rho0"])
    w = np.concatenate([p.detach().flatten().numpy() for p in mod.parameters()])
    assert np.array_equal(w, g["w_net"])


def test_g6_ensemble_trajectories_bitwise():
    g = load_golden("g6_ens.npz")
    spec = spec_of(g)
    rng = np.random.RandomState(int(g["np_seed"]))
    gen = torch.Generator(); gen.manual_seed(int(g["torch_seed"]))
    members = fit_ref.fit_ensemble(spec, g["w0"], g["x"], g["y"], g["xval"], g["yval"], int(g["nens"]),
                                   float(g["dfrac"]), int(g["nepochs"]), int(g["batch_size"]),
                                   float(g["lrate"]), rng, gen)
    for j, m in enumerate(members):
        assert np.array_equal(m["history"], g["history"][j])
        assert np.array_equal(m["best"], g["best"][j])
        assert np.array_equal(m["final"], g["final"][j])
    # predict_ens: permuted members' predictions (nn_ens.py:100-108)
    prs = np.random.RandomState(int(g["predict_seed"]))
    order = prs.permutation(int(g["nens"]))
    mod = mlp_ref.build_module(spec)
    yens = np.array([mlp_ref.forward_flat(mod, members[k]["best"], g["xg"]) for k in order])
    assert np.array_equal(yens, g["yens"])


def test_g6_fullbatch_members_identical():
    g = load_golden("g6_ens_fullbatch.npz")
    # shared deepcopy => same trajectory up to the row order of each member's permuted full batch
    np.testing.assert_allclose(g["best"][0], g["best"][1], rtol=1e-9, atol=1e-12)
    spec = spec_of(g)
    rng = np.random.RandomState(int(g["np_seed"]))
    gen = torch.Generator(); gen.manual_seed(int(g["torch_seed"]))
    members = fit_ref.fit_ensemble(spec, g["w0"], g["x"], g["y"], g["x"].copy(), g["y"].copy(), 2, 1.0,
                                   int(g["nepochs"]), None, float(g["lrate"]), rng, gen)
    # dfrac=1 permutes the rows; MSE over a permuted full batch differs by summation order only
    for j in range(2):
        np.testing.assert_allclose(members[j]["best"], g["best"][j], rtol=1e-9, atol=1e-12)


def test_g7_prediction_from_chain():
    g = load_golden("g7_predict.npz")
    spec = spec_of(g)
    mod = mlp_ref.build_module(spec)
    chain, nens, nburn = g["chain"], int(g["nens"]), int(g["nburn"])
    nevery = int((chain.shape[0] - nburn) / nens)              # nn_mcmc.py:194
    yens = np.array([mlp_ref.forward_flat(mod, chain[nburn + j * nevery], g["xg"]) for j in range(nens)])
    assert np.array_equal(yens, g["yens"])
    assert np.array_equal(np.mean(yens, axis=0), g["ymean"])   # quinn.py:88
    cov = np.cov(yens[:, :, 0], rowvar=False, ddof=1)          # quinn.py:93-94
    assert np.array_equal(cov, g["ycov"][:, :, 0])
    assert np.array_equal(np.diag(cov), g["yvar"][:, 0])


def test_g11_ensemble_without_validation_set_bitwise():
    g = load_golden("g11_ens_noval.npz")
    spec = spec_of(g)
    rng = np.random.RandomState(int(g["np_seed"]))
    gen = torch.Generator(); gen.manual_seed(int(g["torch_seed"]))
    members = fit_ref.fit_ensemble(spec, g["w0"], g["x"], g["y"], None, None, int(g["nens"]), float(g["dfrac"]),
                                   int(g["nepochs"]), int(g["batch_size"]), float(g["lrate"]), rng, gen)
    for j, m in enumerate(members):
        assert np.array_equal(m["history"], g["history"][j])
        assert np.array_equal(m["best"], g["best"][j])
        assert np.array_equal(m["final"], g["final"][j])


def test_g9_rms_anchored_ensemble_bitwise():
    g = load_golden("g9_rms.npz")
    spec = spec_of(g)
    rng = np.random.RandomState(int(g["np_seed"]))
    gen = torch.Generator(); gen.manual_seed(int(g["torch_seed"]))
    members = fit_ref.fit_rms(spec, g["w0"], g["x"], g["y"], g["xval"], g["yval"], int(g["nens"]), float(g["dfrac"]),
                              int(g["nepochs"]), int(g["batch_size"]), float(g["lrate"]), rng, gen,
                              float(g["datanoise"]), float(g["priorsigma"]))
    for j, m in enumerate(members):
        assert np.array_equal(m["history"], g["history"][j])
        assert np.array_equal(m["best"], g["best"][j])
        assert np.array_equal(m["final"], g["final"][j])


# ---------------------------------------------------------------- G12: the shapes the build's DEFAULT (int8-slice) kernels take
def _check_g12_chain(res, g):
    """G12 fixtures hold the acceptance mask, every step of 256 strided columns and three full states (p = 8513)."""
    assert np.array_equal(res["accepted"], g["accepted"])                       # acceptance indices: bit for bit
    assert res["accrate"] == float(g["accrate"])
    assert np.array_equal(res["uniforms"], g["uniforms"])
    n = int(g["nmcmc"])
    for a, b in ((res["chain"][:, g["cols"]], g["chain_cols"]), (res["chain"][n // 2], g["chain_mid"]),
                 (res["chain"][-1], g["chain_final"]), (res["logpost"], g["logpost"]), (res["mapparams"], g["mapparams"])):
        assert np.array_equal(a, b)
    fin = np.isfinite(g["alphas"]) & (g["alphas"] < 1e300)
    np.testing.assert_allclose(res["alphas"][fin], g["alphas"][fin], rtol=1e-6, atol=1e-300)


@pytest.mark.parametrize("name", ["g12_hmc_0.npz", "g12_hmc_1.npz", "g12_mala.npz", "g13_relu_hmc.npz", "g13_relu_mala.npz"])
def test_g12_gradient_chains_3x64_bitwise(name):
    """(G13: the same on the reference's default activation, relu -- tests/golden/gen_golden.py::g13)"""
    g = load_golden(name)
    spec, lp, lg = _closures(g)
    assert spec.dims == (1, 64, 64, 64, 1) and spec.activ == ("relu" if "relu" in name else "tanh")
    rng = np.random.RandomState(int(g["seed"]))
    prop = mcmc_ref.MalaState(epsilon=float(g["epsilon"])) if "mala" in name else \
        mcmc_ref.HmcState(epsilon=float(g["epsilon"]), L=int(g["L"]))
    res = mcmc_ref.run_chain(lp, prop, int(g["nmcmc"]), g["param_ini"], rng, logpostgrad=lg, record_uniforms=True)
    assert 0 < res["accepted"].sum() < len(res["accepted"])                    # both outcomes occur
    _check_g12_chain(res, g)


def test_g12_amcmc_p1761_adaptation_fires():
    """Bit for bit on the host the fixture was made on.  From the first ADAPTED proposal on (step 20) the reference draws through
    numpy's SVD of a rank-deficient covariance (<= 21 distinct states + 1e-8 I in 1761 dimensions, admcmc.py:66-70); the basis
    LAPACK returns for the degenerate subspace is implementation-defined, so on another CPU the reference's own chain leaves the
    fixture there (profiles/r04_diag_g12_amcmc.txt): such a host is held to the fixture up to the first adaptation only."""
    g = load_golden("g12_amcmc.npz")
    spec, lp, _ = _closures(g)
    assert spec.dims == (1, 40, 40, 1)
    n, tadapt = int(g["nmcmc"]), int(g["tadapt"])
    rng = np.random.RandomState(int(g["seed"]))
    prop = mcmc_ref.AmcmcState(cov_ini=float(g["cov_ini_diag"]) * np.eye(spec.nparams), gamma=float(g["gamma"]),
                               t0=int(g["t0"]), tadapt=tadapt)
    res = mcmc_ref.run_chain(lp, prop, n, g["param_ini"], rng, record_uniforms=True)
    assert np.array_equal(res["uniforms"], g["uniforms"])
    acc = (g["chain"][1:] != g["chain"][:-1]).any(axis=1)
    assert 0 < acc[:tadapt].sum() < tadapt and 0 < acc[tadapt + 1:].sum() < n - tadapt - 1     # both outcomes, before and after adapting
    same = np.abs(res["chain"] - g["chain"]).max(axis=1) <= 1e-9 * (1 + np.abs(g["chain"]).max(axis=1))
    if same.all():
        _check_chain(res, g)
        assert np.array_equal(res["accepted"], acc)
    else:
        import warnings
        upto = int(np.flatnonzero(~same)[0])
        warnings.warn(f"this host's LAPACK leaves the fixture's adapted proposals at state {upto} (rank-deficient covariance)")
        assert upto > tadapt
        assert np.array_equal(res["chain"][:upto], g["chain"][:upto]) or np.allclose(res["chain"][:upto], g["chain"][:upto], rtol=1e-9, atol=1e-11)
        assert np.array_equal(res["accepted"][:upto - 1], acc[:upto - 1])
        np.testing.assert_allclose(res["logpost"][:upto], g["logpost"][:upto], rtol=1e-12)


def test_g13_relu_amcmc_p1761_before_adaptation():
    """Adaptive Metropolis on MLP(1,1,(40,40),'relu') with cov_ini = 1e-5 I, 50 steps, no adaptation (t0 = 100): bit for bit."""
    g = load_golden("g13_relu_amcmc.npz")
    spec, lp, _ = _closures(g)
    assert spec.dims == (1, 40, 40, 1) and spec.activ == "relu"
    prop = mcmc_ref.AmcmcState(cov_ini=float(g["cov_ini_diag"]) * np.eye(spec.nparams), gamma=float(g["gamma"]),
                               t0=int(g["t0"]), tadapt=int(g["tadapt"]))
    res = mcmc_ref.run_chain(lp, prop, int(g["nmcmc"]), g["param_ini"], np.random.RandomState(int(g["seed"])), record_uniforms=True)
    assert 0 < res["accepted"].sum() < len(res["accepted"])
    _check_g12_chain(res, g)


@pytest.mark.parametrize("name", ["g12_viloss.npz", "g13_relu_viloss.npz"])
def test_g12_viloss_2x128(name):
    g = load_golden(name)
    spec = spec_of(g)
    pr = dict(pi=float(g["prior"][0]), sigma1=float(g["prior"][1]), sigma2=float(g["prior"][2]))
    r = vi_ref.viloss(spec, g["mu"], g["rho"], g["eps_elbo"], g["x"], g["y"], float(g["datanoise"]), int(g["num_batches"]),
                      want_grad=False, **pr)
    assert r["log_prior"] == float(g["elbo_log_prior"]) and r["log_q"] == float(g["elbo_log_q"]) and r["nll"] == float(g["elbo_nll"])
    r = vi_ref.viloss(spec, g["mu"], g["rho"], g["eps_loss"], g["x"], g["y"], float(g["datanoise"]), int(g["num_batches"]), **pr)
    assert r["loss"] == float(g["loss"])
    np.testing.assert_allclose(r["dmu"], g["dmu"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(r["drho"], g["drho"], rtol=1e-12, atol=1e-12)
