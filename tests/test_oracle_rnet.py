"""Pin the oracle's residual-network restatement (oracle/rnet_ref.py) against fixtures produced by
the imported reference RNet (tests/golden/gen_golden.py, group `rnet`), and the host-side mirror of
the RNet module (quinn_amd/nns/rnet.py) against the same fixtures.  CPU only."""
import numpy as np
import pytest
import torch

from conftest import load_golden, spec_of, assert_chain_matches_fixture
from oracle import mlp_ref, mcmc_ref, vi_ref, fit_ref

NCASES = 6


@pytest.mark.parametrize("ci", range(NCASES))
def test_g10_logpost_grad_pred_bitwise(ci):
    g = load_golden(f"g10_rnet_logpost_{ci}.npz")
    spec = spec_of(g)
    mod = mlp_ref.build_module(spec)
    assert spec.nparams == g["W"].shape[1]
    yd = [yy for yy in g["y"]]
    sigma = float(g["sigma"])
    for k, w in enumerate(g["W"]):
        assert mlp_ref.logpost(mod, w, g["x"], yd, sigma) == g["logpost"][k]
        assert np.array_equal(mlp_ref.logpostgrad(mod, w, g["x"], yd, sigma), g["grad"][k])
        assert np.array_equal(mlp_ref.forward_flat(mod, w, g["x"]), g["pred"][k])


def _closures(g):
    spec = spec_of(g)
    mod = mlp_ref.build_module(spec)
    yd = [yy for yy in g["y"]]
    sigma = float(g["sigma"])
    return (spec, lambda w: mlp_ref.logpost(mod, w, g["x"], yd, sigma),
            lambda w: mlp_ref.logpostgrad(mod, w, g["x"], yd, sigma))


def test_g10_amcmc_chain():
    g = load_golden("g10_rnet_amcmc.npz")
    spec, lp, _ = _closures(g)
    rng = np.random.RandomState(int(g["seed"]))
    ini = rng.rand(spec.nparams)
    prop = mcmc_ref.AmcmcState(gamma=float(g["gamma"]), t0=int(g["t0"]), tadapt=int(g["tadapt"]))
    res = mcmc_ref.run_chain(lp, prop, int(g["nmcmc"]), ini, rng, record_uniforms=True)
    assert_chain_matches_fixture(res, g)
    assert np.array_equal(res["uniforms"], g["uniforms"])


def test_g10_hmc_chain():
    g = load_golden("g10_rnet_hmc.npz")
    spec, lp, lg = _closures(g)
    rng = np.random.RandomState(int(g["seed"]))
    ini = rng.rand(spec.nparams)
    prop = mcmc_ref.HmcState(epsilon=float(g["epsilon"]), L=int(g["L"]))
    assert_chain_matches_fixture(mcmc_ref.run_chain(lp, prop, int(g["nmcmc"]), ini, rng, logpostgrad=lg), g)


def test_g10_ensemble_trajectories_bitwise():
    g = load_golden("g10_rnet_ens.npz")
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
    order = np.random.RandomState(int(g["predict_seed"])).permutation(int(g["nens"]))
    mod = mlp_ref.build_module(spec)
    yens = np.array([mlp_ref.forward_flat(mod, members[k]["best"], g["xg"]) for k in order])
    assert np.array_equal(yens, g["yens"])


def test_g10_vifit_trajectory():
    g = load_golden("g10_rnet_vifit.npz")
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


# ------------------------------------------------------------------ host-side mirror of the module
def _mirror(g):
    from quinn_amd.nns import rnet as R
    kind, arg = str(g["wp_kind"]), int(g["wp_arg"])
    wp = {"const": R.Const, "lin": R.Lin, "quad": R.Quad, "cubic": R.Cubic}.get(kind)
    wp = wp() if wp else R.Poly(arg) if kind == "poly" else (R.NonPar(arg) if arg else None)
    return R.RNet(int(g["rdim"]), int(g["nlayers"]), wp_function=wp, indim=int(g["indim"]) or None,
                  outdim=int(g["outdim"]) or None, biasorno=bool(g["biasorno"]), nonlin=bool(g["nonlin"]),
                  mlp=bool(g["mlp"]), layer_pre=bool(g["layer_pre"]), layer_post=bool(g["layer_post"]))


@pytest.mark.parametrize("ci", range(NCASES))
def test_mirror_module_init_layout_and_forward(ci):
    """Seeded construction draws the reference's initial weights; parameters() order is the
    fixture's flat layout; the arch descriptor carries the right coefficients."""
    from quinn_amd.ops import RNetArch, MLPArch, flatten_module
    g = load_golden(f"g10_rnet_logpost_{ci}.npz")
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.double)              # the reference sets this at import (tchutils.py:9)
    try:
        torch.manual_seed(int(g["torch_seed"]))
        net = _mirror(g)
        assert np.array_equal(flatten_module(net), g["w_init"])
        arch = MLPArch.from_module(net)
        assert isinstance(arch, RNetArch) and arch.nparams == g["W"].shape[1]
        spec = spec_of(g)
        assert arch.param_shapes() == spec.param_shapes()
        # coefficient table == the oracle's layer_weight on unit parameters
        from oracle.rnet_ref import layer_weight
        h = 1.0 / (spec.nlayers + 1.0)
        for i in range(arch.nsteps):
            for k in range(arch.npar):
                e = [1.0 if q == k else 0.0 for q in range(arch.npar)]
                assert arch.coef[i][k] == layer_weight(spec.wp_kind, spec.npar, e, h * i)
        # torch forward of the mirror module == fixture predictions
        off = 0
        with torch.no_grad():
            for p in net.parameters():
                p.copy_(torch.from_numpy(g["W"][0][off:off + p.numel()]).view_as(p)); off += p.numel()
            assert np.array_equal(net(torch.from_numpy(g["x"])).numpy(), g["pred"][0])
    finally:
        torch.set_default_dtype(old)
