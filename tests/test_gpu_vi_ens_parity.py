"""GPU: VI (BNet / NN_VI) and deep-ensemble (NN_Ens / nnfit) paths against the reference's
fixtures and the live oracle.  float64; tolerances stated per assertion."""
import numpy as np
import pytest
import torch

from conftest import load_golden, spec_of
from oracle import fit_ref, mlp_ref, vi_ref
from quinn_amd.nns.mlp import MLP
from quinn_amd.nns.nnfit import load_flat_into, nnfit
from quinn_amd.solvers.nn_ens import NN_Ens
from quinn_amd.solvers.nn_vi import NN_VI
from quinn_amd.vi.bnet import BNet

pytestmark = pytest.mark.gpu


def _net(g):
    dims = [int(v) for v in g["dims"]]
    return MLP(dims[0], dims[-1], tuple(dims[1:-1]), activ=str(g["activ"]))


def _set_theta(bm, mu, rho):
    with torch.no_grad():
        bm.theta.copy_(torch.as_tensor(np.concatenate([mu, rho]), device=bm.theta.device))


@pytest.mark.parametrize("ci", range(3))
def test_g4_viloss_terms_and_gradients(ci):
    g = load_golden(f"g4_viloss_{ci}.npz")
    bm = BNet(_net(g), pi=float(g["prior"][0]), sigma1=float(g["prior"][1]), sigma2=float(g["prior"][2]))
    _set_theta(bm, g["mu"], g["rho"])
    S = int(g["nsam"])
    feed = [g["eps_elbo"], g["eps_loss"]]
    bm._draw_eps = lambda n: torch.as_tensor(feed.pop(0), device=bm.device)
    lp, lq, nll = bm.sample_elbo(g["x"], g["y"], S, likparams=[float(g["datanoise"])])
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
    assert np.abs(gr[:p] - g["dmu"]).max() <= 1e-10 * sc
    assert np.abs(gr[p:] - g["drho"]).max() <= 1e-10 * sc


def test_g5_vi_fit_trajectory():
    g = load_golden("g5_vifit.npz")
    vi = NN_VI(_net(g), verbose=False)
    _set_theta(vi.bmodel, g["mu0"], g["rho0"])
    torch.set_rng_state(torch.from_numpy(g["gen_state"]))
    vi.fit(g["x"], g["y"], val=[g["xval"], g["yval"]], datanoise=float(g["datanoise"]), lrate=float(g["lrate"]),
           batch_size=int(g["batch_size"]), nsam=int(g["nsam"]), nepochs=int(g["nepochs"]), freq_out=1000)
    hist = np.array(vi.fit_info["history"])
    np.testing.assert_allclose(hist, g["history"], rtol=1e-8, atol=1e-9)
    p = vi.bmodel.p
    th = vi.bmodel.theta.detach().cpu().numpy()
    np.testing.assert_allclose(th[:p], g["mu_final"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(th[p:], g["rho_final"], rtol=1e-8, atol=1e-10)
    tb = vi.best_model.theta.detach().cpu().numpy()
    np.testing.assert_allclose(tb[:p], g["mu_best"], rtol=1e-8, atol=1e-10)
    assert vi.fit_info["best_epoch"] == int(g["best_epoch"])
    y = vi.predict_sample(g["xval"])
    assert y.shape == g["yval"].shape and np.isfinite(y).all()
    vi.nens = 7
    ye = vi.predict_ens(g["xval"])
    assert ye.shape == (7,) + g["yval"].shape and np.mean(np.var(ye, axis=0)) > 0      # test_vi.py:73-132


def test_bnet_init_matches_reference_draw_order():
    g = load_golden("g5_vifit.npz")
    torch.manual_seed(int(g["torch_seed"]))
    net = _net(g)                                     # consumes the generator like the reference's MLP()
    bm = BNet(net)
    np.testing.assert_array_equal(bm.mu.cpu().numpy(), g["mu0"])
    np.testing.assert_array_equal(bm.rho.cpu().numpy(), g["rho0"])
    assert bm.theta.shape == (2 * bm.p,)
    out = bm(g["x"], sample=True)                     # test_vi.py:22-46 forward shapes
    assert out.shape == (g["x"].shape[0], 1)
    bm.eval()
    a, b = bm(g["x"]), bm(g["x"])                     # eval mode -> variational mean, deterministic
    assert torch.equal(a, b)


def test_g6_ensemble_trajectories():
    g = load_golden("g6_ens.npz")
    net = _net(g)
    load_flat_into(net, g["w0"])
    ens = NN_Ens(net, nens=int(g["nens"]), dfrac=float(g["dfrac"]), verbose=False)
    np.random.seed(int(g["np_seed"]))
    torch.manual_seed(int(g["torch_seed"]))
    ens.fit(g["x"], g["y"], val=[g["xval"], g["yval"]], lrate=float(g["lrate"]), batch_size=int(g["batch_size"]),
            nepochs=int(g["nepochs"]), freq_out=1000)
    hist = np.array([l.history for l in ens.learners])
    np.testing.assert_allclose(hist, g["history"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(ens.fit_results["best_w"], g["best"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(ens.fit_results["final_w"], g["final"], rtol=1e-9, atol=1e-11)
    np.random.seed(int(g["predict_seed"]))
    yens = ens.predict_ens(g["xg"])
    assert yens.shape == g["yens"].shape                                   # (nens, N, 1), test_ensemble.py
    np.testing.assert_allclose(yens, g["yens"], rtol=1e-9, atol=1e-11)
    ym, yv, _ = ens.predict_mom_sample(g["xg"], msc=1, nsam=3)
    assert ym.shape == (len(g["xg"]), 1) and (yv >= 0).all()
    assert np.allclose(ens.learners[0].predict(g["xg"]), g["yens"][list(np.random.RandomState(int(g["predict_seed"])).permutation(3)).index(0)], rtol=1e-9)


@pytest.mark.parametrize("bs,opt", [(7, "adam"), (None, "adam"), (5, "sgd")])
def test_nnfit_single_member_vs_live_oracle(bs, opt):
    """Ragged last minibatch (24 rows, batches of 7 -> 7,7,7,3), full batch, and SGD."""
    g = load_golden("g6_ens.npz")
    spec = spec_of(g)
    net = _net(g)
    load_flat_into(net, g["w0"])
    x, y, xv, yv = g["x"][:24], g["y"][:24], g["xval"], g["yval"]
    gen = torch.Generator(); gen.manual_seed(5)
    ref = fit_ref.fit_member_mse(spec, g["w0"], x, y, xv, yv, 12, bs, 0.02, gen, wd=1e-3, optimizer=opt)
    torch.manual_seed(5)
    info = nnfit(net, x, y, val=[xv, yv], lrate=0.02, batch_size=bs, nepochs=12, wd=1e-3, optimizer=opt, freq_out=1000)
    np.testing.assert_allclose(np.array(info["history"]), ref["history"], rtol=1e-9, atol=1e-12)
    assert info["best_epoch"] == ref["best_epoch"]
    best = np.concatenate([q.detach().flatten().numpy() for q in info["best_nnmodel"].parameters()])
    np.testing.assert_allclose(best, ref["best"], rtol=1e-9, atol=1e-11)
    fin = np.concatenate([q.detach().flatten().numpy() for q in net.parameters()])
    np.testing.assert_allclose(fin, ref["final"], rtol=1e-9, atol=1e-11)
    assert info["best_loss"] < info["history"][0][3] + 1e-12           # test_nnfit.py: loss decreases


def test_ensemble_shapes_two_outputs_device_rng():
    """test_ensemble.py: 2-output net, dfrac=0.8, predict shapes; device-side permutations."""
    rs = np.random.RandomState(0)
    x = rs.rand(40, 2); y = np.stack([np.sin(x[:, 0]), np.cos(x[:, 1])], axis=1)
    ens = NN_Ens(MLP(2, 2, (8, 8), activ='tanh'), nens=4, dfrac=0.8, verbose=False)
    ens.fit(x, y, val=[x[:10], y[:10]], lrate=0.01, batch_size=16, nepochs=30, perm_mode='device', freq_out=1000)
    assert ens.predict_ens(x).shape == (4, 40, 2)
    assert ens.predict_sample(x).shape == (40, 2)
    h = np.array([l.history for l in ens.learners])
    assert h.shape == (4, 30 * 2, 4) and (h[:, -1, 3] < h[:, 0, 3]).all()


def test_g9_rms_anchored_ensemble_trajectories():
    from quinn_amd.solvers.nn_rms import NN_RMS
    g = load_golden("g9_rms.npz")
    net = _net(g)
    load_flat_into(net, g["w0"])
    rms = NN_RMS(net, nens=int(g["nens"]), dfrac=float(g["dfrac"]), verbose=False, datanoise=float(g["datanoise"]),
                 priorsigma=float(g["priorsigma"]))
    np.random.seed(int(g["np_seed"]))
    torch.manual_seed(int(g["torch_seed"]))
    rms.fit(g["x"], g["y"], val=[g["xval"], g["yval"]], lrate=float(g["lrate"]), batch_size=int(g["batch_size"]),
            nepochs=int(g["nepochs"]), freq_out=1000)
    hist = np.array([l.history for l in rms.learners])
    np.testing.assert_allclose(hist, g["history"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(rms.fit_results["best_w"], g["best"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(rms.fit_results["final_w"], g["final"], rtol=1e-9, atol=1e-11)
    assert rms.predict_ens(g["x"]).shape == (3, g["x"].shape[0], 1)


def test_reduce_lr_on_plateau_matches_torch_scheduler():
    g = load_golden("g6_ens.npz")
    spec = spec_of(g)
    net = _net(g)
    load_flat_into(net, g["w0"])
    x, y, xv, yv = g["x"][:24], g["y"][:24], g["xval"], g["yval"]
    gen = torch.Generator(); gen.manual_seed(9)
    ref = fit_ref.fit_member_plateau(spec, g["w0"], x, y, xv, yv, 60, 12, 0.05, gen, cooldown=3, factor=0.5)
    assert ref["lrs"][-1] < 0.05                                        # the schedule actually fired
    torch.manual_seed(9)
    info = nnfit(net, x, y, val=[xv, yv], lrate=0.05, batch_size=12, nepochs=60, scheduler_lr="ReduceLROnPlateau",
                 cooldown=3, factor=0.5, freq_out=1000)
    h = np.array(info["history"])
    np.testing.assert_allclose(h[:, 1], ref["history"][:, 0], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(h[:, 3], ref["history"][:, 1], rtol=1e-8, atol=1e-11)
    fin = np.concatenate([q.detach().flatten().numpy() for q in net.parameters()])
    np.testing.assert_allclose(fin, ref["final"], rtol=1e-8, atol=1e-10)


def test_g11_ensemble_without_validation_set():
    """dfrac < 1 and no val=: members validate on their own subsets (tests/test_ensemble.py:96-111 of the reference)."""
    g = load_golden("g11_ens_noval.npz")
    net = _net(g)
    load_flat_into(net, g["w0"])
    ens = NN_Ens(net, nens=int(g["nens"]), dfrac=float(g["dfrac"]), verbose=False)
    np.random.seed(int(g["np_seed"]))
    torch.manual_seed(int(g["torch_seed"]))
    ens.fit(g["x"], g["y"], lrate=float(g["lrate"]), batch_size=int(g["batch_size"]), nepochs=int(g["nepochs"]),
            freq_out=1000)
    hist = np.array([l.history for l in ens.learners])
    np.testing.assert_allclose(hist, g["history"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(ens.fit_results["best_w"], g["best"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(ens.fit_results["final_w"], g["final"], rtol=1e-9, atol=1e-11)


def test_nnfit_logpost_with_prior_vs_live_oracle():
    """nnfit(loss_fn='logpost', priorparams=...): the single-module form of the anchored trainer (nnfit.py:64-66)."""
    g = load_golden("g9_rms.npz")
    spec = spec_of(g)
    net = _net(g)
    load_flat_into(net, g["w0"])
    anchor = np.random.RandomState(5).randn(spec.nparams) * 0.5
    x, y = g["x"][:20], g["y"][:20]
    gen = torch.Generator(); gen.manual_seed(9)
    ref = fit_ref.fit_member_logpost(spec, g["w0"], x, y, g["xval"], g["yval"], 10, 5, 0.02, gen, 0.1,
                                     anchor=anchor, prior_sigma=0.5)
    torch.manual_seed(9)
    res = nnfit(net, x, y, val=[g["xval"], g["yval"]], loss_fn='logpost', datanoise=0.1,
                priorparams={'sigma': 0.5, 'anchor': torch.as_tensor(anchor)}, lrate=0.02, batch_size=5, nepochs=10,
                freq_out=1000)
    np.testing.assert_allclose(np.array(res['history']), ref["history"], rtol=1e-9, atol=1e-10)
    from quinn_amd.ops import flatten_module
    np.testing.assert_allclose(flatten_module(res['best_nnmodel']), ref["best"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(flatten_module(net), ref["final"], rtol=1e-9, atol=1e-11)
