#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference; the GPU box has neither
the reference nor a need for this script -- it reads the committed .npz files):

    cd /tmp && PYTHONPATH=/root/reference MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 \
        python3 /root/repo/tests/golden/gen_golden.py

The fixtures are DATA: inputs (architecture, weights, x, y, sigma, seeds, RNG
draws) and the outputs the reference produced for them.  Groups follow SURVEY.md
section 8c (G1..G8).  Versions at generation time are stored in each file.
"""
import os
import sys
import tempfile

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("QUINN_REFERENCE", "/root/reference")
if REF not in sys.path:
    sys.path.insert(0, REF)
os.environ.setdefault("MPLBACKEND", "Agg")
os.chdir(tempfile.mkdtemp(prefix="quinn_golden_"))      # nnfit drops PNGs into the CWD

from quinn.nns.mlp import MLP                            # noqa: E402
from quinn.nns.nnwrap import NNWrap, nn_p                # noqa: E402
from quinn.solvers.nn_mcmc import NN_MCMC                # noqa: E402
from quinn.solvers.nn_vi import NN_VI                    # noqa: E402
from quinn.solvers.nn_ens import NN_Ens                  # noqa: E402
from quinn.mcmc.admcmc import AMCMC                      # noqa: E402
from quinn.mcmc.hmc import HMC                           # noqa: E402
from quinn.mcmc.mala import MALA                         # noqa: E402
from quinn.vi.bnet import BNet                           # noqa: E402

VERS = np.array([torch.__version__, np.__version__])


def data(N, d, o, noise, seed):
    rs = np.random.RandomState(seed)
    x = (rs.rand(N, d) * 2 - 1) * np.pi
    y = np.stack([np.sum(np.sin((k + 1) * x), axis=1) for k in range(o)], axis=1) + noise * rs.randn(N, o)
    return x, y


def save(name, **kw):
    np.savez_compressed(os.path.join(OUT, name), versions=VERS, **kw)
    print("wrote", name, {k: np.asarray(v).shape for k, v in kw.items()})


# ---------------------------------------------------------------- G1: logpost / grad
def g1():
    cases = [(1, 1, (16, 16), "tanh", 64), (1, 1, (64, 64, 64), "tanh", 48),
             (2, 1, (8, 8), "relu", 40), (2, 2, (8,), "tanh", 33), (3, 2, (5, 7), "identity", 20)]
    for ci, (d, o, hls, act, N) in enumerate(cases):
        net = MLP(d, o, hls, activ=act)
        solver = NN_MCMC(net, verbose=False)
        p = solver.pdim
        x, y = data(N, d, o, 0.05, 10 + ci)
        sigma = 0.3
        lpinfo = {"model": nn_p, "xd": x, "yd": [yy for yy in y], "ltype": "classical",
                  "lparams": {"sigma": sigma}}
        rs = np.random.RandomState(100 + ci)
        W = 0.4 * rs.randn(8, p)
        lp = np.array([solver.logpost(w, lpinfo) for w in W])
        gr = np.array([solver.logpostgrad(w, lpinfo) for w in W])
        pred = np.array([nn_p(w, x, solver.nnmodel) for w in W])
        save(f"g1_logpost_{ci}.npz", dims=np.array((d,) + tuple(hls) + (o,)), activ=np.array(act),
             x=x, y=y, sigma=sigma, W=W, logpost=lp, grad=gr, pred=pred)


# ---------------------------------------------------------------- G2/G3/G8: chains
def _solver(d, o, hls, act, N, seed, sigma):
    net = MLP(d, o, hls, activ=act)
    solver = NN_MCMC(net, verbose=False)
    x, y = data(N, d, o, 0.05, seed)
    return solver, x, y


def _record_uniforms():
    """Wrap np.random.random_sample to capture the accept-test uniforms."""
    drawn = []
    orig = np.random.random_sample

    def wrapped(*a, **k):
        u = orig(*a, **k)
        drawn.append(u)
        return u
    np.random.random_sample = wrapped
    return drawn, orig


def run_fit(solver, x, y, sigma, nmcmc, sampler, sp, seed):
    np.random.seed(seed)
    drawn, orig = _record_uniforms()
    try:
        solver.fit(x, y, zflag=False, datanoise=sigma, nmcmc=nmcmc, sampler=sampler, sampler_params=dict(sp))
    finally:
        np.random.random_sample = orig
    return np.array(drawn)


def g2_g3_g8():
    d, o, hls, act, N, sigma = 1, 1, (8, 8), "tanh", 32, 0.2
    # G2 adaptive Metropolis, adaptation fires (t0 / tadapt small)
    for gi, (gamma, nmcmc, seed) in enumerate([(0.1, 400, 7), (0.01, 400, 8)]):
        solver, x, y = _solver(d, o, hls, act, N, 20, sigma)
        sp = {"gamma": gamma, "t0": 20, "tadapt": 50}
        u = run_fit(solver, x, y, sigma, nmcmc, "amcmc", sp, seed)
        # re-run through the sampler class directly to get the full result dict
        np.random.seed(seed)
        ini = np.random.rand(solver.pdim)
        mc = AMCMC(**sp)
        mc.setLogPost(solver.logpost, None, lpinfo=solver.lpinfo)
        res = mc.run(param_ini=ini, nmcmc=nmcmc)
        assert np.array_equal(res["chain"], solver.samples)
        save(f"g2_amcmc_{gi}.npz", dims=np.array((d,) + hls + (o,)), activ=np.array(act), x=x, y=y,
             sigma=sigma, seed=seed, nmcmc=nmcmc, gamma=gamma, t0=20, tadapt=50,
             chain=res["chain"], logpost=res["logpost"], alphas=res["alphas"], accrate=res["accrate"],
             mapparams=res["mapparams"], maxpost=res["maxpost"], uniforms=u)
    # cfg1-shaped short chain (2x16, N=256), default t0/tadapt -> no adaptation inside 60 steps
    solver, x, y = _solver(1, 1, (16, 16), "tanh", 256, 21, 0.02)
    np.random.seed(3)
    ini = np.random.rand(solver.pdim)
    mc = AMCMC(gamma=0.01)
    solver.lpinfo = {"model": nn_p, "xd": x, "yd": [yy for yy in y], "ltype": "classical",
                     "lparams": {"sigma": 0.1}}
    mc.setLogPost(solver.logpost, None, lpinfo=solver.lpinfo)
    res = mc.run(param_ini=ini, nmcmc=60)
    save("g2_amcmc_cfg1.npz", dims=np.array((1, 16, 16, 1)), activ=np.array("tanh"), x=x, y=y, sigma=0.1,
         seed=3, nmcmc=60, gamma=0.01, t0=100, tadapt=1000, chain=res["chain"], logpost=res["logpost"],
         alphas=res["alphas"], accrate=res["accrate"], mapparams=res["mapparams"], maxpost=res["maxpost"])
    # G3 HMC (through NN_MCMC.fit) and MALA (standalone: unreachable via fit in the reference)
    for gi, (L, eps, nmcmc, seed) in enumerate([(3, 0.002, 120, 11), (10, 0.001, 60, 12)]):
        solver, x, y = _solver(d, o, hls, act, N, 22, sigma)
        np.random.seed(seed)
        ini = np.random.rand(solver.pdim)
        solver.lpinfo = {"model": nn_p, "xd": x, "yd": [yy for yy in y], "ltype": "classical",
                         "lparams": {"sigma": sigma}}
        mc = HMC(epsilon=eps, L=L)
        mc.setLogPost(solver.logpost, solver.logpostgrad, lpinfo=solver.lpinfo)
        res = mc.run(param_ini=ini, nmcmc=nmcmc)
        save(f"g3_hmc_{gi}.npz", dims=np.array((d,) + hls + (o,)), activ=np.array(act), x=x, y=y,
             sigma=sigma, seed=seed, nmcmc=nmcmc, L=L, epsilon=eps, chain=res["chain"],
             logpost=res["logpost"], alphas=res["alphas"], accrate=res["accrate"],
             mapparams=res["mapparams"], maxpost=res["maxpost"])
    solver, x, y = _solver(d, o, hls, act, N, 23, sigma)
    np.random.seed(13)
    ini = np.random.rand(solver.pdim)
    solver.lpinfo = {"model": nn_p, "xd": x, "yd": [yy for yy in y], "ltype": "classical",
                     "lparams": {"sigma": sigma}}
    mc = MALA(epsilon=0.002)
    mc.setLogPost(solver.logpost, solver.logpostgrad, lpinfo=solver.lpinfo)
    res = mc.run(param_ini=ini, nmcmc=100)
    save("g3_mala.npz", dims=np.array((d,) + hls + (o,)), activ=np.array(act), x=x, y=y, sigma=sigma,
         seed=13, nmcmc=100, epsilon=0.002, chain=res["chain"], logpost=res["logpost"],
         alphas=res["alphas"], accrate=res["accrate"], mapparams=res["mapparams"], maxpost=res["maxpost"])
    # G8 multi-chain definition: C sequential fits, chain c preceded by np.random.seed(seed0 + c)
    C, seed0, nmcmc = 4, 40, 150
    chains, lps, als, accs, maps = [], [], [], [], []
    solver, x, y = _solver(d, o, hls, act, N, 24, sigma)
    for c in range(C):
        np.random.seed(seed0 + c)
        ini = np.random.rand(solver.pdim)
        mc = AMCMC(gamma=0.05, t0=30, tadapt=40)
        solver.lpinfo = {"model": nn_p, "xd": x, "yd": [yy for yy in y], "ltype": "classical",
                         "lparams": {"sigma": sigma}}
        mc.setLogPost(solver.logpost, None, lpinfo=solver.lpinfo)
        res = mc.run(param_ini=ini, nmcmc=nmcmc)
        chains.append(res["chain"]); lps.append(res["logpost"]); als.append(res["alphas"])
        accs.append(res["accrate"]); maps.append(res["mapparams"])
    save("g8_multichain.npz", dims=np.array((d,) + hls + (o,)), activ=np.array(act), x=x, y=y, sigma=sigma,
         seed0=seed0, nchains=C, nmcmc=nmcmc, gamma=0.05, t0=30, tadapt=40, chain=np.array(chains),
         logpost=np.array(lps), alphas=np.array(als), accrate=np.array(accs), mapparams=np.array(maps))
    # G7 prediction from a chain: predict_ens thinning + predict_mom_sample(msc=2)
    solver.samples = chains[0]
    solver.cmode = maps[0]
    xg = np.linspace(-3, 3, 17)[:, None]
    yens = solver.predict_ens(xg, nens=10, nburn=50)
    solver.nens = 10
    import functools
    solver.predict_ens = functools.partial(solver.predict_ens, nburn=50)
    ymean, yvar, ycov = solver.predict_mom_sample(xg, msc=2, nsam=10)
    save("g7_predict.npz", dims=np.array((d,) + hls + (o,)), activ=np.array(act), chain=chains[0], xg=xg,
         nens=10, nburn=50, yens=yens, ymean=ymean, yvar=yvar, ycov=ycov, ymap=solver.predict_MAP(xg))


# ---------------------------------------------------------------- G4/G5: VI
def _bnet_flat(bm):
    mus = [p.detach().flatten().numpy() for n, p in bm.named_parameters() if n.endswith("_mu")]
    rhos = [p.detach().flatten().numpy() for n, p in bm.named_parameters() if n.endswith("_rho")]
    return np.concatenate(mus), np.concatenate(rhos)


def _bnet_grads(bm):
    gm = [p.grad.detach().flatten().numpy() for n, p in bm.named_parameters() if n.endswith("_mu")]
    gr = [p.grad.detach().flatten().numpy() for n, p in bm.named_parameters() if n.endswith("_rho")]
    return np.concatenate(gm), np.concatenate(gr)


def _record_normals():
    drawn = []
    orig = torch.distributions.Normal.sample

    def wrapped(self, sample_shape=torch.Size()):
        z = orig(self, sample_shape)
        drawn.append(z.detach().flatten().numpy().copy())
        return z
    torch.distributions.Normal.sample = wrapped
    return drawn, orig


def g4_g5():
    for ci, (d, o, hls, act, N, S, prior) in enumerate([
            (1, 1, (8, 8), "tanh", 24, 1, (0.5, 1.0, 1.0)),
            (2, 1, (16, 16, 16), "tanh", 40, 3, (0.5, 1.0, 1.0)),
            (2, 2, (8,), "relu", 30, 8, (0.3, 0.5, 2.0))]):
        torch.manual_seed(50 + ci)
        net = MLP(d, o, hls, activ=act)
        x, y = data(N, d, o, 0.05, 30 + ci)
        bm = BNet(net, pi=prior[0], sigma1=prior[1], sigma2=prior[2])
        mu, rho = _bnet_flat(bm)
        datanoise, nb = 0.1, 3
        bm.loss_params = [datanoise, S, nb]
        drawn, orig = _record_normals()
        try:
            xt, yt = torch.tensor(x), torch.tensor(y)
            lp, lq, nll = bm.sample_elbo(xt, yt, S, likparams=[datanoise])
            n_first = len(drawn)
            loss = bm.viloss(xt, yt)
        finally:
            torch.distributions.Normal.sample = orig
        loss.backward()
        dmu, drho = _bnet_grads(bm)
        eps_elbo = np.concatenate(drawn[:n_first]).reshape(S, -1)
        eps_loss = np.concatenate(drawn[n_first:]).reshape(S, -1)
        save(f"g4_viloss_{ci}.npz", dims=np.array((d,) + hls + (o,)), activ=np.array(act), x=x, y=y,
             mu=mu, rho=rho, nsam=S, datanoise=datanoise, num_batches=nb, prior=np.array(prior),
             eps_elbo=eps_elbo, elbo_log_prior=lp.item(), elbo_log_q=lq.item(), elbo_nll=nll.item(),
             eps_loss=eps_loss, loss=loss.item(), dmu=dmu, drho=drho, torch_seed=50 + ci)
    # G5 NN_VI.fit, 20 epochs, minibatches
    d, o, hls, act, N = 1, 1, (8, 8), "tanh", 24
    torch.manual_seed(60)
    net = MLP(d, o, hls, activ=act)
    w_net = NNWrap(net).p_flatten().detach().numpy().flatten()
    x, y = data(N, d, o, 0.05, 33)
    xv, yv = data(9, d, o, 0.05, 34)
    vi = NN_VI(net, verbose=False)
    mu0, rho0 = _bnet_flat(vi.bmodel)
    gen_state = torch.get_rng_state().numpy().copy()
    vi.fit(x, y, val=[xv, yv], datanoise=0.1, lrate=0.01, batch_size=10, nsam=2, nepochs=20, freq_out=1000)
    mu1, rho1 = _bnet_flat(vi.bmodel)
    mub, rhob = _bnet_flat(vi.best_model)
    # history lives only inside nnfit's return; rerun identically through nnfit to capture it
    from quinn.nns.nnfit import nnfit
    torch.manual_seed(60)
    net2 = MLP(d, o, hls, activ=act)
    vi2 = NN_VI(net2, verbose=False)
    vi2.bmodel.loss_params = [0.1, 2, (N + 1) // 10]
    info = nnfit(vi2.bmodel, x, y, val=[xv, yv], loss_xy=vi2.bmodel.viloss, lrate=0.01, batch_size=10,
                 nepochs=20, freq_out=1000)
    mu2, rho2 = _bnet_flat(vi2.bmodel)
    assert np.array_equal(mu1, mu2) and np.array_equal(rho1, rho2)
    save("g5_vifit.npz", dims=np.array((d,) + hls + (o,)), activ=np.array(act), x=x, y=y, xval=xv, yval=yv,
         mu0=mu0, rho0=rho0, gen_state=gen_state, datanoise=0.1, lrate=0.01, batch_size=10, nsam=2,
         nepochs=20, mu_final=mu1, rho_final=rho1, mu_best=mub, rho_best=rhob,
         history=np.array(info["history"]), best_loss=info["best_loss"], best_epoch=info["best_epoch"],
         torch_seed=60, w_net=w_net)


# ---------------------------------------------------------------- G6: deep ensemble
def g6():
    d, o, hls, act, N = 1, 1, (8, 8), "tanh", 30
    torch.manual_seed(70)
    net = MLP(d, o, hls, activ=act)
    w0 = NNWrap(net).p_flatten().detach().numpy().flatten()
    x, y = data(N, d, o, 0.05, 35)
    xv, yv = data(8, d, o, 0.05, 36)
    M = 3
    ens = NN_Ens(net, nens=M, dfrac=0.8, verbose=False)
    np.random.seed(71)
    torch.manual_seed(72)
    ens.fit(x, y, val=[xv, yv], lrate=0.01, batch_size=8, nepochs=20, freq_out=1000)
    hist = np.array([np.array(l.nnmodel.history) for l in ens.learners])
    best = np.array([NNWrap(l.best_model).p_flatten().detach().numpy().flatten() for l in ens.learners])
    final = np.array([np.concatenate([p.detach().flatten().numpy() for p in l.nnmodel.nnmodel.parameters()])
                      for l in ens.learners])
    xg = np.linspace(-3, 3, 11)[:, None]
    np.random.seed(73)
    yens = ens.predict_ens(xg)
    save("g6_ens.npz", dims=np.array((d,) + hls + (o,)), activ=np.array(act), x=x, y=y, xval=xv, yval=yv,
         w0=w0, nens=M, dfrac=0.8, lrate=0.01, batch_size=8, nepochs=20, np_seed=71, torch_seed=72,
         history=hist, best=best, final=final, xg=xg, predict_seed=73, yens=yens)
    # full-batch, dfrac=1: all members identical (shared deepcopy) -- a property the build must keep
    ens2 = NN_Ens(net, nens=2, dfrac=1.0, verbose=False)
    np.random.seed(74); torch.manual_seed(75)
    ens2.fit(x, y, lrate=0.05, nepochs=15, freq_out=1000)
    best2 = np.array([NNWrap(l.best_model).p_flatten().detach().numpy().flatten() for l in ens2.learners])
    hist2 = np.array([np.array(l.nnmodel.history) for l in ens2.learners])
    save("g6_ens_fullbatch.npz", dims=np.array((d,) + hls + (o,)), activ=np.array(act), x=x, y=y, w0=w0,
         nens=2, lrate=0.05, nepochs=15, np_seed=74, torch_seed=75, best=best2, history=hist2)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "chains", "vi", "ens"]
    if "g1" in which:
        g1()
    if "chains" in which:
        g2_g3_g8()
    if "vi" in which:
        g4_g5()
    if "ens" in which:
        g6()


# ---------------------------------------------------------------- timing equivalence
def timing_ratio():
    """The bench's CPU baseline times the oracle's sequential log-posterior, because the reference
    cannot travel to the GPU box.  Record here, where both run, that the two cost the same:
    median time per evaluation of NN_MCMC.logpost (reference) and oracle.mlp_ref.logpost on the
    cfg2 workload (3x64 tanh MLP, N=4096), same thread count."""
    import json
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import mlp_ref
    N, hls = 4096, (64, 64, 64)
    x, y = mlp_ref.synthetic_data(N, 1, 0.02, seed=0)
    net = MLP(1, 1, hls, activ='tanh')
    solver = NN_MCMC(net, verbose=False)
    lpinfo = {"model": nn_p, "xd": x, "yd": [yy for yy in y], "ltype": "classical", "lparams": {"sigma": 0.02}}
    mod = mlp_ref.build_module(mlp_ref.MLPSpec((1,) + hls + (1,), "tanh"))
    ws = [0.1 * np.random.RandomState(1000 + c).randn(solver.pdim) for c in range(8)]
    out = {}
    for nt in (1, 8):
        torch.set_num_threads(nt)
        def med(fn):
            for i in range(10):
                fn(ws[i % 8])
            ts = []
            for i in range(120):
                t0 = time.perf_counter(); fn(ws[i % 8]); ts.append(time.perf_counter() - t0)
            return float(np.median(ts))
        tr = med(lambda w: solver.logpost(w, lpinfo))
        to = med(lambda w: mlp_ref.logpost(mod, w, x, lpinfo["yd"], 0.02))
        assert solver.logpost(ws[0], lpinfo) == mlp_ref.logpost(mod, ws[0], x, lpinfo["yd"], 0.02)
        out[f"threads_{nt}"] = {"reference_ms": 1e3 * tr, "oracle_ms": 1e3 * to, "ratio_oracle_over_reference": to / tr}
    json.dump(out, open(os.path.join(OUT, "timing_ratio.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__" and "timing" in sys.argv[1:]:
    timing_ratio()


# ---------------------------------------------------------------- G9: NN_RMS (anchored ensemble)
def g9():
    from quinn.solvers.nn_rms import NN_RMS
    d, o, hls, act, N = 1, 1, (8, 8), "tanh", 30
    torch.manual_seed(80)
    net = MLP(d, o, hls, activ=act)
    w0 = NNWrap(net).p_flatten().detach().numpy().flatten()
    x, y = data(N, d, o, 0.05, 37)
    xv, yv = data(8, d, o, 0.05, 38)
    rms = NN_RMS(net, nens=3, dfrac=0.8, verbose=False, datanoise=0.1, priorsigma=0.5)
    np.random.seed(81)
    torch.manual_seed(82)
    rms.fit(x, y, val=[xv, yv], lrate=0.01, batch_size=8, nepochs=15, freq_out=1000)
    hist = np.array([np.array(l.nnmodel.history) for l in rms.learners])
    best = np.array([NNWrap(l.best_model).p_flatten().detach().numpy().flatten() for l in rms.learners])
    final = np.array([np.concatenate([p.detach().flatten().numpy() for p in l.nnmodel.nnmodel.parameters()])
                      for l in rms.learners])
    save("g9_rms.npz", dims=np.array((d,) + hls + (o,)), activ=np.array(act), x=x, y=y, xval=xv, yval=yv, w0=w0,
         nens=3, dfrac=0.8, lrate=0.01, batch_size=8, nepochs=15, np_seed=81, torch_seed=82, datanoise=0.1,
         priorsigma=0.5, history=hist, best=best, final=final)


if __name__ == "__main__" and "rms" in sys.argv[1:]:
    g9()


# ---------------------------------------------------------------- G10: residual network (RNet)
def _rnet_fields(kw, wp_kind, wp_arg):
    return dict(rdim=kw["rdim"], nlayers=kw["nlayers"], wp_kind=np.array(wp_kind), wp_arg=wp_arg,
                indim=kw.get("indim") or 0, outdim=kw.get("outdim") or 0, biasorno=kw.get("biasorno", True),
                nonlin=kw.get("nonlin", True), mlp=kw.get("mlp", False), layer_pre=kw.get("layer_pre", False),
                layer_post=kw.get("layer_post", False))


def _rnet(kw, wp_kind, wp_arg):
    from quinn.nns import rnet as R
    wp = {"const": lambda: R.Const(), "lin": lambda: R.Lin(), "quad": lambda: R.Quad(), "cubic": lambda: R.Cubic(),
          "poly": lambda: R.Poly(wp_arg), "nonpar": lambda: (R.NonPar(wp_arg) if wp_arg else None)}[wp_kind]()
    return R.RNet(kw["rdim"], kw["nlayers"], wp_function=wp, **{k: v for k, v in kw.items()
                                                                if k not in ("rdim", "nlayers")})


RNET_CASES = [
    # the network of examples/ex_ufit.py:72-77
    (dict(rdim=3, nlayers=3, indim=1, outdim=1, layer_pre=True, layer_post=True), "poly", 0),
    # tests/test_mlp.py:135-145 (default NonPar(nlayers+1))
    (dict(rdim=5, nlayers=3, indim=2, outdim=1, layer_pre=True, layer_post=True), "nonpar", 0),
    # same width everywhere, no pre / post layer (tests/test_mlp.py:147-156)
    (dict(rdim=3, nlayers=4), "lin", 0),
    (dict(rdim=4, nlayers=3, indim=2, outdim=1, mlp=True, layer_pre=True, layer_post=True), "quad", 0),
    (dict(rdim=6, nlayers=2, indim=2, outdim=2, layer_pre=True, layer_post=True, biasorno=False, nonlin=False),
     "cubic", 0),
    (dict(rdim=12, nlayers=5, indim=3, outdim=2, layer_pre=True, layer_post=True), "poly", 2),
]


def g10():
    for ci, (kw, wk, wa) in enumerate(RNET_CASES):
        torch.manual_seed(90 + ci)
        net = _rnet(kw, wk, wa)
        w_init = np.concatenate([p.detach().flatten().numpy() for p in net.parameters()])
        solver = NN_MCMC(net, verbose=False)
        p = solver.pdim
        d, o = net.indim, net.outdim
        N = 37 + 5 * ci
        x, y = data(N, d, o, 0.05, 110 + ci)
        sigma = 0.3
        lpinfo = {"model": nn_p, "xd": x, "yd": [yy for yy in y], "ltype": "classical",
                  "lparams": {"sigma": sigma}}
        W = 0.5 * np.random.RandomState(120 + ci).randn(8, p)
        lp = np.array([solver.logpost(w, lpinfo) for w in W])
        gr = np.array([solver.logpostgrad(w, lpinfo) for w in W])
        pred = np.array([nn_p(w, x, solver.nnmodel) for w in W])
        save(f"g10_rnet_logpost_{ci}.npz", x=x, y=y, sigma=sigma, W=W, logpost=lp, grad=gr, pred=pred,
             torch_seed=90 + ci, w_init=w_init, **_rnet_fields(kw, wk, wa))
    # chains on the ex_ufit network: adaptive Metropolis through NN_MCMC.fit, HMC through the sampler class
    kw, wk, wa = RNET_CASES[0]
    x, y = data(14, 1, 1, 0.02, 130)
    torch.manual_seed(96)
    solver = NN_MCMC(_rnet(kw, wk, wa), verbose=False)
    sp = {"gamma": 0.05, "t0": 20, "tadapt": 50}
    u = run_fit(solver, x, y, 0.2, 300, "amcmc", sp, 17)
    np.random.seed(17)
    ini = np.random.rand(solver.pdim)
    mc = AMCMC(**sp)
    mc.setLogPost(solver.logpost, None, lpinfo=solver.lpinfo)
    res = mc.run(param_ini=ini, nmcmc=300)
    assert np.array_equal(res["chain"], solver.samples)
    save("g10_rnet_amcmc.npz", x=x, y=y, sigma=0.2, seed=17, nmcmc=300, gamma=0.05, t0=20, tadapt=50,
         chain=res["chain"], logpost=res["logpost"], alphas=res["alphas"], accrate=res["accrate"],
         mapparams=res["mapparams"], maxpost=res["maxpost"], uniforms=u, **_rnet_fields(kw, wk, wa))
    kw, wk, wa = RNET_CASES[1]
    x, y = data(25, 2, 1, 0.05, 131)
    solver = NN_MCMC(_rnet(kw, wk, wa), verbose=False)
    np.random.seed(18)
    ini = np.random.rand(solver.pdim)
    solver.lpinfo = {"model": nn_p, "xd": x, "yd": [yy for yy in y], "ltype": "classical",
                     "lparams": {"sigma": 0.2}}
    mc = HMC(epsilon=0.002, L=3)
    mc.setLogPost(solver.logpost, solver.logpostgrad, lpinfo=solver.lpinfo)
    res = mc.run(param_ini=ini, nmcmc=80)
    save("g10_rnet_hmc.npz", x=x, y=y, sigma=0.2, seed=18, nmcmc=80, L=3, epsilon=0.002, chain=res["chain"],
         logpost=res["logpost"], alphas=res["alphas"], accrate=res["accrate"], mapparams=res["mapparams"],
         maxpost=res["maxpost"], **_rnet_fields(kw, wk, wa))
    # deep ensemble and VI fits on the ex_ufit network (call pattern of examples/ex_ufit.py:107-113)
    kw, wk, wa = RNET_CASES[0]
    torch.manual_seed(97)
    net = _rnet(kw, wk, wa)
    w0 = np.concatenate([p.detach().flatten().numpy() for p in net.parameters()])
    x, y = data(30, 1, 1, 0.05, 132)
    xv, yv = data(8, 1, 1, 0.05, 133)
    ens = NN_Ens(net, nens=3, dfrac=0.8, verbose=False)
    np.random.seed(98)
    torch.manual_seed(99)
    ens.fit(x, y, val=[xv, yv], lrate=0.01, batch_size=8, nepochs=20, freq_out=1000)
    hist = np.array([np.array(l.nnmodel.history) for l in ens.learners])
    best = np.array([NNWrap(l.best_model).p_flatten().detach().numpy().flatten() for l in ens.learners])
    final = np.array([np.concatenate([p.detach().flatten().numpy() for p in l.nnmodel.parameters()])
                      for l in ens.learners])
    xg = np.linspace(-3, 3, 11)[:, None]
    np.random.seed(100)
    yens = ens.predict_ens(xg)
    save("g10_rnet_ens.npz", x=x, y=y, xval=xv, yval=yv, w0=w0, nens=3, dfrac=0.8, lrate=0.01, batch_size=8,
         nepochs=20, np_seed=98, torch_seed=99, history=hist, best=best, final=final, xg=xg, predict_seed=100,
         yens=yens, **_rnet_fields(kw, wk, wa))
    torch.manual_seed(101)
    net = _rnet(kw, wk, wa)
    w_net = np.concatenate([p.detach().flatten().numpy() for p in net.parameters()])
    vi = NN_VI(net, verbose=False)
    mu0, rho0 = _bnet_flat(vi.bmodel)
    gen_state = torch.get_rng_state().numpy().copy()
    from quinn.nns.nnfit import nnfit
    vi.bmodel.loss_params = [0.1, 2, (30 + 1) // 10]
    info = nnfit(vi.bmodel, x, y, val=[xv, yv], loss_xy=vi.bmodel.viloss, lrate=0.01, batch_size=10,
                 nepochs=20, freq_out=1000)
    mu1, rho1 = _bnet_flat(vi.bmodel)
    mub, rhob = _bnet_flat(info["best_nnmodel"])
    save("g10_rnet_vifit.npz", x=x, y=y, xval=xv, yval=yv, mu0=mu0, rho0=rho0, gen_state=gen_state, datanoise=0.1,
         lrate=0.01, batch_size=10, nsam=2, nepochs=20, mu_final=mu1, rho_final=rho1, mu_best=mub, rho_best=rhob,
         history=np.array(info["history"]), best_loss=info["best_loss"], best_epoch=info["best_epoch"],
         torch_seed=101, w_net=w_net, **_rnet_fields(kw, wk, wa))


if __name__ == "__main__" and "rnet" in sys.argv[1:]:
    g10()


# ---------------------------------------------------------------- G11: ensemble without a validation set
def g11():
    """NN_Ens.fit with dfrac < 1 and val=None: every member validates on its own training subset
    (nnfit.py:106-109) -- the call pattern of tests/test_ensemble.py:96-111."""
    d, o, hls, act, N = 1, 1, (8,), "tanh", 40
    torch.manual_seed(110)
    net = MLP(d, o, hls, activ=act)
    w0 = NNWrap(net).p_flatten().detach().numpy().flatten()
    x, y = data(N, d, o, 0.05, 140)
    ens = NN_Ens(net, nens=2, dfrac=0.8, verbose=False)
    np.random.seed(111)
    torch.manual_seed(112)
    ens.fit(x, y, lrate=0.01, batch_size=10, nepochs=15, freq_out=1000)
    hist = np.array([np.array(l.nnmodel.history) for l in ens.learners])
    best = np.array([NNWrap(l.best_model).p_flatten().detach().numpy().flatten() for l in ens.learners])
    final = np.array([np.concatenate([p.detach().flatten().numpy() for p in l.nnmodel.nnmodel.parameters()])
                      for l in ens.learners])
    save("g11_ens_noval.npz", dims=np.array((d,) + hls + (o,)), activ=np.array(act), x=x, y=y, w0=w0, nens=2,
         dfrac=0.8, lrate=0.01, batch_size=10, nepochs=15, np_seed=111, torch_seed=112, history=hist, best=best,
         final=final)


if __name__ == "__main__" and "ens_noval" in sys.argv[1:]:
    g11()


# ---------------------------------------------------------------- G12: chains / ELBO on the shapes the DEFAULT kernels take
def g12():
    """Chains whose log-posterior / gradient evaluations are routed to the sliced int8-product kernels by the build
    (64-wide tanh networks: k_fused_fwd_i8 / k_fused_bwd_i8; a 40-wide network through its zero-padded 64-wide twin;
    a 128-wide network through k_i8_wide_*): HMC (L = 3, L = 10) and MALA on MLP(1,1,(64,64,64),'tanh'), adaptive
    Metropolis with the adaptation firing three times on MLP(1,1,(40,40),'tanh'), viloss + gradients at (2,128,128,1).
    Step sizes are chosen so that acceptances AND rejections occur (mh_prob on both sides of the uniform draw).
    The start point is data (0.1 * RandomState(s).rand(p)): the sampler classes are driven directly, as in G3."""
    import contextlib
    import io
    d, o, hls, act, N, sigma = 1, 1, (64, 64, 64), "tanh", 96, 0.2
    x, y = data(N, d, o, 0.05, 200)
    for name, kind, eps, L, nmcmc, seed in [("g12_hmc_0.npz", "hmc", 0.006, 3, 80, 11), ("g12_hmc_1.npz", "hmc", 0.004, 10, 60, 12),
                                            ("g12_mala.npz", "mala", 0.005, 1, 80, 13)]:
        solver = NN_MCMC(MLP(d, o, hls, activ=act), verbose=False)
        solver.lpinfo = {"model": nn_p, "xd": x, "yd": [yy for yy in y], "ltype": "classical", "lparams": {"sigma": sigma}}
        ini = 0.1 * np.random.RandomState(seed + 1000).rand(solver.pdim)
        np.random.seed(seed)
        drawn, orig = _record_uniforms()
        mc = HMC(epsilon=eps, L=L) if kind == "hmc" else MALA(epsilon=eps)
        mc.setLogPost(solver.logpost, solver.logpostgrad, lpinfo=solver.lpinfo)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                res = mc.run(param_ini=ini, nmcmc=nmcmc)
        finally:
            np.random.random_sample = orig
        acc = (res["chain"][1:] != res["chain"][:-1]).any(axis=1)
        assert 0.2 < acc.mean() < 0.95, acc.mean()
        # (p = 8513: the full chain would be 5 MB per fixture -- every step of 256 strided columns, plus three full states)
        cols = np.arange(0, solver.pdim, 34)[:256]
        save(name, dims=np.array((d,) + hls + (o,)), activ=np.array(act), x=x, y=y, sigma=sigma, seed=seed, nmcmc=nmcmc, L=L,
             epsilon=eps, param_ini=ini, accepted=acc, cols=cols, chain_cols=res["chain"][:, cols], chain_mid=res["chain"][nmcmc // 2],
             chain_final=res["chain"][-1], logpost=res["logpost"], alphas=res["alphas"], accrate=res["accrate"],
             mapparams=res["mapparams"], maxpost=res["maxpost"], uniforms=np.array(drawn))
    # adaptive Metropolis, p = 1761 (the reference refactorises the 1761 x 1761 proposal covariance on every draw: ~1.1 s per step)
    hls, N, seed, nmcmc = (40, 40), 80, 14, 70
    x, y = data(N, d, o, 0.05, 201)
    solver = NN_MCMC(MLP(d, o, hls, activ=act), verbose=False)
    solver.lpinfo = {"model": nn_p, "xd": x, "yd": [yy for yy in y], "ltype": "classical", "lparams": {"sigma": sigma}}
    ini = 0.1 * np.random.RandomState(seed + 1000).rand(solver.pdim)
    sp = {"gamma": 3.0, "t0": 10, "tadapt": 20}
    np.random.seed(seed)
    drawn, orig = _record_uniforms()
    mc = AMCMC(cov_ini=1e-5 * np.eye(solver.pdim), **sp)
    mc.setLogPost(solver.logpost, None, lpinfo=solver.lpinfo)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            res = mc.run(param_ini=ini, nmcmc=nmcmc)
    finally:
        np.random.random_sample = orig
    acc = (res["chain"][1:] != res["chain"][:-1]).any(axis=1)
    assert 0.2 < acc.mean() < 0.95 and 0 < acc[21:].mean() < 1, acc.mean()
    save("g12_amcmc.npz", dims=np.array((d,) + hls + (o,)), activ=np.array(act), x=x, y=y, sigma=sigma, seed=seed, nmcmc=nmcmc,
         cov_ini_diag=1e-5, param_ini=ini, chain=res["chain"], logpost=res["logpost"], alphas=res["alphas"],
         accrate=res["accrate"], mapparams=res["mapparams"], maxpost=res["maxpost"], uniforms=np.array(drawn), **sp)
    # viloss + gradients on a 128-wide network (the wide int8-slice kernels), S = 3
    _viloss_fixture("g12_viloss.npz", act, 120, 202)


def _viloss_fixture(name, act, torch_seed, data_seed):
    d, o, hls, N, S, prior = 2, 1, (128, 128), 200, 3, (0.5, 1.0, 1.0)
    torch.manual_seed(torch_seed)
    net = MLP(d, o, hls, activ=act)
    x, y = data(N, d, o, 0.05, data_seed)
    bm = BNet(net, pi=prior[0], sigma1=prior[1], sigma2=prior[2])
    mu, rho = _bnet_flat(bm)
    datanoise, nb = 0.1, 2
    bm.loss_params = [datanoise, S, nb]
    drawn, orig = _record_normals()
    try:
        xt, yt = torch.tensor(x), torch.tensor(y)
        lp, lq, nll = bm.sample_elbo(xt, yt, S, likparams=[datanoise])
        n_first = len(drawn)
        loss = bm.viloss(xt, yt)
    finally:
        torch.distributions.Normal.sample = orig
    loss.backward()
    dmu, drho = _bnet_grads(bm)
    save(name, dims=np.array((d,) + hls + (o,)), activ=np.array(act), x=x, y=y, mu=mu, rho=rho, nsam=S,
         datanoise=datanoise, num_batches=nb, prior=np.array(prior), eps_elbo=np.concatenate(drawn[:n_first]).reshape(S, -1),
         elbo_log_prior=lp.item(), elbo_log_q=lq.item(), elbo_nll=nll.item(),
         eps_loss=np.concatenate(drawn[n_first:]).reshape(S, -1), loss=loss.item(), dmu=dmu, drho=drho, torch_seed=torch_seed)


def g13():
    """The reference's DEFAULT activation (quinn/nns/mlp.py:23 `activ='relu'`) on the shapes the build's int8-slice kernels take
    since round 4 (per-row activation scales, k_fused_fwd_i8<.., relu> / k_fused_bwd_i8<.., relu>): HMC (L = 3), MALA and adaptive
    Metropolis before its first adaptation on MLP(1,1,(40,40),'relu') (p = 1761: the reference draws every proposal through an SVD
    of the p x p covariance, ~1 s per step there and minutes at p = 8513; the build runs the 40-wide network as its zero-padded
    64-wide twin).  Step sizes chosen so that acceptances AND rejections occur."""
    import contextlib
    import io
    d, o, hls, act, N, sigma = 1, 1, (64, 64, 64), "relu", 96, 0.2
    x, y = data(N, d, o, 0.05, 210)
    for name, kind, eps, L, nmcmc, seed in [("g13_relu_hmc.npz", "hmc", float(sys.argv[2]) if len(sys.argv) > 2 else 0.004, 3, 80, 21),
                                            ("g13_relu_mala.npz", "mala", float(sys.argv[3]) if len(sys.argv) > 3 else 0.005, 1, 80, 22),
                                            ("g13_relu_amcmc.npz", "amcmc", 0.0, 0, 50, 23)]:
        if len(sys.argv) > 4 and sys.argv[4] not in name:
            continue
        torch.manual_seed(seed)
        if kind == "amcmc":
            hls, N = (40, 40), 80
            x, y = data(N, d, o, 0.05, 211)
        solver = NN_MCMC(MLP(d, o, hls, activ=act), verbose=False)
        solver.lpinfo = {"model": nn_p, "xd": x, "yd": [yy for yy in y], "ltype": "classical", "lparams": {"sigma": sigma}}
        ini = 0.2 * (np.random.RandomState(seed + 1000).rand(solver.pdim) - 0.3)
        np.random.seed(seed)
        drawn, orig = _record_uniforms()
        mc = HMC(epsilon=eps, L=L) if kind == "hmc" else MALA(epsilon=eps) if kind == "mala" else \
            AMCMC(cov_ini=1e-5 * np.eye(solver.pdim), gamma=0.1, t0=100, tadapt=1000)
        mc.setLogPost(solver.logpost, solver.logpostgrad if kind != "amcmc" else None, lpinfo=solver.lpinfo)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                res = mc.run(param_ini=ini, nmcmc=nmcmc)
        finally:
            np.random.random_sample = orig
        acc = (res["chain"][1:] != res["chain"][:-1]).any(axis=1)
        print(name, "acceptance", acc.mean(), flush=True)
        assert 0.15 < acc.mean() < 0.95, acc.mean()
        cols = np.arange(0, solver.pdim, 34)[:256]
        save(name, dims=np.array((d,) + hls + (o,)), activ=np.array(act), x=x, y=y, sigma=sigma, seed=seed, nmcmc=nmcmc, L=L,
             epsilon=eps, param_ini=ini, accepted=acc, cols=cols, chain_cols=res["chain"][:, cols], chain_mid=res["chain"][nmcmc // 2],
             chain_final=res["chain"][-1], logpost=res["logpost"], alphas=res["alphas"], accrate=res["accrate"],
             mapparams=res["mapparams"], maxpost=res["maxpost"], uniforms=np.array(drawn), gamma=0.1, t0=100, tadapt=1000, cov_ini_diag=1e-5)


if __name__ == "__main__" and "g12" in sys.argv[1:]:
    g12()
if __name__ == "__main__" and "g13" in sys.argv[1:2]:
    g13()
if __name__ == "__main__" and "g13_viloss" in sys.argv[1:2]:
    _viloss_fixture("g13_relu_viloss.npz", "relu", 130, 212)     # (the wide int8-slice kernels with per-row activation scales)
