"""Oracle: the `nnfit` training loop as used by deep-ensemble members and by VI.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  float64 torch on the CPU, one
member at a time, torch.optim.Adam / SGD as the reference.

Restates (reference file:line, relative to /root/reference):
  * nnfit loop ................ quinn/nns/nnfit.py:96-174 (losses :59-63, optimiser :74-77)
  * Learner (deepcopy + fit) .. quinn/ens/learner.py:28, 59-73
  * NN_Ens.fit / predict_ens .. quinn/solvers/nn_ens.py:51-69, 85-110
  * NN_VI.fit ................. quinn/solvers/nn_vi.py:94-113
Side effects of the reference that are not results (progress prints nnfit.py:177-192,
loss-curve PNGs :195-216) are not restated.
"""
import copy

import numpy as np
import torch

from .mlp_ref import F64, MLPSpec, build_module, load_flat
from . import vi_ref


def flat_params(module):
    return torch.cat([p.detach().flatten() for p in module.parameters()]).numpy().copy()


def train_loop(params, loss_xy, xtrn, ytrn, xval, yval, nepochs, batch_size, lrate, wd,
               optimizer, gen, snapshot):
    """The epoch / minibatch loop of nnfit.py:125-166 for an arbitrary closure
    loss_xy(x, y).  `params`: list of leaf tensors being optimised; `gen`: torch
    generator that stands for the reference's global CPU generator (randperm);
    `snapshot()` returns what to keep when validation improves (deepcopy in the
    reference, nnfit.py:149-156)."""
    ntrn = xtrn.shape[0]
    if batch_size is None or batch_size > ntrn:
        batch_size = ntrn
    if optimizer == "adam":
        opt = torch.optim.Adam(params, lr=lrate, weight_decay=wd)
    else:
        opt = torch.optim.SGD(params, lr=lrate, weight_decay=wd)
    info = {"best_fepoch": 0, "best_epoch": 0, "best_loss": 1.e+100, "best": snapshot(),
            "history": [], "perms": []}
    fepoch = 0
    for t in range(nepochs):
        perm = torch.randperm(ntrn, generator=gen)
        info["perms"].append(perm.numpy().copy())
        nsub = len(range(0, ntrn, batch_size))
        for i in range(0, ntrn, batch_size):
            idx = perm[i:i + batch_size]
            loss_trn = loss_xy(xtrn[idx, :], ytrn[idx, :])
            with torch.no_grad():
                loss_val = loss_xy(xval, yval)
            if i == 0:
                with torch.no_grad():
                    loss_full = loss_xy(xtrn, ytrn)
            fepoch += 1. / nsub
            crit = loss_val.item()
            info["history"].append([fepoch + 0.0, loss_trn.item(), loss_full.item(), crit])
            if crit < info["best_loss"]:
                info["best_loss"] = crit
                info["best"] = snapshot()
                info["best_fepoch"] = fepoch
                info["best_epoch"] = t
            opt.zero_grad()
            loss_trn.backward()
            opt.step()
    info["history"] = np.array(info["history"])
    return info


def fit_member_mse(spec, w0, xtrn, ytrn, xval, yval, nepochs, batch_size, lrate, gen,
                   wd=0.0, optimizer="adam"):
    """One ensemble member: module initialised from flat w0, MSELoss(mean) (nnfit.py:59-63)."""
    mod = build_module(spec)
    load_flat(mod, w0)
    xt, yt = torch.as_tensor(xtrn, dtype=F64), torch.as_tensor(ytrn, dtype=F64)
    xv, yv = torch.as_tensor(xval, dtype=F64), torch.as_tensor(yval, dtype=F64)
    mse = torch.nn.MSELoss(reduction="mean")
    info = train_loop(list(mod.parameters()), lambda a, b: mse(mod(a), b), xt, yt, xv, yv,
                      nepochs, batch_size, lrate, wd, optimizer, gen, lambda: flat_params(mod))
    info["final"] = flat_params(mod)
    return info


def neg_log_post_with_prior(mod, x, y, sigma, fulldatasize, anchor, prior_sigma):
    """NegLogPost.forward with a Gaussian prior (losses.py:197-204 + NegLogPrior.forward :238-256): the prior is
    summed parameter tensor by parameter tensor and weighted len(batch) / fulldatasize."""
    sig = torch.tensor(float(sigma), dtype=F64)
    pi = torch.tensor(np.pi, dtype=F64)
    pred = mod(x)
    val = 0.5 * torch.sum(torch.pow(y - pred, 2)) / sig ** 2
    val = val + (len(pred) / 2) * torch.log(2 * pi)
    val = val + len(pred) * torch.log(sig)
    if anchor is not None:
        ps = torch.tensor(float(prior_sigma), dtype=F64)
        nlp, i = 0, 0
        for p in mod.parameters():
            n = p.flatten().size()[0]
            nlp = nlp + torch.sum(torch.pow(p.flatten() - anchor[i:i + n], 2)) / 2 / ps ** 2
            i += n
        nlp = nlp + (i / 2) * torch.log(2 * pi * ps ** 2)
        val = val + len(pred) * nlp / fulldatasize
    return val


def fit_member_logpost(spec, w0, xtrn, ytrn, xval, yval, nepochs, batch_size, lrate, gen, datanoise,
                       anchor=None, prior_sigma=None, wd=0.0):
    """nnfit(loss_fn='logpost', datanoise, priorparams) for one module (nnfit.py:64-66): NegLogPost over the
    member's ntrn rows, optionally with the anchored Gaussian prior."""
    mod = build_module(spec)
    load_flat(mod, w0)
    xt, yt = torch.as_tensor(xtrn, dtype=F64), torch.as_tensor(ytrn, dtype=F64)
    xv, yv = torch.as_tensor(xval, dtype=F64), torch.as_tensor(yval, dtype=F64)
    a = None if anchor is None else torch.as_tensor(anchor, dtype=F64)
    ntrn = xt.shape[0]
    info = train_loop(list(mod.parameters()),
                      lambda xb, yb: neg_log_post_with_prior(mod, xb, yb, datanoise, ntrn, a, prior_sigma),
                      xt, yt, xv, yv, nepochs, batch_size, lrate, wd, "adam", gen, lambda: flat_params(mod))
    info["final"] = flat_params(mod)
    return info


def fit_rms(spec, w0, xtrn, ytrn, xval, yval, nens, dfrac, nepochs, batch_size, lrate, np_rng, gen, datanoise,
            priorsigma):
    """NN_RMS.fit (nn_rms.py:41-57): per member, in this order: np permutation -> data subset; anchor =
    randn(p) * priorsigma from the global torch generator; then nnfit with the 'logpost' loss and that prior."""
    members = []
    ntrn = ytrn.shape[0]
    for _ in range(nens):
        rows = np_rng.permutation(ntrn)[:int(ntrn * dfrac)]
        anchor = torch.randn(size=(spec.nparams,), dtype=F64, generator=gen) * priorsigma
        xv, yv = (xtrn[rows], ytrn[rows]) if xval is None else (xval, yval)
        info = fit_member_logpost(spec, w0, xtrn[rows], ytrn[rows], xv, yv, nepochs, batch_size, lrate, gen,
                                  datanoise, anchor=anchor, prior_sigma=priorsigma)
        info["rows"], info["anchor"] = rows, anchor.numpy().copy()
        members.append(info)
    return members


def fit_ensemble(spec, w0, xtrn, ytrn, xval, yval, nens, dfrac, nepochs, batch_size, lrate,
                 np_rng, gen, wd=0.0):
    """NN_Ens.fit: members are deep copies of ONE module (identical w0, learner.py:28);
    member j trains on rows np_rng.permutation(ntrn)[:int(ntrn*dfrac)] (nn_ens.py:63-64),
    members strictly one after another (so both RNG streams are consumed member-major).
    xval None: each member validates on a copy of its own training subset (nnfit.py:106-109)."""
    members = []
    ntrn = ytrn.shape[0]
    for _ in range(nens):
        rows = np_rng.permutation(ntrn)[:int(ntrn * dfrac)]
        xv, yv = (xtrn[rows], ytrn[rows]) if xval is None else (xval, yval)
        info = fit_member_mse(spec, w0, xtrn[rows], ytrn[rows], xv, yv, nepochs, batch_size,
                              lrate, gen, wd=wd)
        info["rows"] = rows
        members.append(info)
    return members


def fit_vi(spec, mu0, rho0, xtrn, ytrn, xval, yval, nepochs, batch_size, lrate, nsam, datanoise,
           gen, wd=0.0, prior=None):
    """NN_VI.fit: nnfit with loss_xy = viloss; every loss evaluation (train batch, validation,
    full train) draws nsam fresh samples from `gen` (the model is never put in eval mode)."""
    prior = prior or {}
    nb = vi_ref.num_batches(xtrn.shape[0], batch_size)
    shapes = vi_ref.param_shapes(spec)
    mu_t = torch.tensor(mu0, dtype=F64, requires_grad=True)
    rho_t = torch.tensor(rho0, dtype=F64, requires_grad=True)
    # the reference optimises one (mu, rho) Parameter per tensor; Adam is elementwise, so a
    # flat leaf gives the same trajectory
    xt, yt = torch.as_tensor(xtrn, dtype=F64), torch.as_tensor(ytrn, dtype=F64)
    xv, yv = torch.as_tensor(xval, dtype=F64), torch.as_tensor(yval, dtype=F64)
    eps_log = []

    def loss_xy(a, b):
        eps = vi_ref.draw_eps(spec, nsam, gen)
        eps_log.append(eps)
        return _viloss_attached(spec, mu_t, rho_t, eps, a, b, datanoise, nb, shapes, **prior)

    info = train_loop([mu_t, rho_t], loss_xy, xt, yt, xv, yv, nepochs, batch_size, lrate, wd,
                      "adam", gen, lambda: (mu_t.detach().numpy().copy(), rho_t.detach().numpy().copy()))
    info["final"] = (mu_t.detach().numpy().copy(), rho_t.detach().numpy().copy())
    info["eps"] = eps_log
    return info


def _viloss_attached(spec, mu_t, rho_t, eps, x_t, y_t, datanoise, nb, shapes,
                     pi=0.5, sigma1=1.0, sigma2=1.0):
    """vi_ref.elbo_terms, but attached to caller-owned leaves (for the optimiser)."""
    import math
    S, B, o = eps.shape[0], x_t.shape[0], y_t.shape[1]
    n1 = torch.distributions.Normal(torch.tensor(0.0, dtype=F64), torch.tensor(float(sigma1), dtype=F64))
    n2 = torch.distributions.Normal(torch.tensor(0.0, dtype=F64), torch.tensor(float(sigma2), dtype=F64))
    outputs = torch.zeros(S, B, o, dtype=F64)
    lps = torch.zeros(S, dtype=F64)
    lqs = torch.zeros(S, dtype=F64)
    for s in range(S):
        off, ws, ms, sgs = 0, [], [], []
        for shp in shapes:
            n = int(np.prod(shp))
            m = mu_t[off:off + n].view(shp)
            sg = torch.exp(rho_t[off:off + n].view(shp))
            e = torch.tensor(eps[s, off:off + n], dtype=F64).view(shp)
            ws.append(m + sg * e); ms.append(m); sgs.append(sg)
            off += n
        lp = 0.0
        for w in ws:
            lp = lp + (torch.log(pi * torch.exp(n1.log_prob(w)) + (1 - pi) * torch.exp(n2.log_prob(w)))).sum()
        lq = 0.0
        for w, m, sg in zip(ws, ms, sgs):
            lq = lq + (-math.log(math.sqrt(2 * math.pi)) - torch.log(sg) - ((w - m) ** 2) / (2 * sg ** 2)).sum()
        outputs[s] = vi_ref._functional_forward(spec, ws, x_t)
        lps[s], lqs[s] = lp, lq
    dsig = torch.tensor([datanoise], dtype=F64)
    nll = (B * torch.log(dsig) + 0.5 * B * torch.log(2.0 * torch.tensor(math.pi, dtype=F64))
           + 0.5 * B * ((outputs - y_t) ** 2).mean() / dsig ** 2)
    return ((lqs.mean() - lps.mean()) / nb + nll).squeeze()


def fit_member_plateau(spec, w0, xtrn, ytrn, xval, yval, nepochs, batch_size, lrate, gen, cooldown, factor):
    """fit_member_mse with torch's ReduceLROnPlateau stepped once per epoch on the epoch's last
    validation loss (nnfit.py:91-92, 170-172)."""
    mod = build_module(spec)
    load_flat(mod, w0)
    xt, yt = torch.as_tensor(xtrn, dtype=F64), torch.as_tensor(ytrn, dtype=F64)
    xv, yv = torch.as_tensor(xval, dtype=F64), torch.as_tensor(yval, dtype=F64)
    mse = torch.nn.MSELoss(reduction="mean")
    opt = torch.optim.Adam(mod.parameters(), lr=lrate)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode='min', cooldown=cooldown, factor=factor)
    ntrn = xt.shape[0]
    bs = ntrn if batch_size is None or batch_size > ntrn else batch_size
    hist, lrs = [], []
    for t in range(nepochs):
        perm = torch.randperm(ntrn, generator=gen)
        for i in range(0, ntrn, bs):
            idx = perm[i:i + bs]
            loss = mse(mod(xt[idx]), yt[idx])
            with torch.no_grad():
                lv = mse(mod(xv), yv)
            hist.append([loss.item(), lv.item()])
            opt.zero_grad(); loss.backward(); opt.step()
        sched.step(hist[-1][1])
        lrs.append(opt.param_groups[0]['lr'])
    return {"history": np.array(hist), "lrs": np.array(lrs), "final": flat_params(mod)}
