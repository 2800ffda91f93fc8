"""Oracle: mean-field Gaussian VI (Bayes-by-backprop) ELBO Monte-Carlo estimator.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  float64 torch on the CPU; the
Python loop over MC samples is the reference's; gradients come from autograd as
in the reference.

Restates (reference file:line, relative to /root/reference):
  * BNet.__init__ (mu ~ U[lo,hi], rho ~ U[lo,hi], per parameter tensor, mu first)
        ........................................ quinn/vi/bnet.py:62-90
  * BNet.forward (sample, log_prior, log_q) .... quinn/vi/bnet.py:142-178
  * BNet.sample_elbo / viloss .................. quinn/vi/bnet.py:181-217, 219-232
  * Gaussian_1d.sample / log_prob (sigma = exp(rho) because BNet passes
    logsigma=rho, bnet.py:80) .................. quinn/rvar/rvs.py:96-127
  * GMM2_1d.log_prob ........................... quinn/rvar/rvs.py:159-173
  * NN_VI.fit num_batches ...................... quinn/solvers/nn_vi.py:94-100
"""
import math

import numpy as np
import torch

from .mlp_ref import F64, MLPSpec, build_module


def param_shapes(spec: MLPSpec):
    """Shapes in named_parameters() order: W_0, b_0, W_1, b_1, ... (an RNetSpec lists its own)."""
    if hasattr(spec, "param_shapes"):
        return spec.param_shapes()
    shapes = []
    for a, b in zip(spec.dims[:-1], spec.dims[1:]):
        shapes.append((b, a))
        if spec.bias:
            shapes.append((b,))
    return shapes


def init_variational(spec, gen, mu_lo=-0.2, mu_hi=0.2, rho_lo=-5.0, rho_hi=-4.0):
    """Draw (mu, rho) per parameter tensor from generator `gen` in the reference's
    order (bnet.py:69-72: mu then rho, tensor by tensor).  Returns flat [p] arrays."""
    mus, rhos = [], []
    for shp in param_shapes(spec):
        mus.append(torch.empty(shp, dtype=F64).uniform_(mu_lo, mu_hi, generator=gen).flatten())
        rhos.append(torch.empty(shp, dtype=F64).uniform_(rho_lo, rho_hi, generator=gen).flatten())
    return torch.cat(mus).numpy().copy(), torch.cat(rhos).numpy().copy()


def draw_eps(spec, nsam, gen):
    """Standard normals for nsam MC samples in consumption order: sample-major, then
    parameter tensor by tensor (bnet.py:145-146 -> rvs.py:107).  torch's CPU normal_
    takes different code paths below / from 16 elements, so the draw is made per
    tensor with the tensor's own shape.  Returns [nsam, p]."""
    out = np.empty((nsam, spec.nparams))
    for s in range(nsam):
        off = 0
        for shp in param_shapes(spec):
            n = int(np.prod(shp))
            z = torch.normal(torch.zeros(shp, dtype=F64), torch.ones(shp, dtype=F64), generator=gen)
            out[s, off:off + n] = z.flatten().numpy()
            off += n
    return out


def _functional_forward(spec, tensors, x):
    if hasattr(spec, "functional_forward"):
        return spec.functional_forward(tensors, x)
    h = x
    nl = len(spec.dims) - 1
    k = 0
    for i in range(nl):
        W = tensors[k]; k += 1
        b = None
        if spec.bias:
            b = tensors[k]; k += 1
        h = torch.nn.functional.linear(h, W, b)
        if i < nl - 1:
            if spec.activ == "tanh":
                h = torch.tanh(h)
            elif spec.activ == "relu":
                h = torch.relu(h)
    return h


def elbo_terms(spec, mu, rho, eps, x, y, datanoise, pi=0.5, sigma1=1.0, sigma2=1.0):
    """(log_prior, log_q, nll) as tensors attached to leaf tensors mu_t, rho_t.
    mu, rho: [p]; eps: [S, p] (the draws the reference would have made); x: [B, d]; y: [B, o]."""
    mu_t = torch.tensor(mu, dtype=F64, requires_grad=True)
    rho_t = torch.tensor(rho, dtype=F64, requires_grad=True)
    x_t = torch.as_tensor(x, dtype=F64)
    y_t = torch.as_tensor(y, dtype=F64)
    S, B, o = eps.shape[0], x_t.shape[0], y_t.shape[1]
    shapes = param_shapes(spec)
    n1 = torch.distributions.Normal(torch.tensor(0.0, dtype=F64), torch.tensor(float(sigma1), dtype=F64))
    n2 = torch.distributions.Normal(torch.tensor(0.0, dtype=F64), torch.tensor(float(sigma2), dtype=F64))
    outputs = torch.zeros(S, B, o, dtype=F64)
    lps = torch.zeros(S, dtype=F64)
    lqs = torch.zeros(S, dtype=F64)
    for s in range(S):
        off = 0
        ws = []
        lp = 0.0
        lq = 0.0
        for shp in shapes:
            n = int(np.prod(shp))
            m = mu_t[off:off + n].view(shp)
            r = rho_t[off:off + n].view(shp)
            e = torch.tensor(eps[s, off:off + n], dtype=F64).view(shp)
            sig = torch.exp(r)                                   # rvs.py:105
            w = m + sig * e                                      # rvs.py:108
            ws.append(w)
            off += n
        for w in ws:                                             # bnet.py:157-159
            p1 = torch.exp(n1.log_prob(w))
            p2 = torch.exp(n2.log_prob(w))
            lp = lp + (torch.log(pi * p1 + (1 - pi) * p2)).sum()
        off = 0
        for w, shp in zip(ws, shapes):                           # bnet.py:161-163
            n = int(np.prod(shp))
            m = mu_t[off:off + n].view(shp)
            r = rho_t[off:off + n].view(shp)
            sig = torch.exp(r)
            lq = lq + (-math.log(math.sqrt(2 * math.pi)) - torch.log(sig)
                       - ((w - m) ** 2) / (2 * sig ** 2)).sum()
            off += n
        outputs[s] = _functional_forward(spec, ws, x_t)
        lps[s] = lp
        lqs[s] = lq
    log_prior = lps.mean()
    log_q = lqs.mean()
    dsig = torch.tensor([datanoise], dtype=F64)
    nll = (B * torch.log(dsig) + 0.5 * B * torch.log(2.0 * torch.tensor(math.pi, dtype=F64))
           + 0.5 * B * ((outputs - y_t) ** 2).mean() / dsig ** 2)   # bnet.py:215
    return mu_t, rho_t, log_prior, log_q, nll, outputs


def viloss(spec, mu, rho, eps, x, y, datanoise, num_batches, want_grad=True, **prior):
    """viloss = (log_q - log_prior)/num_batches + nll (bnet.py:232) and its gradient
    w.r.t. (mu, rho).  Returns dict of numpy values."""
    mu_t, rho_t, lp, lq, nll, outs = elbo_terms(spec, mu, rho, eps, x, y, datanoise, **prior)
    loss = (lq - lp) / num_batches + nll
    res = {"loss": loss.item(), "log_prior": lp.item(), "log_q": lq.item(), "nll": nll.item(),
           "outputs": outs.detach().numpy()}
    if want_grad:
        loss.backward()
        res["dmu"] = mu_t.grad.numpy().copy()
        res["drho"] = rho_t.grad.numpy().copy()
    return res


def num_batches(ntrn, batch_size):
    """nn_vi.py:94-100."""
    if batch_size is None or batch_size > ntrn:
        batch_size = ntrn
    return ntrn if batch_size == 1 else (ntrn + 1) // batch_size
