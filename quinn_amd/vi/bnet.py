"""Mean-field Gaussian Bayesian network; all MC samples of the ELBO evaluated in one batch.

Mirror of the reference's `BNet` (quinn/vi/bnet.py:10-232): per parameter tensor a variational
mean mu ~ U[mu_lo, mu_hi] and rho ~ U[rho_lo, rho_hi] with sigma = exp(rho) (bnet.py:69-80), a
zero-mean two-Gaussian mixture prior, `sample_elbo` -> (log_prior, log_q, NLL) and
`viloss = (log_q - log_prior)/num_batches + NLL` (bnet.py:181-232).  The reference draws one
weight sample at a time and pushes it through the module in a Python loop; here the S samples are
rows of a `[S, p]` matrix produced by `qn_vi_sample_kl`, pushed through the batched MLP kernels,
and the gradient w.r.t. (mu, rho) comes from `qn_mlp_sse_fwdbwd` + `qn_vi_grad` wrapped in a
`torch.autograd.Function`, so `loss.backward()` works as in the reference.

The variational parameters live in ONE flat float64 CUDA parameter `theta = [mu (p), rho (p)]`
in the reference's flat weight order.  Random draws: with `rng='reference'` the standard normals
are drawn on the host from torch's global CPU generator, sample by sample and parameter tensor by
parameter tensor (the reference's consumption order, rvs.py:107); `rng='device'` draws on the GPU.
"""
import copy
import ctypes

import numpy as np
import torch

from .. import _lib
from ..ops import MLPArch, BatchedMLP


def _param_shapes(arch):
    return arch.param_shapes()


class _ViLoss(torch.autograd.Function):
    """viloss(theta) with the forward (and, when grad is enabled, the backward) on the HIP kernels."""

    @staticmethod
    def forward(ctx, theta, bnet, x, y, nsam, datanoise, num_batches, grad_mode):
        p = bnet.p
        mu, rho = theta.detach()[:p], theta.detach()[p:]
        eps = bnet._draw_eps(nsam)
        # grad mode is off inside forward() and needs_input_grad ignores torch.no_grad(): the caller passes
        # the mode it saw, so that the validation / full-train evaluations of nnfit (no_grad) stay forward-only
        need_grad = bool(grad_mode and ctx.needs_input_grad[0])
        lp, lq, nll, W, gW = bnet._elbo(mu, rho, eps, x, y, nsam, datanoise, need_grad)
        loss = (lq - lp) / num_batches + nll
        ctx.bnet, ctx.nsam, ctx.datanoise, ctx.num_batches = bnet, nsam, datanoise, num_batches
        ctx.o = y.shape[1]
        ctx.save_for_backward(mu, rho, eps, gW if gW is not None else torch.empty(0, device=theta.device))
        return loss

    @staticmethod
    def backward(ctx, gout):
        mu, rho, eps, gW = ctx.saved_tensors
        bnet = ctx.bnet
        p = bnet.p
        L = _lib.lib()
        dtheta = torch.empty(2 * p, dtype=torch.float64, device=mu.device)
        gw_scale = 0.5 / (ctx.nsam * ctx.o * ctx.datanoise ** 2)      # dNLL/dSSE_s, bnet.py:215
        kl_scale = 1.0 / ctx.num_batches
        st = ctypes.c_void_p(torch.cuda.current_stream(mu.device).cuda_stream)
        with torch.cuda.device(mu.device):
            _lib.check(L.qn_vi_grad(mu.data_ptr(), rho.data_ptr(), eps.data_ptr(), gW.data_ptr(), ctx.nsam, p,
                                    bnet.pi, bnet.sigma1, bnet.sigma2, gw_scale, kl_scale, bnet.op.qdt,
                                    dtheta.data_ptr(), dtheta[p:].data_ptr(), st), "qn_vi_grad")
        return dtheta * gout, None, None, None, None, None, None, None


class BNet(torch.nn.Module):
    def __init__(self, nnmodel, pi=0.5, sigma1=1.0, sigma2=1.0, mu_init_lower=-0.2, mu_init_upper=0.2,
                 rho_init_lower=-5.0, rho_init_upper=-4.0, device=None, dtype="float64", rng="reference"):
        super().__init__()
        assert isinstance(nnmodel, torch.nn.Module)
        self.arch = MLPArch.from_module(nnmodel)
        self.p = self.arch.nparams
        self.pi, self.sigma1, self.sigma2 = float(pi), float(sigma1), float(sigma2)
        self.rng = rng
        # one operator; its dataset is swapped per call (minibatch / validation / full)
        self.op = BatchedMLP(self.arch, np.zeros((1, self.arch.dims[0])), None, device=device, dtype=dtype)
        self.device = self.op.device
        mus, rhos = [], []
        for shp in _param_shapes(self.arch):                     # bnet.py:69-72: mu then rho, per tensor
            mus.append(torch.empty(shp, dtype=torch.float64).uniform_(mu_init_lower, mu_init_upper).flatten())
            rhos.append(torch.empty(shp, dtype=torch.float64).uniform_(rho_init_lower, rho_init_upper).flatten())
        # ONE flat parameter [mu (p), rho (p)]; its name keeps `mu` / `rho` visible in named_parameters()
        # (the reference registers one `*_mu` / `*_rho` pair per tensor, bnet.py:69-72)
        self.mu_rho = torch.nn.Parameter(torch.cat(mus + rhos).to(self.device))
        self.log_prior = 0.0
        self.log_variational_posterior = 0.0
        self.loss_params = None
        self.nparams = len(_param_shapes(self.arch))

    def __deepcopy__(self, memo):
        new = BNet.__new__(BNet)
        torch.nn.Module.__init__(new)
        for k in ('arch', 'p', 'pi', 'sigma1', 'sigma2', 'rng', 'op', 'device', 'nparams', 'loss_params'):
            setattr(new, k, getattr(self, k))
        new.mu_rho = torch.nn.Parameter(self.mu_rho.detach().clone())
        new.log_prior, new.log_variational_posterior = self.log_prior, self.log_variational_posterior
        return new

    @property
    def theta(self):
        """The flat variational parameter [mu, rho] (alias of `mu_rho`)."""
        return self.mu_rho

    @property
    def mu(self):
        return self.theta.detach()[:self.p]

    @property
    def rho(self):
        return self.theta.detach()[self.p:]

    # -- random draws ---------------------------------------------------------------------------
    def _draw_eps(self, nsam):
        if self.rng == "device":
            return torch.randn(nsam, self.p, dtype=torch.float64, device=self.device)
        out = np.empty((nsam, self.p))
        for s in range(nsam):
            off = 0
            for shp in _param_shapes(self.arch):
                n = int(np.prod(shp))
                out[s, off:off + n] = torch.normal(torch.zeros(shp, dtype=torch.float64),
                                                   torch.ones(shp, dtype=torch.float64)).flatten().numpy()
                off += n
        return torch.as_tensor(out, device=self.device)

    # -- kernels ------------------------------------------------------------------------------------
    def _sample_kl(self, mu, rho, eps):
        S = eps.shape[0]
        L = _lib.lib()
        W = torch.empty(S, self.p, dtype=self.op.tdt, device=self.device)
        lq = torch.empty(S, dtype=torch.float64, device=self.device)
        lp = torch.empty(S, dtype=torch.float64, device=self.device)
        st = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        with torch.cuda.device(self.device):
            _lib.check(L.qn_vi_sample_kl(mu.contiguous().data_ptr(), rho.contiguous().data_ptr(), eps.data_ptr(), S,
                                         self.p, self.pi, self.sigma1, self.sigma2, self.op.qdt, W.data_ptr(),
                                         lq.data_ptr(), lp.data_ptr(), st), "qn_vi_sample_kl")
        return W, lq, lp

    def _elbo(self, mu, rho, eps, x, y, nsam, datanoise, need_grad):
        W, lq, lp = self._sample_kl(mu, rho, eps)
        self.op.set_data(x, y)
        B, o = x.shape[0], y.shape[1]
        if need_grad:
            sse, gW = self.op.sse_grad(W)
        else:
            sse, gW = self.op.sse(W), None
        dn = torch.tensor(float(datanoise), dtype=torch.float64, device=self.device)
        mean_sq = sse.sum() / (nsam * B * o)                    # ((outputs - target)**2).mean(), bnet.py:215
        nll = B * torch.log(dn) + 0.5 * B * np.log(2.0 * np.pi) + 0.5 * B * mean_sq / dn ** 2
        return lp.mean(), lq.mean(), nll, W, gW

    # -- reference API ---------------------------------------------------------------------------------
    def forward(self, x, sample=False, par_samples=None):
        """Prediction `(N,o)` (device tensor) with one sampled weight vector (training mode or
        sample=True), with given `par_samples` (flat `(p,)`), or with the variational mean."""
        xt = torch.as_tensor(x, dtype=torch.float64, device=self.device)
        if self.training or sample:
            assert par_samples is None
            W, lq, lp = self._sample_kl(self.mu, self.rho, self._draw_eps(1))
            if self.training:
                self.log_prior, self.log_variational_posterior = lp[0], lq[0]
        else:
            w = self.mu if par_samples is None else torch.as_tensor(par_samples, device=self.device)
            W = w.reshape(1, -1).to(self.op.tdt)
            self.log_prior, self.log_variational_posterior = 0, 0
        return self.op.predict(W, xt)[0].double()

    def sample_elbo(self, x, target, nsam, likparams=None):
        """(log_prior, log_variational_posterior, negative_log_likelihood): float64 0-d tensors."""
        xt = torch.as_tensor(x, dtype=torch.float64, device=self.device)
        yt = torch.as_tensor(target, dtype=torch.float64, device=self.device)
        assert xt.shape[0] == yt.shape[0]
        lp, lq, nll, _, _ = self._elbo(self.mu, self.rho, self._draw_eps(nsam), xt, yt, nsam, likparams[0], False)
        return lp, lq, nll

    def viloss(self, data, target):
        """`(log_q - log_prior)/num_batches + NLL`; differentiable w.r.t. `self.theta`."""
        datanoise, nsam, num_batches = self.loss_params
        xt = torch.as_tensor(data, dtype=torch.float64, device=self.device)
        yt = torch.as_tensor(target, dtype=torch.float64, device=self.device)
        return _ViLoss.apply(self.theta, self, xt, yt, int(nsam), float(datanoise), num_batches,
                             torch.is_grad_enabled())
