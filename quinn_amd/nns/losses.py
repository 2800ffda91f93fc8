"""Negative log-posterior loss.

Mirror of the reference's `NegLogPost` (quinn/nns/losses.py:152-206):
    0.5*||y - f(x)||^2 / sigma^2 + (n/2) log 2 pi + n log sigma            (n = len(predictions))
  + n / fulldatasize * [ ||w - anchor||^2 / (2 sigma_p^2) + (K/2) log(2 pi sigma_p^2) ]   if priorparams
The data term comes from the batched HIP operator (sum of squared errors of the module's CURRENT
weights); the Gaussian prior term (`NegLogPrior`, losses.py:212-256) is an O(p) host formula.
"""
import numpy as np
import torch

from ..ops import MLPArch, BatchedMLP, flatten_module, neg_log_post_from_sse


class NegLogPrior(torch.nn.Module):
    """Gaussian negative log-prior of a module's weights around an anchor (quinn/nns/losses.py:212-256):
    sum (w - anchor)^2 / (2 sigma^2) + (K/2) log(2 pi sigma^2).  O(p) host formula."""

    def __init__(self, sigma, anchor):
        super().__init__()
        self.sigma = float(sigma)
        self.anchor = anchor

    def forward(self, model):
        w = torch.cat([p.detach().flatten().double().cpu() for p in model.parameters()])
        a = torch.as_tensor(np.asarray(self.anchor.detach().cpu() if isinstance(self.anchor, torch.Tensor)
                                       else self.anchor), dtype=torch.float64).flatten()
        return ((w - a) ** 2).sum() / 2 / self.sigma ** 2 + (w.numel() / 2) * np.log(2 * np.pi * self.sigma ** 2)


class NegLogPost(torch.nn.Module):
    def __init__(self, nnmodel, fulldatasize, sigma, priorparams, device=None, dtype="float64"):
        super().__init__()
        self.nnmodel = nnmodel
        self.sigma = float(sigma)
        self.priorparams = priorparams
        self.fulldatasize = fulldatasize
        self._arch = MLPArch.from_module(nnmodel)
        self._opargs = dict(device=device, dtype=dtype)
        self._op = None

    def _operator(self, x, y):
        x = np.asarray(x.detach().cpu() if isinstance(x, torch.Tensor) else x, dtype=np.float64)
        y = np.asarray(y.detach().cpu() if isinstance(y, torch.Tensor) else y, dtype=np.float64)
        if self._op is None:
            self._op = BatchedMLP(self._arch, x, y.reshape(x.shape[0], -1), **self._opargs)
        else:
            self._op.set_data(x, y.reshape(x.shape[0], -1))
        return self._op

    def value_and_grad(self, weights, inputs, targets, want_grad=False):
        """(loss float, d loss / d weights (p,) or None) for one flat weight vector."""
        op = self._operator(inputs, targets)
        w = np.asarray(weights, dtype=np.float64).reshape(1, -1)
        n = op.N
        if want_grad:
            sse, g = op.sse_grad(w)
            grad = 0.5 * g[0].double().cpu().numpy() / self.sigma ** 2
        else:
            sse, grad = op.sse(w), None
        val = float(neg_log_post_from_sse(sse.cpu().numpy()[0], n, self.sigma))
        if self.priorparams is not None:
            sp = float(self.priorparams['sigma'])
            anchor = np.asarray(self.priorparams['anchor'].detach().cpu() if isinstance(
                self.priorparams['anchor'], torch.Tensor) else self.priorparams['anchor'], dtype=np.float64)
            K = w.shape[1]
            prior = np.sum((w[0] - anchor) ** 2) / 2 / sp ** 2 + (K / 2) * np.log(2 * np.pi * sp ** 2)
            val += n * prior / self.fulldatasize
            if want_grad:
                grad = grad + (n / self.fulldatasize) * (w[0] - anchor) / sp ** 2
        return val, grad

    def forward(self, inputs, targets):
        val, _ = self.value_and_grad(flatten_module(self.nnmodel), inputs, targets)
        return torch.tensor(val, dtype=torch.float64)
