"""Log-densities used by the variational path, as closed forms on numpy / torch values.

Mirror of the reference's `Gaussian_1d` and `GMM2_1d` (quinn/rvar/rvs.py:51-173) for the two
members the hot path uses: `log_prob` of a mean-field Gaussian parameterised by
(mu, log sigma) -- `BNet` passes `logsigma=rho`, i.e. sigma = exp(rho) (bnet.py:80) -- and of a
two-component zero-mean Gaussian mixture.  The device kernels (`qn_vi_sample_kl`) compute the same
sums for all MC samples at once; these host classes exist for API parity and tests.
"""
import math

import torch


class RV(torch.nn.Module):
    def sample(self, num_samples=1):
        raise NotImplementedError

    def log_prob(self, x):
        raise NotImplementedError


class Gaussian_1d(RV):
    def __init__(self, mu, rho=None, logsigma=None):
        super().__init__()
        self.mu = mu
        self.rho, self.logsigma = None, None
        if rho is not None:
            assert logsigma is None and rho.shape == mu.shape
            self.rho = rho
        else:
            assert logsigma is not None and logsigma.shape == mu.shape
            self.logsigma = logsigma

    def _sigma(self):
        return torch.log1p(torch.exp(self.rho)) if self.rho is not None else torch.exp(self.logsigma)

    def sample(self):
        sigma = self._sigma()
        eps = torch.normal(torch.zeros(sigma.shape, dtype=sigma.dtype), torch.ones(sigma.shape, dtype=sigma.dtype))
        return self.mu + sigma * eps.to(self.mu.device)

    def log_prob(self, x):
        sigma = self._sigma()
        return (-math.log(math.sqrt(2 * math.pi)) - torch.log(sigma) - ((x - self.mu) ** 2) / (2 * sigma ** 2)).sum()


class GMM2_1d(RV):
    def __init__(self, pi, sigma1, sigma2):
        super().__init__()
        self.pi, self.sigma1, self.sigma2 = pi, sigma1, sigma2

    @staticmethod
    def _npdf(x, s):
        return torch.exp(-(x ** 2) / (2 * s ** 2) - math.log(s) - math.log(math.sqrt(2 * math.pi)))

    def log_prob(self, x):
        return torch.log(self.pi * self._npdf(x, self.sigma1) + (1 - self.pi) * self._npdf(x, self.sigma2)).sum()
