"""Adaptive Metropolis (Haario et al. 2001) for C lock-step chains.

Mirror of the reference's `AMCMC` (quinn/mcmc/admcmc.py:7-74): same constructor, same
recursion for the running mean / covariance, same adaptation rule, and the same draw --
numpy's legacy `multivariate_normal`, i.e. `z @ (sqrt(s)[:,None] * v)` with
`(u, s, v) = svd(propcov)` -- so chains are bit-identical to the reference's.  The SVD
factor is cached per chain and recomputed only when the proposal covariance changes
(the reference refactorises it on every draw, which is where 98 % of its time goes at
p = 321); `exact_mvn=True` calls `multivariate_normal` itself instead.
"""
import numpy as np
from numpy.linalg import svd

from .mcmc import MCMCBase


class AMCMC(MCMCBase):
    """Adaptive MCMC.

    Args:
        cov_ini (np.ndarray, optional): initial proposal covariance `(p,p)`; default
            `0.01 + diag(0.09*|x0|)` (0.01 is added to EVERY entry, admcmc.py:65).
        gamma (float): proposal scale factor (default 0.1).
        t0 (int): step after which adaptation may start (default 100).
        tadapt (int): adapt every `tadapt` steps (default 1000).
        exact_mvn (bool): build-only switch, see module docstring.
    """

    def __init__(self, cov_ini=None, gamma=0.1, t0=100, tadapt=1000, exact_mvn=False):
        super().__init__()
        self.cov_ini = cov_ini
        self.t0 = t0
        self.tadapt = tadapt
        self.gamma = gamma
        self.exact_mvn = exact_mvn
        self._Xm = None        # [C] running means
        self._cov = None       # [C] running covariances (p,p)
        self._propcov = None   # [C] proposal covariances
        self._factor = None    # [C] cached sqrt(s)[:,None]*v of _propcov

    def _draw(self, c, p):
        rng = self.rngs[c]
        if self.exact_mvn:
            return rng.multivariate_normal(np.zeros(p,), self._propcov[c])
        z = rng.standard_normal((p,)).reshape(-1, p)
        if self._factor[c] is None:
            (_, s, v) = svd(self._propcov[c].astype(np.double))
            self._factor[c] = np.sqrt(s)[:, None] * v
        x = np.dot(z, self._factor[c])
        x += np.zeros(p,)
        return x.reshape(p)

    def sampler_batch(self, current, imcmc):
        C, p = current.shape
        if imcmc == 0 or self._Xm is None or len(self._Xm) != C:
            self._Xm = [None] * C
            self._cov = [None] * C
            self._propcov = [None] * C
            self._factor = [None] * C
        prop = current.copy()
        for c in range(C):
            x = current[c]
            if imcmc == 0:
                self._Xm[c] = x.copy()
                self._cov[c] = np.zeros((p, p))
                self._propcov[c] = self.cov_ini if self.cov_ini is not None \
                    else 0.01 + np.diag(0.09 * np.abs(x))
                self._factor[c] = None
            else:
                self._Xm[c] = (imcmc * self._Xm[c] + x) / (imcmc + 1.0)
                keep = (imcmc - 1.0) / imcmc
                gain = (imcmc + 1.0) / imcmc ** 2
                dev = x - self._Xm[c]
                self._cov[c] = keep * self._cov[c] + gain * np.dot(dev.reshape(p, 1), dev.reshape(1, p))
                if imcmc > self.t0 and imcmc % self.tadapt == 0:
                    self._propcov[c] = (self.gamma * 2.4 ** 2 / p) * (self._cov[c] + 10 ** (-8) * np.eye(p))
                    self._factor[c] = None
            prop[c] += self._draw(c, p)
        zeros = np.zeros(C)
        return prop, zeros, zeros
