"""Metropolis-adjusted Langevin for C lock-step chains.

Mirror of the reference's `MALA` (quinn/mcmc/mala.py:8-53): one Langevin step, two
gradient evaluations per proposal (each ONE batched call).  Unlike the reference
(where `sampler='mala'` falls through NN_MCMC.fit's dispatch, nn_mcmc.py:130-135, and
raises UnboundLocalError) `NN_MCMC.fit(sampler='mala')` reaches this class.
"""
import numpy as np

from .mcmc import MCMCBase
from .hmc import _kinetic


class MALA(MCMCBase):
    """Args: epsilon (float): step size (default 0.05)."""

    def __init__(self, epsilon=0.05):
        super().__init__()
        self.epsilon = epsilon

    def sampler_batch(self, current, imcmc):
        assert self.logPostGrad is not None or self.logPostGradBatch is not None
        C, p = current.shape
        eps = self.epsilon
        q = current.copy()
        mom = np.stack([self.rngs[c].randn(p) for c in range(C)])
        g_cur = self._lpg(current)
        q += 0.5 * eps ** 2 * g_cur + eps * mom
        g_prop = self._lpg(q)
        k_cur = _kinetic(mom)
        mom += eps * (g_cur + g_prop) / 2
        return q, k_cur, _kinetic(mom)
