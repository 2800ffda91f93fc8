"""Hamiltonian Monte Carlo for C lock-step chains.

Mirror of the reference's `HMC` (quinn/mcmc/hmc.py:8-70): momentum ~ N(0, I), half kick,
L position steps with L-1 inner kicks, half kick, negate; L+1 gradient evaluations per
proposal -- each of them ONE batched call for all chains.
"""
import numpy as np

from .mcmc import MCMCBase


def _kinetic(P):
    # row by row on contiguous 1-D views: the same pairwise summation the reference's
    # np.sum(np.square(p)) performs on its 1-D momentum
    return np.array([np.sum(np.square(P[c])) / 2 for c in range(P.shape[0])])


class HMC(MCMCBase):
    """Args: epsilon (float): leapfrog step size (default 0.05); L (int): leapfrog steps (default 3)."""

    def __init__(self, epsilon=0.05, L=3):
        super().__init__()
        self.epsilon = epsilon
        self.L = L

    def sampler_batch(self, current, imcmc):
        assert self.logPostGrad is not None or self.logPostGradBatch is not None
        C, p = current.shape
        eps = self.epsilon
        q = current.copy()
        mom = np.stack([self.rngs[c].randn(p) for c in range(C)])
        k_cur = _kinetic(mom)
        mom += eps * self._lpg(q) / 2
        for j in range(self.L):
            q += eps * mom
            if j != self.L - 1:
                mom += eps * self._lpg(q)
        mom += eps * self._lpg(q) / 2
        mom = -mom
        return q, k_cur, _kinetic(mom)
