"""Metropolis-Hastings stepper for C independent chains advanced in lock-step.

Mirror of the reference's `MCMCBase` (quinn/mcmc/mcmc.py:10-115): same `setLogPost` /
`run` / `sampler` interface and the same result-dict keys.  What differs is the shape of
the work: every step asks for the log-posterior of ALL chains' proposals at once, so one
batched kernel launch replaces C sequential Python evaluations.

Randomness.  Chain c consumes its own numpy legacy `RandomState` in exactly the order the
reference consumes the global one (proposal normals, then one uniform per step), so a
chain seeded `s` here equals a reference run preceded by `np.random.seed(s)`.  A 1-D
`param_ini` runs a single chain on numpy's *global* RandomState -- the reference's behaviour.
"""
import numpy as np


def global_rng():
    """numpy's process-global legacy RandomState (what np.random.randn & co. draw from)."""
    return np.random.mtrand._rand


class MCMCBase(object):
    """Base class: the accept/reject loop; children supply `sampler_batch`."""

    def __init__(self):
        self.logPost = None
        self.logPostGrad = None
        self.logPostBatch = None
        self.logPostGradBatch = None
        self.postInfo = {}
        self.rngs = None

    # -- model hooks ---------------------------------------------------------------
    def setLogPost(self, logPost, logPostGrad, **postInfo):
        """Single-vector callables, as in the reference (mcmc.py:25-35):
        logPost(params(p,), **postInfo) -> float, logPostGrad(...) -> (p,) array."""
        self.logPost = logPost
        self.logPostGrad = logPostGrad
        self.postInfo = postInfo

    def setLogPostBatch(self, logPostBatch, logPostGradBatch=None, **postInfo):
        """Batched callables: logPostBatch(params(C,p), **postInfo) -> (C,) float64,
        logPostGradBatch(...) -> (C,p).  Used in preference to the single-vector ones."""
        self.logPostBatch = logPostBatch
        self.logPostGradBatch = logPostGradBatch
        self.postInfo = postInfo

    def _lp(self, X):
        if self.logPostBatch is not None:
            return np.asarray(self.logPostBatch(X, **self.postInfo), dtype=np.float64).reshape(-1)
        return np.array([self.logPost(x, **self.postInfo) for x in X], dtype=np.float64)

    def _lpg(self, X):
        if self.logPostGradBatch is not None:
            return np.asarray(self.logPostGradBatch(X, **self.postInfo), dtype=np.float64)
        assert self.logPostGrad is not None
        return np.array([self.logPostGrad(x, **self.postInfo) for x in X], dtype=np.float64)

    # -- the chain loop -------------------------------------------------------------
    def run(self, nmcmc, param_ini, rngs=None, verbose=True):
        """Run `nmcmc` MH steps.

        Args:
            nmcmc (int): number of steps.
            param_ini (np.ndarray): `(p,)` -> one chain, results shaped as the reference's;
                `(C,p)` -> C chains in lock-step, every result gains a leading C axis.
            rngs (list[np.random.RandomState], optional): one generator per chain.  Default:
                the global numpy generator for a single chain.

        Returns:
            dict: 'chain' (nmcmc+1,p), 'mapparams' (p,), 'maxpost', 'accrate',
            'logpost' (nmcmc+1,), 'alphas' (nmcmc+1,) [alphas[0] = 0].
        """
        assert self.logPost is not None or self.logPostBatch is not None
        param_ini = np.asarray(param_ini, dtype=np.float64)
        single = param_ini.ndim == 1
        cur = np.array(param_ini.reshape(1, -1) if single else param_ini, dtype=np.float64, copy=True)
        C, p = cur.shape
        if rngs is None:
            if C != 1:
                raise ValueError("multi-chain runs need one RandomState per chain (rngs=...)")
            rngs = [global_rng()]
        if len(rngs) != C:
            raise ValueError(f"{C} chains but {len(rngs)} generators")
        self.rngs = list(rngs)

        cur_U = -self._lp(cur)
        best = cur.copy()
        best_lp = -cur_U
        chain = np.empty((C, nmcmc + 1, p))
        alphas = np.zeros((C, nmcmc + 1))
        lps = np.empty((C, nmcmc + 1))
        chain[:, 0] = cur
        lps[:, 0] = -cur_U
        nacc = np.zeros(C, dtype=np.int64)
        acc_rate = np.zeros(C)

        for i in range(nmcmc):
            prop, k_cur, k_prop = self.sampler_batch(cur, i)
            prop_U = -self._lp(prop)
            with np.errstate(over="ignore", invalid="ignore"):
                mh = np.exp((cur_U + k_cur) - (prop_U + k_prop))
            u = np.array([r.random_sample() for r in self.rngs])
            take = u < mh                                   # NaN / inf behave as in `u < mh_prob`
            if take.any():
                nacc += take
                cur = np.where(take[:, None], prop + 0.0, cur)
                cur_U = np.where(take, prop_U + 0.0, cur_U)
                better = take & (-cur_U >= best_lp)
                best_lp = np.where(better, -cur_U, best_lp)
                best = np.where(better[:, None], cur, best)
            chain[:, i + 1] = cur
            alphas[:, i + 1] = mh
            lps[:, i + 1] = -cur_U
            acc_rate = nacc / float(i + 1)
            if verbose and nmcmc >= 10 and ((i + 2) % (nmcmc / 10) == 0 or i == nmcmc - 2):
                print('%d / %d completed, acceptance rate %lg' % (i + 2, nmcmc, float(np.mean(acc_rate))))

        if single:
            return {'chain': chain[0], 'mapparams': best[0], 'maxpost': float(best_lp[0]),
                    'accrate': float(acc_rate[0]), 'logpost': lps[0], 'alphas': alphas[0]}
        return {'chain': chain, 'mapparams': best, 'maxpost': best_lp, 'accrate': acc_rate,
                'logpost': lps, 'alphas': alphas}

    # -- proposals --------------------------------------------------------------------
    def sampler_batch(self, current, imcmc):
        """Proposals for all chains: (C,p) -> ((C,p) proposal, (C,) current K, (C,) proposed K)."""
        raise NotImplementedError("sampler_batch is implemented by the sampler classes")

    def sampler(self, current, imcmc):
        """Single-chain form of the reference's sampler(current, imcmc) (mcmc.py:104-115)."""
        if self.rngs is None:
            self.rngs = [global_rng()]
        prop, kc, kp = self.sampler_batch(np.asarray(current, dtype=np.float64).reshape(1, -1), imcmc)
        return prop[0], float(kc[0]), float(kp[0])
