"""Oracle: Metropolis-Hastings chain stepper and the AMCMC / HMC / MALA proposals.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  One chain at a time, float64
numpy, consuming a numpy legacy RandomState in exactly the order the reference
consumes the *global* numpy RNG; `np.random.seed(s)` followed by a reference run
sees the same stream as `RandomState(s)` passed here.

Restates (reference file:line, relative to /root/reference):
  * MCMCBase.run ...... quinn/mcmc/mcmc.py:39-101
  * AMCMC.sampler ..... quinn/mcmc/admcmc.py:38-74
  * HMC.sampler ....... quinn/mcmc/hmc.py:27-70
  * MALA.sampler ...... quinn/mcmc/mala.py:24-53
A "multi-chain" run (which the reference does not have) is DEFINED as C such
sequential runs, chain c using RandomState(seed0 + c).
"""
import numpy as np


class AmcmcState:
    """Running mean / covariance / proposal covariance of one adaptive chain."""

    def __init__(self, cov_ini=None, gamma=0.1, t0=100, tadapt=1000):
        self.cov_ini, self.gamma, self.t0, self.tadapt = cov_ini, gamma, t0, tadapt
        self.mean = None
        self.cov = None
        self.propcov = None

    def propose(self, x, step, rng, logpostgrad=None):
        p = len(x)
        if step == 0:                                           # admcmc.py:52-54
            self.mean = x.copy()
            self.cov = np.zeros((p, p))
        else:                                                   # admcmc.py:56-59
            self.mean = (step * self.mean + x) / (step + 1.0)
            keep = (step - 1.0) / step
            gain = (step + 1.0) / step ** 2
            dev = x - self.mean
            self.cov = keep * self.cov + gain * np.dot(np.reshape(dev, (p, 1)), np.reshape(dev, (1, p)))
        if step == 0:                                           # admcmc.py:61-65
            if self.cov_ini is not None:
                self.propcov = self.cov_ini
            else:
                self.propcov = 0.01 + np.diag(0.09 * np.abs(x))
        elif step > self.t0 and step % self.tadapt == 0:        # admcmc.py:66-67
            self.propcov = (self.gamma * 2.4 ** 2 / p) * (self.cov + 10 ** (-8) * np.eye(p))
        prop = x.copy()
        prop += rng.multivariate_normal(np.zeros(p,), self.propcov)   # admcmc.py:70
        return prop, 0.0, 0.0


class HmcState:
    def __init__(self, epsilon=0.05, L=3):
        self.epsilon, self.L = epsilon, L

    def propose(self, x, step, rng, logpostgrad=None):
        eps, L = self.epsilon, self.L
        q = x.copy()
        mom = rng.randn(len(x))                                 # hmc.py:43
        k_cur = np.sum(np.square(mom)) / 2
        mom += eps * logpostgrad(q) / 2                         # hmc.py:48
        for j in range(L):                                      # hmc.py:50-57
            q += eps * mom
            if j != L - 1:
                mom += eps * logpostgrad(q)
        mom += eps * logpostgrad(q) / 2                         # hmc.py:60
        mom = -mom
        k_prop = np.sum(np.square(mom)) / 2
        return q, k_cur, k_prop


class MalaState:
    def __init__(self, epsilon=0.05):
        self.epsilon = epsilon

    def propose(self, x, step, rng, logpostgrad=None):
        eps = self.epsilon
        q = x.copy()
        mom = rng.randn(len(x))                                 # mala.py:42
        g_cur = logpostgrad(x)
        q += 0.5 * eps ** 2 * g_cur + eps * mom                 # mala.py:45
        g_prop = logpostgrad(q)
        k_cur = np.sum(np.square(mom)) / 2
        mom += eps * (g_cur + g_prop) / 2
        k_prop = np.sum(np.square(mom)) / 2
        return q, k_cur, k_prop


def run_chain(logpost, proposal, nmcmc, param_ini, rng, logpostgrad=None, record_uniforms=False):
    """The MH loop of mcmc.py:55-99.  Returns the reference's result dict (same keys)
    plus 'accepted' (bool per step) and optionally 'uniforms'."""
    cur = param_ini.copy()
    cur_U = -logpost(cur)
    best, best_lp = cur, -cur_U
    chain, alphas, lps = [cur], [0.0], [-cur_U]
    accepted, uniforms = [], []
    n_acc = 0
    acc_rate = 0.0
    for i in range(nmcmc):
        prop, k_cur, k_prop = proposal.propose(cur, i, rng, logpostgrad)
        prop_U = -logpost(prop)
        with np.errstate(over="ignore", invalid="ignore"):
            mh = np.exp((cur_U + k_cur) - (prop_U + k_prop))    # mcmc.py:69-72
        u = rng.random_sample()                                 # mcmc.py:75
        take = bool(u < mh)
        if take:
            n_acc += 1
            cur = prop + 0.0
            cur_U = prop_U + 0.0
            if -cur_U >= best_lp:                               # mcmc.py:79-81
                best_lp = -cur_U
                best = cur + 0.0
        chain.append(cur)
        alphas.append(mh)
        lps.append(-cur_U)
        accepted.append(take)
        uniforms.append(u)
        acc_rate = float(n_acc) / (i + 1)
    out = {"chain": np.array(chain), "mapparams": best, "maxpost": best_lp,
           "accrate": acc_rate, "logpost": np.array(lps), "alphas": np.array(alphas),
           "accepted": np.array(accepted, dtype=bool)}
    if record_uniforms:
        out["uniforms"] = np.array(uniforms)
    return out


def run_multichain(make_logpost, make_proposal, nmcmc, pdim, seeds, make_logpostgrad=None):
    """C independent sequential chains; chain c: RandomState(seeds[c]) -> param_ini =
    rand(pdim) (nn_mcmc.py:124) -> run_chain.  Stacked outputs [C, ...]."""
    outs = []
    for s in seeds:
        rng = np.random.RandomState(s)
        ini = rng.rand(pdim)
        g = make_logpostgrad() if make_logpostgrad is not None else None
        outs.append(run_chain(make_logpost(), make_proposal(), nmcmc, ini, rng, logpostgrad=g))
    keys = ["chain", "mapparams", "maxpost", "accrate", "logpost", "alphas", "accepted"]
    return {k: np.stack([np.asarray(o[k]) for o in outs]) for k in keys}
