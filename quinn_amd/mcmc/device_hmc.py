"""Hamiltonian Monte Carlo with states, momenta and history resident on the GPU (throughput engine).

Same leapfrog scheme and accept rule as the reference (quinn/mcmc/hmc.py:43-66, mcmc.py:69-75):
momentum ~ N(0, I), half kick, L drifts with L-1 inner kicks, half kick; L+1 gradient evaluations per
proposal, each ONE batched fused forward+backward launch for all chains.  The log-posterior of the
proposal is taken from the SSE of the last gradient evaluation (the reference spends an extra forward
pass on it, mcmc.py:68).  No host synchronisation inside the loop; momenta and uniforms come from
the device generator, so chains agree with the host `HMC` in distribution, not bit for bit.
"""
import numpy as np
import torch

from ..ops import BatchedMLP


class DeviceHMC:
    def __init__(self, op: BatchedMLP, sigma, epsilon=0.05, L=3, seed=0):
        self.op, self.sigma, self.epsilon, self.L = op, float(sigma), float(epsilon), int(L)
        self.dev = op.device
        self.gen = torch.Generator(device=self.dev)
        self.gen.manual_seed(int(seed))
        n = op.N
        self._const = (n / 2) * np.log(2 * np.pi) + n * np.log(self.sigma)

    def _lp_grad(self, q):
        """(log-posterior [C], its gradient [C,p]) in float64."""
        qc = q if self.op.tdt == torch.float64 else q.to(self.op.tdt)
        sse, g = self.op.sse_grad(qc)
        return -(0.5 * sse / self.sigma ** 2 + self._const), g.double().mul_(-0.5 / self.sigma ** 2)

    def run(self, nmcmc, param_ini, store_chain=True, verbose=False):
        dev, f64 = self.dev, torch.float64
        cur = torch.as_tensor(np.asarray(param_ini), dtype=f64, device=dev).clone().reshape(-1, self.op.p)
        C, p = cur.shape
        eps, L = self.epsilon, self.L
        cur_lp, cur_g = self._lp_grad(cur)
        best, best_lp = cur.clone(), cur_lp.clone()
        chain = torch.empty(C, nmcmc + 1, p, dtype=f64, device=dev) if store_chain else None
        lps = torch.empty(C, nmcmc + 1, dtype=f64, device=dev)
        alphas = torch.zeros(C, nmcmc + 1, dtype=f64, device=dev)
        if store_chain:
            chain[:, 0] = cur
        lps[:, 0] = cur_lp
        nacc = torch.zeros(C, dtype=torch.int64, device=dev)
        for i in range(nmcmc):
            mom = torch.randn(C, p, dtype=f64, device=dev, generator=self.gen)
            k_cur = mom.square().sum(dim=1) / 2
            q = cur.clone()
            mom.add_(cur_g, alpha=eps / 2)                 # gradient at the current state is cached
            for j in range(L):
                q.add_(mom, alpha=eps)
                lp_q, g_q = self._lp_grad(q)
                mom.add_(g_q, alpha=eps if j != L - 1 else eps / 2)
            k_prop = mom.square().sum(dim=1) / 2
            mh = torch.exp((-cur_lp + k_cur) - (-lp_q + k_prop))
            u = torch.rand(C, dtype=f64, device=dev, generator=self.gen)
            take = u < mh
            nacc += take
            cur = torch.where(take[:, None], q, cur)
            cur_g = torch.where(take[:, None], g_q, cur_g)
            cur_lp = torch.where(take, lp_q, cur_lp)
            better = take & (cur_lp >= best_lp)
            best_lp = torch.where(better, cur_lp, best_lp)
            best = torch.where(better[:, None], cur, best)
            if store_chain:
                chain[:, i + 1] = cur
            alphas[:, i + 1] = mh
            lps[:, i + 1] = cur_lp
            if verbose and nmcmc >= 10 and (i + 2) % (nmcmc // 10) == 0:
                print('%d / %d completed, acceptance rate %lg' % (i + 2, nmcmc, float(nacc.double().mean()) / (i + 1)))
        return {'chain': chain, 'mapparams': best, 'maxpost': best_lp, 'accrate': nacc.double() / max(nmcmc, 1),
                'logpost': lps, 'alphas': alphas}
