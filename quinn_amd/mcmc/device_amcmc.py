"""Adaptive Metropolis with EVERYTHING resident on the GPU (throughput engine).

The reference-exact host sampler (`AMCMC`) keeps a p x p covariance per chain and draws through an
SVD; at the headline configuration (64 chains, p = 8513) that is 74 GB of host state and minutes of
LAPACK per adaptation -- the reference itself cannot run there.  This engine keeps states, proposal
factors and the chain history in HBM and never synchronises with the host inside the loop:

  * log-posterior of all chains' proposals: the batched HIP kernel (`BatchedMLP.sse`);
  * the reference's covariance recursion (admcmc.py:52-59) is, in closed form, the unbiased sample
    covariance of x_0..x_i (checked in tests/test_amcmc_math.py); it is only USED every `tadapt`
    steps (admcmc.py:66-67), so it is accumulated per adaptation window as one batched SYRK
    (Gram matrix of the window, shifted by x_0 against cancellation) instead of a rank-1 update
    of 580 MB per chain and step;
  * initial proposal covariance 0.01 + diag(0.09|x0|) (admcmc.py:65) = diagonal + rank one:
    drawn exactly as sqrt(0.09|x0|) * z + 0.1 * z0 without forming a p x p matrix;
  * adapted proposals: batched Cholesky factor L of (gamma 2.4^2/p)(cov + 1e-8 I), draw = L z
    (one batched GEMV per step; HBM-bound: p^2 * 8 B per chain and step).

Same target distribution and the same adaptation schedule as the reference; the random streams
differ (device Philox instead of numpy MT19937, Cholesky instead of SVD factor), so chains agree
with the host sampler in distribution, not bit for bit.  Use `AMCMC` for bit-exact parity.
"""
import numpy as np
import torch

from ..ops import BatchedMLP


class DeviceAMCMC:
    def __init__(self, op: BatchedMLP, sigma, gamma=0.1, t0=100, tadapt=1000, cov_ini=None, seed=0,
                 factor_dtype=torch.float64, chol_chunk=8):
        self.op, self.sigma = op, float(sigma)
        self.gamma, self.t0, self.tadapt = gamma, t0, tadapt
        self.cov_ini = cov_ini
        self.dev = op.device
        self.gen = torch.Generator(device=self.dev)
        self.gen.manual_seed(int(seed))
        self.factor_dtype = factor_dtype
        self.chol_chunk = chol_chunk
        n = op.N
        self._const = (n / 2) * np.log(2 * np.pi) + n * np.log(self.sigma)

    def logpost(self, W):
        """[C] float64 device tensor: -(0.5 sse/sigma^2 + n/2 log 2pi + n log sigma)."""
        Wc = W if self.op.tdt == torch.float64 else W.to(self.op.tdt)
        return -(0.5 * self.op.sse(Wc) / self.sigma ** 2 + self._const)

    def run(self, nmcmc, param_ini, store_chain=True, verbose=False):
        dev, f64 = self.dev, torch.float64
        cur = torch.as_tensor(np.asarray(param_ini), dtype=f64, device=dev).clone().reshape(-1, self.op.p)
        C, p = cur.shape
        cur_lp = self.logpost(cur)
        best, best_lp = cur.clone(), cur_lp.clone()
        chain = torch.empty(C, nmcmc + 1, p, dtype=f64, device=dev) if store_chain else None
        lps = torch.empty(C, nmcmc + 1, dtype=f64, device=dev)
        alphas = torch.zeros(C, nmcmc + 1, dtype=f64, device=dev)
        if store_chain:
            chain[:, 0] = cur
        lps[:, 0] = cur_lp
        nacc = torch.zeros(C, dtype=torch.int64, device=dev)
        # proposal state
        x0 = cur.clone()
        std0 = torch.sqrt(0.09 * x0.abs())                       # diag part of the initial covariance
        L = None                                                  # [C,p,p] (or [1,p,p]) Cholesky factor once set
        if self.cov_ini is not None:
            L = torch.linalg.cholesky(torch.as_tensor(np.asarray(self.cov_ini), dtype=f64, device=dev))[None].to(self.factor_dtype)
        S2 = None                                                 # sum (x - x0)(x - x0)^T over absorbed samples
        s1 = torch.zeros(C, p, dtype=f64, device=dev)
        nabs = 0                                                  # samples absorbed into S2 / s1
        win = torch.empty(C, self.tadapt, p, dtype=f64, device=dev)
        nwin = 0

        def absorb():
            nonlocal S2, s1, nabs, nwin
            if nwin == 0:
                return
            Y = win[:, :nwin]
            G = torch.bmm(Y.transpose(1, 2), Y)
            S2 = G if S2 is None else S2.add_(G)
            s1 += Y.sum(dim=1)
            nabs += nwin
            nwin = 0

        for i in range(nmcmc):
            # sample i of the chain enters the statistics the sampler sees at step i (admcmc.py:52-59)
            win[:, nwin] = cur - x0
            nwin += 1
            if nwin == self.tadapt:
                absorb()
            if i > self.t0 and i % self.tadapt == 0:
                absorb()
                n1 = nabs                                         # = i + 1 samples x_0..x_i
                scale = self.gamma * 2.4 ** 2 / p
                if L is None or L.shape[0] != C:
                    L = torch.empty(C, p, p, dtype=self.factor_dtype, device=dev)
                for c0 in range(0, C, self.chol_chunk):           # chunked: bounds the p x p temporaries
                    sl = slice(c0, c0 + self.chol_chunk)
                    cov = S2[sl] - s1[sl, :, None] * s1[sl, None, :] / n1
                    cov.mul_(scale / (n1 - 1))
                    cov.diagonal(dim1=1, dim2=2).add_(scale * 1e-8)
                    L[sl] = torch.linalg.cholesky(cov).to(self.factor_dtype)
                    del cov
            z = torch.randn(C, p, dtype=f64, device=dev, generator=self.gen)
            if L is None:
                z0 = torch.randn(C, 1, dtype=f64, device=dev, generator=self.gen)
                prop = cur + std0 * z + 0.1 * z0
            elif L.shape[0] == 1:
                prop = cur + (z.to(L.dtype) @ L[0].T).to(f64)
            else:
                prop = cur + torch.bmm(L, z.to(L.dtype)[:, :, None])[:, :, 0].to(f64)
            prop_lp = self.logpost(prop)
            mh = torch.exp(prop_lp - cur_lp)                      # exp(current_U - proposed_U), mcmc.py:72
            u = torch.rand(C, dtype=f64, device=dev, generator=self.gen)
            take = u < mh
            nacc += take
            cur = torch.where(take[:, None], prop, cur)
            cur_lp = torch.where(take, prop_lp, cur_lp)
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
