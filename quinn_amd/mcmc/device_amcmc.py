"""Adaptive Metropolis with EVERYTHING resident on the GPU (throughput engine).

The reference-exact host sampler (`AMCMC`) keeps a p x p covariance per chain and draws through an
SVD; at the headline configuration (64 chains, p = 8513) that is 74 GB of host state and minutes of
LAPACK per adaptation -- the reference itself cannot run there.  This engine keeps states, proposal
factors and the chain history in HBM and never synchronises with the host inside a window:

  * one MH step = `qn_mcmc_propose` (in-kernel Philox normals) -> batched log-posterior kernel ->
    `qn_mcmc_accept` (accept test, state / MAP / history / window update); the step counter lives in
    device memory, so the step is a static launch sequence that can also be captured in ONE HIP graph
    (`use_graph=True`); measured at cfg2: 5.9 k steps/s with direct launches (4 launches, ~3 us of
    host time each), 5.2 k steps/s replayed as a graph (replay floor ~10-16 us), so direct launch is
    the default;
  * the reference's covariance recursion (admcmc.py:52-59) is, in closed form, the unbiased sample
    covariance of x_0..x_i (tests/test_amcmc_math.py); it is only USED every `tadapt` steps
    (admcmc.py:66-67), so it is accumulated per window as one batched SYRK (Gram matrix of the
    window, shifted by x_0 against cancellation) instead of a rank-1 update of 580 MB per chain
    and step;
  * initial proposal covariance 0.01 + diag(0.09|x0|) (admcmc.py:65) = diagonal + rank one:
    drawn exactly as sqrt(0.09|x0|) * z + 0.1 * z0 without forming a p x p matrix;
  * adapted proposals: batched Cholesky factor L of (gamma 2.4^2/p)(cov + 1e-8 I), draw = L z
    (one batched GEMV per step; HBM-bound: p^2 * 8 B per chain and step).

Same target distribution and the same adaptation schedule as the reference; the random streams
differ (Philox instead of numpy MT19937, Cholesky instead of SVD factor), so chains agree with the
host sampler in distribution, not bit for bit.  Use `AMCMC` for bit-exact parity.
"""
import ctypes

import numpy as np
import torch

from .. import _lib
from ..ops import BatchedMLP


class DeviceAMCMC:
    def __init__(self, op: BatchedMLP, sigma, gamma=0.1, t0=100, tadapt=1000, cov_ini=None, seed=0,
                 factor_dtype=torch.float64, chol_chunk=8, use_graph=False):
        if op.dtype != "float64":
            raise NotImplementedError("the device AMCMC engine runs the float64 operator")
        self.op, self.sigma = op, float(sigma)
        self.gamma, self.t0, self.tadapt = gamma, int(t0), int(tadapt)
        self.cov_ini = cov_ini
        self.dev = op.device
        self.seed = int(seed) & (2 ** 63 - 1)
        self.factor_dtype = factor_dtype
        self.chol_chunk = chol_chunk
        self.use_graph = use_graph
        self._L = _lib.lib()

    # -- kernel wrappers (enqueue on the current stream) ---------------------------------------------
    def _propose(self, cur, sd, c1, step_ptr, out):
        st = ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        C, p = out.shape
        _lib.check(self._L.qn_mcmc_propose(cur.data_ptr() if cur is not None else None,
                                           sd.data_ptr() if sd is not None else None, c1, C, p, self.seed,
                                           step_ptr.data_ptr(), out.data_ptr(), st), "qn_mcmc_propose")

    def _accept(self, s, prop, sse, nmcmc):
        st = ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        C, p = prop.shape
        _lib.check(self._L.qn_mcmc_accept(
            prop.data_ptr(), sse.data_ptr(), self.sigma, self.op.N, C, p, nmcmc, self.seed, s['cur'].data_ptr(),
            s['cur_lp'].data_ptr(), s['best'].data_ptr(), s['best_lp'].data_ptr(),
            s['chain'].data_ptr() if s['chain'] is not None else None, s['lps'].data_ptr(), s['alphas'].data_ptr(),
            s['nacc'].data_ptr(), s['x0'].data_ptr(), s['win'].data_ptr(), self.tadapt, s['step'].data_ptr(), st),
            "qn_mcmc_accept")

    def run(self, nmcmc, param_ini, store_chain=True, verbose=False):
        dev, f64 = self.dev, torch.float64
        cur = torch.as_tensor(np.asarray(param_ini), dtype=f64, device=dev).clone().reshape(-1, self.op.p)
        C, p = cur.shape
        n = self.op.N
        const = (n / 2) * np.log(2 * np.pi) + n * np.log(self.sigma)
        cur_lp = -(0.5 * self.op.sse(cur) / self.sigma ** 2 + const)
        s = {'cur': cur, 'cur_lp': cur_lp, 'best': cur.clone(), 'best_lp': cur_lp.clone(), 'x0': cur.clone(),
             'chain': torch.empty(C, nmcmc + 1, p, dtype=f64, device=dev) if store_chain else None,
             'lps': torch.empty(C, nmcmc + 1, dtype=f64, device=dev),
             'alphas': torch.zeros(C, nmcmc + 1, dtype=f64, device=dev),
             'nacc': torch.zeros(C, dtype=torch.int64, device=dev),
             'win': torch.zeros(C, self.tadapt, p, dtype=f64, device=dev),     # slot 0 = x_0 - x_0 = 0
             'step': torch.zeros(2, dtype=torch.int64, device=dev)}
        if store_chain:
            s['chain'][:, 0] = cur
        s['lps'][:, 0] = cur_lp
        std0 = torch.sqrt(0.09 * s['x0'].abs())
        prop = torch.empty(C, p, dtype=f64, device=dev)
        z = torch.empty(C, p, dtype=f64, device=dev)
        state = {'L': None}
        if self.cov_ini is not None:
            state['L'] = torch.linalg.cholesky(torch.as_tensor(np.asarray(self.cov_ini), dtype=f64, device=dev))[None].to(self.factor_dtype)
        S2, s1 = None, torch.zeros(C, p, dtype=f64, device=dev)
        nabs = 1                                                           # sample 0 (zero after the shift)

        def one_step():
            L = state['L']
            if L is None:
                self._propose(s['cur'], std0, 0.1, s['step'], prop)
            else:
                self._propose(None, None, 0.0, s['step'], z)
                if L.shape[0] == 1:
                    prop.copy_(s['cur'] + (z.to(L.dtype) @ L[0].T).to(f64))
                else:
                    prop.copy_(s['cur'] + torch.bmm(L, z.to(L.dtype)[:, :, None])[:, :, 0].to(f64))
            sse = self.op.sse(prop)
            self._accept(s, prop, sse, nmcmc)

        graph = None

        def make_graph():
            if not self.use_graph:
                return None
            # warm-up on a side stream (allocator / library handles), then capture one step.
            # The warm-up steps are real steps: rewind the counter and every piece of state afterwards.
            # (chain row / window slot written by the warm-up step are rewritten by the real step)
            snap = {k: v.clone() for k, v in s.items() if k not in ('chain', 'win') and isinstance(v, torch.Tensor)}
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                one_step()
            torch.cuda.current_stream(dev).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                one_step()
            torch.cuda.synchronize(dev)
            for k, v in snap.items():                                      # capture itself does not execute
                if isinstance(v, torch.Tensor):
                    s[k].copy_(v)
            return g

        i = 0
        while i < nmcmc:
            if i > 0 and i % self.tadapt == 0:
                # the window holds samples i-tadapt+1 .. i (slot i % tadapt = 0 is sample i)
                Y = s['win']
                G = torch.bmm(Y.transpose(1, 2), Y)
                S2 = G if S2 is None else S2.add_(G)
                del G
                s1 += Y.sum(dim=1)
                nabs += self.tadapt
                if i > self.t0:
                    scale = self.gamma * 2.4 ** 2 / p
                    if state['L'] is None or state['L'].shape[0] != C:
                        state['L'] = torch.empty(C, p, p, dtype=self.factor_dtype, device=dev)
                        graph = None                                       # the step changes shape: recapture
                    for c0 in range(0, C, self.chol_chunk):
                        sl = slice(c0, c0 + self.chol_chunk)
                        cov = S2[sl] - s1[sl, :, None] * s1[sl, None, :] / nabs
                        cov.mul_(scale / (nabs - 1))
                        cov.diagonal(dim1=1, dim2=2).add_(scale * 1e-8)
                        state['L'][sl] = torch.linalg.cholesky(cov).to(self.factor_dtype)
                        del cov
            if self.use_graph and graph is None:
                graph = make_graph()
            nrun = min(nmcmc, (i // self.tadapt + 1) * self.tadapt) - i     # up to the next window boundary
            for _ in range(nrun):
                if graph is not None:
                    graph.replay()
                else:
                    one_step()
            i += nrun
            if verbose:
                print('%d / %d completed, acceptance rate %lg' % (i, nmcmc, float(s['nacc'].double().mean()) / i))
        return {'chain': s['chain'], 'mapparams': s['best'], 'maxpost': s['best_lp'],
                'accrate': s['nacc'].double() / max(nmcmc, 1), 'logpost': s['lps'], 'alphas': s['alphas']}
