"""Hamiltonian Monte Carlo with states, momenta and history resident on the GPU (throughput engine).

Same leapfrog scheme and accept rule as the reference (quinn/mcmc/hmc.py:43-66, mcmc.py:65-85): momentum ~ N(0, I),
half kick, L drifts with L-1 inner kicks, half kick.  One step is a static sequence of launches through the C ABI and
nothing else -- no torch op, no host synchronisation:

    qn_hmc_begin   momenta (in-kernel Philox keyed by the GLOBAL chain id), K_cur partials, half kick with the cached
                   gradient of the current state, first drift
    L x [ qn_mlp_sse_fwdbwd at q  ->  qn_hmc_leap (kick + drift; last: half kick + K_prop partials) ]
    qn_hmc_accept  log-posterior of the proposal from the SSE of the LAST gradient call, MH test, state / gradient /
                   MAP / history rows, device step counter (double-buffered by step parity, as in the AMCMC engine)

so L gradient launches per step instead of the reference's L + 1 gradients + 1 forward (the gradient at the current
state is the accepted proposal's, the proposal's log-posterior comes with its gradient).  With `use_graph=True` pairs
of steps (parity 0, 1) are captured once in a HIP graph and replayed.  Random streams are keyed by (seed, step, global
chain id): a chain's path does not depend on how the chains are split over ranks or launches (only the summation order
of its SSE does, through the row split the gradient kernel picks for the batch size).  Chains agree with the host `HMC`
in distribution, not bit for bit (Philox instead of numpy's MT19937).
"""
import ctypes

import numpy as np
import torch

from .. import _lib
from ..ops import BatchedMLP


class DeviceHMC:
    def __init__(self, op: BatchedMLP, sigma, epsilon=0.05, L=3, seed=0, chain0=0, use_graph=False, groups=None):
        self.op, self.sigma, self.epsilon, self.L = op, float(sigma), float(epsilon), int(L)
        if self.L < 1:
            raise ValueError("HMC needs L >= 1 leapfrog steps")
        # groups > 1: the chains run as that many independent groups on their own HIP streams, enqueued step by step from this
        # one host thread; a group's gradient launch splits a chain's rows as the launch of all chains would
        # (qn_mlp_desc_set_plan_batch), so the chains do not depend on the number of groups, bit for bit.  Default 1: unlike the
        # AMCMC engine's forward (two workgroups per CU, a lone one runs 1.8 x faster) the gradient kernel puts ONE workgroup on a
        # CU, so a group's launch cannot spread into the CUs the other group's small kernels leave idle -- measured at cfg2:
        # 1418 -> 1427 steps/s (HMC, L = 3), 4094 -> 4100 (MALA)
        self.groups = None if groups is None else max(1, int(groups))
        self._subs = None
        self.dev = op.device
        self.seed = int(seed) & (2 ** 63 - 1)
        self.chain0 = int(chain0)          # global id of this engine's first chain (random streams are keyed by it)
        self.use_graph = bool(use_graph)
        self._L = _lib.lib()
        n = op.N
        self._const = (n / 2) * np.log(2 * np.pi) + n * np.log(self.sigma)

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def _step(self, s, nmcmc):
        """Enqueue one HMC step on the current stream (reads slot s['par'] of the scalars, writes the other)."""
        Lb, st = self._L, self._stream()
        C, p = s['cur'].shape
        step_ptr = s['step'].data_ptr() + 8 * s['par']
        _lib.check(Lb.qn_hmc_begin(s['cur'].data_ptr(), s['gcur'].data_ptr(), self.sigma, self.epsilon, C, self.chain0, p,
                                   self.seed, step_ptr, s['mom'].data_ptr(), s['q'].data_ptr(), s['kcur'].data_ptr(), st),
                   "qn_hmc_begin")
        for j in range(self.L):
            last = j == self.L - 1
            qc = s['q'] if s['qc'] is None else s['qc'].copy_(s['q'])          # float32 operator: cast the positions
            self.op.sse_grad(qc, out=(s['sse'], s['gq']))
            _lib.check(Lb.qn_hmc_leap(s['gq'].data_ptr(), self.op.qdt, self.sigma, self.epsilon, int(last), C, p,
                                      s['mom'].data_ptr(), s['q'].data_ptr(), s['kprop'].data_ptr(), st), "qn_hmc_leap")
        g64 = s['gq'] if s['g64'] is None else s['g64'].copy_(s['gq'])
        _lib.check(Lb.qn_hmc_accept(
            s['q'].data_ptr(), g64.data_ptr(), s['sse'].data_ptr(), s['kcur'].data_ptr(), s['kprop'].data_ptr(),
            self.sigma, self.op.N, C, self.chain0, p, nmcmc, self.seed, s['cur'].data_ptr(), s['gcur'].data_ptr(),
            s['cur_lp'].data_ptr(), s['best'].data_ptr(), s['best_lp'].data_ptr(),
            s['chain'].data_ptr() if s['chain'] is not None else None, s['lps'].data_ptr(), s['alphas'].data_ptr(),
            s['nacc'].data_ptr(), s['step'].data_ptr(), s['par'], st), "qn_hmc_accept")
        s['par'] ^= 1

    def _ngroups(self, C):
        if self.groups is None:
            return 1
        return min(self.groups, C)

    def run(self, nmcmc, param_ini, store_chain=True, verbose=False):
        ini = torch.as_tensor(np.asarray(param_ini), dtype=torch.float64, device=self.dev).reshape(-1, self.op.p)
        C, p = ini.shape
        G = self._ngroups(C)
        if G <= 1:
            gen = self._run_gen(nmcmc, ini, store_chain, verbose)
            while True:
                try:
                    next(gen)
                except StopIteration as e:
                    return e.value
        bounds = [C * g // G for g in range(G + 1)]
        if self._subs is None or [e.chain0 - self.chain0 for e in self._subs[0]] != bounds[:-1]:
            op = self.op
            engs = [DeviceHMC(BatchedMLP(op.arch, op.X, op.Y, device=op.device, dtype=op.dtype), self.sigma, self.epsilon,
                              self.L, self.seed, self.chain0 + bounds[g], self.use_graph, groups=1) for g in range(G)]
            for e in engs:
                e.op.set_plan_batch(max(C, op.set_plan_batch(-1)))
                path = op.set_path(_lib.PATH_AUTO)                            # (the kernel family forced on the parent, if any)
                op.set_path(path)
                e.op.set_path(path)
            self._subs = (engs, [torch.cuda.Stream(device=self.dev) for _ in range(G)])
        engs, streams = self._subs
        chain = torch.empty(C, nmcmc + 1, p, dtype=torch.float64, device=self.dev) if store_chain else None
        gens = [engs[g]._run_gen(nmcmc, ini[bounds[g]:bounds[g + 1]], store_chain, verbose and g == 0,
                                 chain_out=None if chain is None else chain[bounds[g]:bounds[g + 1]]) for g in range(G)]
        main = torch.cuda.current_stream(self.dev)
        for st in streams:
            st.wait_stream(main)
        res, live = [None] * G, list(range(G))
        while live:
            for g in list(live):
                with torch.cuda.stream(streams[g]):
                    try:
                        next(gens[g])
                    except StopIteration as e:
                        res[g] = e.value
                        live.remove(g)
        for st in streams:
            main.wait_stream(st)
        out = {k: torch.cat([r[k] for r in res]) for k in ('mapparams', 'maxpost', 'accrate', 'logpost', 'alphas')}
        out['chain'] = chain
        return out

    def _run_gen(self, nmcmc, param_ini, store_chain=True, verbose=False, chain_out=None):
        """The run as a generator: yields after every enqueued step (pair of steps under a graph); nothing is awaited."""
        dev, f64 = self.dev, torch.float64
        cur = torch.as_tensor(param_ini, dtype=f64, device=dev).clone().reshape(-1, self.op.p)
        C, p = cur.shape
        nk = int(self._L.qn_hmc_parts(p))
        f32op = self.op.tdt != f64
        sse0, g0 = self.op.sse_grad(cur if not f32op else cur.to(self.op.tdt))
        cur_lp = -(0.5 * sse0 / self.sigma ** 2 + self._const)
        s = {'cur': cur, 'gcur': g0.double() if f32op else g0.clone(), 'cur_lp': torch.stack([cur_lp, cur_lp]),
             'best': cur.clone(), 'best_lp': torch.stack([cur_lp, cur_lp]), 'par': 0,
             'mom': torch.empty(C, p, dtype=f64, device=dev), 'q': torch.empty(C, p, dtype=f64, device=dev),
             'gq': torch.empty(C, p, dtype=self.op.tdt, device=dev), 'sse': torch.empty(C, dtype=f64, device=dev),
             'qc': torch.empty(C, p, dtype=self.op.tdt, device=dev) if f32op else None,
             'g64': torch.empty(C, p, dtype=f64, device=dev) if f32op else None,
             'kcur': torch.empty(C, nk, dtype=f64, device=dev), 'kprop': torch.empty(C, nk, dtype=f64, device=dev),
             'chain': (chain_out if chain_out is not None else torch.empty(C, nmcmc + 1, p, dtype=f64, device=dev))
                      if store_chain else None,
             'lps': torch.empty(C, nmcmc + 1, dtype=f64, device=dev),
             'alphas': torch.zeros(C, nmcmc + 1, dtype=f64, device=dev),
             'nacc': torch.zeros(C, dtype=torch.int64, device=dev),
             'step': torch.zeros(2, dtype=torch.int64, device=dev)}
        if store_chain:
            s['chain'][:, 0] = cur
        s['lps'][:, 0] = cur_lp
        i = 0
        if self.use_graph and nmcmc >= 4:
            # two steps (parity 0 and 1) captured once; warm the kernels up on a side stream first (torch's capture rule)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                self._step(s, nmcmc)
                self._step(s, nmcmc)
            torch.cuda.current_stream(dev).wait_stream(side)
            i = 2
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._step(s, nmcmc)
                self._step(s, nmcmc)
            i += 2                                         # (capture does not run the kernels: replay once for steps 2, 3)
            graph.replay()
            yield
            while i + 2 <= nmcmc:
                graph.replay()
                i += 2
                yield
                if verbose and nmcmc >= 10 and i % max(2, (nmcmc // 10) // 2 * 2) == 0:
                    print('%d / %d completed, acceptance rate %lg' % (i, nmcmc, float(s['nacc'].double().mean()) / i))
        while i < nmcmc:
            self._step(s, nmcmc)
            i += 1
            yield
            if verbose and nmcmc >= 10 and (i + 1) % (nmcmc // 10) == 0:
                print('%d / %d completed, acceptance rate %lg' % (i + 1, nmcmc, float(s['nacc'].double().mean()) / i))
        return {'chain': s['chain'], 'mapparams': s['best'], 'maxpost': s['best_lp'][s['par']].clone(),
                'accrate': s['nacc'].double() / max(nmcmc, 1), 'logpost': s['lps'], 'alphas': s['alphas']}
