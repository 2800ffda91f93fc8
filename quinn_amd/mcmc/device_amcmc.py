"""Adaptive Metropolis with EVERYTHING resident on the GPU (throughput engine).

The reference-exact host sampler (`AMCMC`) keeps a p x p covariance per chain and draws through an
SVD; at the headline configuration (64 chains, p = 8513) that is 74 GB of host state and minutes of
LAPACK per adaptation -- the reference itself cannot run there.  This engine keeps states, the
proposal's ingredients and the chain history in HBM and never synchronises with the host:

  * one MH step = proposal kernel (in-kernel Philox normals) -> batched log-posterior kernel ->
    `qn_mcmc_accept` (accept test, state / MAP / history update); the step counter lives in device
    memory, so a run of steps is a static launch sequence; `use_graph=True` captures a block of steps at a time in
    ONE HIP graph and replays it (measured at cfg2: no faster than direct launches -- the host keeps the
    queue full either way -- so direct launching is the default);
  * initial proposal covariance 0.01 + diag(0.09|x0|) (admcmc.py:65) = diagonal + rank one:
    drawn exactly as sqrt(0.09|x0|) * z + 0.1 * z0 without forming a p x p matrix (`qn_mcmc_propose`);
  * adapted proposals are drawn in SAMPLE SPACE (`qn_mcmc_propose_hist_block`; one step at a time:
    `qn_mcmc_propose_hist`).  The reference's covariance
    recursion (admcmc.py:52-59) is, in closed form, the unbiased sample covariance of x_0..x_i
    (tests/test_amcmc_math.py), and an adaptation at step i sets the proposal covariance to
    c (cov_i + 1e-8 I), c = gamma 2.4^2 / p (admcmc.py:66-67).  A chain of n = i + 1 samples has only
    K = 1 + (accepted moves) DISTINCT states x_k with multiplicities w_k, and
        delta = sqrt(c/(n-1)) sum_k sqrt(w_k) u_k (x_k - mean) + sqrt(c 1e-8) v,   u, v iid N(0,1)
    has exactly that covariance.  So the engine stores the distinct states (float32, shifted by x_0:
    `qn_mcmc_accept` appends a row per accepted move and counts multiplicities) and a proposal is a
    K x p GEMV over them: at cfg2 K ~ 10^2..10^3 rows of 34 KB per chain and step, where a p x p factor
    is 290-580 MB per chain and step plus an O(p^3) factorisation per adaptation.  An adaptation is a
    snapshot of (K, sqrt(w), mean): a few elementwise torch ops, no SYRK, no Cholesky;
  * delta does not depend on the chain's state, only on the frozen snapshot and the step's random
    numbers, so the increments of the next TB = 64 steps are formed in ONE pass over the history
    (`qn_mcmc_propose_hist_block`: a (TB x K).(K x p) product per chain on the float32 matrix cores, HBM
    traffic per step / TB) and
    a step's proposal is `cur + delta[t]`.

Same target distribution and the same adaptation schedule as the reference; the random streams
differ (Philox instead of numpy MT19937, sample-space draw instead of an SVD factor), so chains agree
with the host sampler in distribution, not bit for bit.  Use `AMCMC` for bit-exact parity.
"""
import ctypes
import os

import numpy as np
import torch

from .. import _lib
from ..ops import BatchedMLP


class DeviceAMCMC:
    def __init__(self, op: BatchedMLP, sigma, gamma=0.1, t0=100, tadapt=1000, cov_ini=None, seed=0,
                 use_graph=False, max_history_bytes=64 << 30, chain0=0, fuse_propose=True, groups=None, max_rows=4096,
                 overlap_hist=True, hist_scale0=256.0, pause_gc=True):
        if op.dtype != "float64":
            raise NotImplementedError("the device AMCMC engine runs the float64 operator")
        self.op, self.sigma = op, float(sigma)
        self.gamma, self.t0, self.tadapt = gamma, int(t0), int(tadapt)
        self.cov_ini = cov_ini
        self.dev = op.device
        self.seed = int(seed) & (2 ** 63 - 1)
        self.use_graph = use_graph
        self.max_history_bytes = int(max_history_bytes)
        # bound on the stored rows per chain (p float32 each).  When a chain's history could overflow before the next
        # adaptation it is COMPRESSED in sample space (`_compress_history`: same multiplicity total and mean, scatter
        # kept up to rank max_rows/8 -- exactly, once max_rows/8 >= p): cost and memory of the adapted proposal stay
        # bounded however long the chain runs
        self.max_rows = int(max_rows)
        # next step's proposal written by the accept kernel (bit-identical to the separate proposal kernel; A/B on one
        # box: 2-3 % slower while the accept kernel ran one workgroup per chain, 1 % faster now that it spreads a
        # chain over several)
        self.fuse_propose = bool(fuse_propose)
        self.chain0 = int(chain0)      # global id of this engine's first chain (random streams are keyed by it)
        # groups > 1: the chains are split into that many independent groups, each on its own HIP stream, enqueued
        # block by block from this one host thread: a group's accept / propose kernel (a chain of memory round trips that
        # leaves the GPU idle: 8.7 of a step's 84 us at cfg2) overlaps the other group's forward kernel.  The groups' launches
        # split a chain's rows as the launch of all chains would (qn_mlp_desc_set_plan_batch): the chains do not depend on
        # the number of groups, bit for bit.  None: 2 groups from 32 chains on (measured at cfg2: 11.9 -> 12.4 k steps/s
        # before the first adaptation; 4 groups are bound by the enqueuing thread) when the fused kernels run the network, else 1
        self.groups = None if groups is None else max(1, int(groups))
        # the increments of the NEXT block of TB steps are formed on a second stream while the current block's steps run
        # (they depend on the frozen snapshot and the step numbers only): the history product streams the chains'
        # histories from HBM, the log-posterior kernel is bound by vector issue, and the accept kernels leave most of the
        # GPU idle.  Same random numbers, same results (tests/test_gpu_device_amcmc.py)
        # history rows are float16((x - ref) * S): S starts at `hist_scale0` (a power of two; rows saturate at |x - ref| =
        # 65504 / S) with ref = the start, and is re-chosen per chain -- together with ref -- at every compression of its history
        self.hist_scale0 = float(2.0 ** round(np.log2(hist_scale0)))
        # run() pauses Python's cyclic garbage collector while it enqueues steps (gc.disable / gc.enable are process-wide and not
        # meant to be toggled from several threads at once): pause_gc=False leaves the collector alone, at the price of an
        # occasional 30 ms hole in the GPU's queue
        self.pause_gc = bool(pause_gc)
        self.overlap_hist = bool(overlap_hist) and not os.environ.get("QUINN_AMD_NO_HIST_OVERLAP")
        self._side = None
        self._subs = None
        self._L = _lib.lib()

    # -- kernel wrappers (enqueue on the current stream) ---------------------------------------------
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    @staticmethod
    def _step_ptr(s):
        """Pointer to the CURRENT slot of the double-buffered device step counter (include/quinn_amd.h)."""
        return s['step'].data_ptr() + 8 * s['par']

    def _propose(self, cur, sd, c1, s, out):
        C, p = out.shape
        _lib.check(self._L.qn_mcmc_propose(cur.data_ptr() if cur is not None else None,
                                           sd.data_ptr() if sd is not None else None, c1, C, self.chain0, p, self.seed,
                                           self._step_ptr(s), out.data_ptr(), self._stream()), "qn_mcmc_propose")

    def _propose_hist_block(self, s, snap, coef, delta, step_abs=None):
        """Increments of TB steps from the device step counter on (step_abs None), or from the absolute step step_abs."""
        C, _, p = delta.shape
        _lib.check(self._L.qn_mcmc_propose_hist_block(
            s['hist'].data_ptr(), snap['w'].data_ptr(), snap['k'].data_ptr(), snap['mean'].data_ptr(), s['hscale'].data_ptr(),
            snap['s_lr'],
            snap['s_iso'], C, self.chain0, p, s['hist'].shape[2], s['hist'].shape[1], self.seed,
            0 if step_abs is None else int(step_abs), self._step_ptr(s) if step_abs is None else None,
            coef.data_ptr(),
            delta.data_ptr(), None if os.environ.get("QUINN_AMD_NO_ORDER") else snap['order'].data_ptr(),     # (env: A/B)
            self._stream()), "qn_mcmc_propose_hist_block")

    def _apply_delta(self, s, snap, delta, t, out):
        C, p = out.shape
        _lib.check(self._L.qn_mcmc_apply_delta(s['cur'].data_ptr(), delta.data_ptr(), int(t), snap['s_iso'], C,
                                               self.chain0, p, self.seed, self._step_ptr(s), out.data_ptr(), self._stream()),
                   "qn_mcmc_apply_delta")

    def _accept(self, s, prop, sse, nmcmc, nxt=None):
        """sse: [C] or [C, parts] (partial sums, added left to right by the kernel)."""
        C, p = prop.shape
        nparts = sse.shape[1] if sse.dim() == 2 else 1
        if nxt is not None:
            mode, sd, c1, delta, t, s_iso = nxt
            _lib.check(self._L.qn_mcmc_accept_propose(
                prop.data_ptr(), sse.data_ptr(), self.sigma, self.op.N, C, self.chain0, p, nmcmc, self.seed,
                s['cur'].data_ptr(), s['cur_lp'].data_ptr(), s['best'].data_ptr(), s['best_lp'].data_ptr(),
                s['chain'].data_ptr() if s['chain'] is not None else None, s['lps'].data_ptr(), s['alphas'].data_ptr(),
                s['nacc'].data_ptr(), s['x0'].data_ptr(), s['hist'].data_ptr(), s['hscale'].data_ptr(), s['mult'].data_ptr(),
                s['kcur'].data_ptr(), s['sumx'].data_ptr(), s['hist'].shape[1], s['hist'].shape[2], s['step'].data_ptr(),
                mode, sd.data_ptr() if sd is not None else None, c1, delta.data_ptr() if delta is not None else None,
                int(t), s_iso, prop.data_ptr(), s['par'], nparts, self._stream()), "qn_mcmc_accept_propose")
            s['par'] ^= 1
            return
        _lib.check(self._L.qn_mcmc_accept(
            prop.data_ptr(), sse.data_ptr(), self.sigma, self.op.N, C, self.chain0, p, nmcmc, self.seed, s['cur'].data_ptr(),
            s['cur_lp'].data_ptr(), s['best'].data_ptr(), s['best_lp'].data_ptr(),
            s['chain'].data_ptr() if s['chain'] is not None else None, s['lps'].data_ptr(), s['alphas'].data_ptr(),
            s['nacc'].data_ptr(), s['x0'].data_ptr(), s['hist'].data_ptr(), s['hscale'].data_ptr(), s['mult'].data_ptr(),
            s['kcur'].data_ptr(), s['sumx'].data_ptr(), s['hist'].shape[1], s['hist'].shape[2],
            s['step'].data_ptr(), s['par'], nparts, self._stream()), "qn_mcmc_accept")
        s['par'] ^= 1              # the kernel read slot `par` of the per-chain scalars / step counter and wrote the other

    def _ngroups(self, C):
        if self.groups is None:
            return 2 if C >= 32 and self.cov_ini is None and self.op.path(C) == _lib.PATH_FUSED else 1
        return min(self.groups, C)

    def run(self, nmcmc, param_ini, store_chain=True, verbose=False):
        """(The cyclic garbage collector is paused while the steps are enqueued: the enqueuing thread makes ~25 000 launches a
        second and nothing it allocates per step is cyclic, but a full collection in the middle of a run is a 30 ms hole in
        the GPU's queue -- one 1000-step window at 6.3 k steps/s among windows at 7.7 k.  Restored on the way out.)"""
        import gc
        was = self.pause_gc and gc.isenabled()
        if was:
            gc.disable()
        try:
            return self._run(nmcmc, param_ini, store_chain, verbose)
        finally:
            if was:
                gc.enable()

    def _run(self, nmcmc, param_ini, store_chain=True, verbose=False):
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
        # ---- several groups of chains side by side
        bounds = [C * g // G for g in range(G + 1)]
        pstride = (p + 3) // 4 * 4
        if C * self._kcap(nmcmc) * pstride * 2 > self.max_history_bytes:
            raise MemoryError(f"state history {C} x {self._kcap(nmcmc)} x {pstride} float16 exceeds max_history_bytes="
                              f"{self.max_history_bytes}: lower max_rows or raise the limit")
        if self._subs is None or [e.chain0 - self.chain0 for e in self._subs[0]] != bounds[:-1]:
            op = self.op
            engs = [DeviceAMCMC(BatchedMLP(op.arch, op.X, op.Y, device=op.device, dtype=op.dtype), self.sigma, self.gamma,
                                self.t0, self.tadapt, self.cov_ini, self.seed, self.use_graph, self.max_history_bytes,
                                self.chain0 + bounds[g], self.fuse_propose, max_rows=self.max_rows,
                                overlap_hist=self.overlap_hist, hist_scale0=self.hist_scale0, pause_gc=False)
                    for g in range(G)]
            for e in engs:
                # a group's launches split every chain's rows as the launch of all C chains does: each chain's SSE is summed in
                # the same order, so the chains are those of groups = 1 bit for bit (include/quinn_amd.h)
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
        self.last_state, self.last_states = None, [e.last_state for e in engs]      # (diagnostics: per group)
        return out

    @staticmethod
    def _sum_of_samples(s):
        """sum over ALL samples so far of (x_i - x0), [C, p] float64: the accept kernel keeps `sumx` = the stays of the states a
        chain has LEFT (mult x (x - x0), added when it leaves: no per-step pass over the vector); the current state's stay is
        added here (a handful of elementwise ops, once per adaptation)."""
        kc = s['kcur'][s['par']].long().clamp_(max=s['mult'].shape[1] - 1)
        wc = s['mult'].gather(1, kc[:, None]).to(torch.float64)
        return s['sumx'] + wc * (s['cur'] - s['x0'])

    def _kcap(self, nmcmc):
        """Rows of the history buffer: every accepted move of the run if that fits max_rows, else max_rows (which must
        hold a compressed history, max_rows/4 + 2 rows, plus one window of new rows between two checks)."""
        if nmcmc + 1 <= self.max_rows:
            return nmcmc + 1
        need = max(64, (4 * (self.tadapt + 4) + 2) // 3)                    # (>= 64: below that max(8, kcap // 8) breaks the room count)
        if self.max_rows < need:
            raise ValueError(f"max_rows = {self.max_rows} is too small for tadapt = {self.tadapt}: need >= {need}")
        return self.max_rows

    def _compress_history(self, s, room, p, step=0, spread=True):
        """Compress the stored history of every chain that has fewer than `room` free rows (one device->host read per call;
        called once per adaptation window) into 2 r + 2 rows, r = kcap / 8, that carry the SAME weighted mean and -- up to rank
        r -- the same weighted scatter about it.  With the rows h_i (multiplicities w_i, n_c = sum w_i, mean m_c) of
        everything but the current state, B = diag(sqrt w)(H - m_c), Y = B Omega (Omega: p x r Gaussian) and an orthonormal
        basis Q of range(Y),
            R = Q^T B       (r x p),      R^T R ~ B^T B
        (exactly when rank(B) <= r; the directions a single randomised range pass finds otherwise).  Q by Cholesky-QR applied
        twice (Gram matrices and factors in float64, two batched triangular solves), or, when Y^T Y is numerically singular
        (rank(B) < r: short chains, p < r), Q = Y V Lambda^-1/2 from its eigen-decomposition with the null directions
        dropped -- the same projector, hence the same scatter R^T R.  The pseudo-states
            m_c + R_j / sqrt 2,  m_c - R_j / sqrt 2   (multiplicity 1 each, j < r),      m_c   (multiplicity n_c - 2 r)
        have total multiplicity n_c, mean m_c and scatter R^T R: for the accept / proposal kernels they are ordinary
        history rows (integer multiplicities adding up to the number of samples; the parallel-axis term of a later,
        different overall mean comes out by itself).  The current state keeps a row of its own behind them.  The chains of
        a group (<= 16: bounds the padded copy) go through ONE set of batched GEMMs / factorisations, no per-chain host work.
        spread: besides the chains that MUST be compressed now, the fullest of those that would come due within the next
        windows are taken as well, up to C / 6 chains per call, so that the chains of a long run do not all come due in the
        same window."""
        kcap = s['hist'].shape[1]
        r = max(8, kcap // 8)
        par = s['par']
        kc_h = s['kcur'][par].cpu().numpy().astype(np.int64)                    # (the one device->host read)
        ok = kc_h > 2 * r + 1
        must = np.nonzero((kc_h + 1 + room > kcap) & ok)[0]
        sel = list(must)
        if spread:
            # chains that would come due within the next windows (one more window even if every step were accepted) are taken early while the call has room in
            # its budget of C / 6 chains: evens out the windows in which many chains fill up together
            # The chains of a run fill up in step (similar acceptance rates), so without help their FIRST compressions all come
            # due in the same window (45 ms for 64 chains: that window ran at 5.7 k steps/s against 7.5-8 k around it).  The
            # early-eligibility threshold is therefore staggered by chain (global id mod 8, a quarter window apart): the first
            # cycle is spread over the windows before, and the later cycles inherit the spread.
            # Only a chain's FIRST compression is staggered (s['ncomp'] counts them): afterwards a chain restarts from
            # 2 r + 1 rows, and an offset of up to 1.75 windows would keep it "due" from the moment it was compressed.
            high = kcap - 2 * room
            ncomp = s.setdefault('ncomp', np.zeros(len(kc_h), dtype=np.int64))
            stag = np.where(ncomp == 0, ((self.chain0 + np.arange(len(kc_h))) % 8) * (room // 4), 0)
            budget = max(len(sel), -(-len(kc_h) // 6))                          # (a chain comes due every ~6 windows at acceptance 0.3)
            extra = [c for c in np.argsort(-(kc_h + stag)) if ok[c] and kc_h[c] + stag[c] > high and c not in set(sel)]
            sel += extra[:max(0, budget - len(sel))]
        dev = s['hist'].device
        for g0 in range(0, len(sel), 16):
            grp = [int(c) for c in sel[g0:g0 + 16]]
            idx = torch.as_tensor(grp, device=dev)
            ks = torch.as_tensor(kc_h[grp], device=dev)                        # current row of chain c; rows 0..k-1 are compressed
            # rows taken: the group's longest history rounded up to a multiple of kcap / 4 (a few fixed shapes: the BLAS / solver
            # kernels prepare() has loaded; rows >= a chain's k are masked)
            kmax = min(kcap - 1, -(-int(kc_h[grp].max()) // max(1, kcap // 4)) * max(1, kcap // 4))
            live = torch.arange(kmax, device=dev)[None, :] < ks[:, None]        # [n, kmax]
            mult = s['mult'][idx, :kmax] * live
            ncs = mult.sum(dim=1)                                               # [n] int64
            w = mult.to(torch.float32)
            sw = w.sqrt()                                                       # [n, kmax] (rows beyond a chain's k: weight 0)
            S_old = s['hscale'][idx]                                            # [n] float64
            inv_s = (1.0 / S_old).to(torch.float32)
            # mean of the compressed states about the chain's reference point: exactly, from the float64 sum of the stays the
            # chain has left (rows 0 .. k-1 ARE those states)
            mc = (s['sumx'][idx] / ncs[:, None].double()).to(torch.float32)     # [n, p]
            gen = torch.Generator(device=dev)
            gen.manual_seed((self.seed * 1000003 + (self.chain0 + grp[0]) * 7919 + int(step)) & (2 ** 62 - 1))   # sketch keyed by (seed, chain, step)
            Om = torch.randn(p, r, dtype=torch.float32, device=dev, generator=gen).to(torch.float16)
            # Y = B Omega with B = diag(sqrt w)(H / S - 1 m^T): the big product runs on the STORED float16 rows as they lie in the
            # history buffer (no [n, kmax, p] float32 copy of the histories: round 3 made six passes over one), float16 operands,
            # float32 result; the mean term is a rank-one correction.  (Round 3: two float32 GEMMs of 36 GFLOP per chain, 10 of
            # the 16 ms a window's compressions took.)
            n = len(grp)
            Yh = torch.zeros(n, kmax, r, dtype=torch.float32, device=dev)
            # (row counts rounded up to a multiple of 128 -- a handful of GEMM shapes for the BLAS library to pick kernels for; the
            # rows in between are free rows of the buffer: zeroed here, weight 0)
            kr = {c: min(kmax, -(-int(kc_h[c]) // 128) * 128) for c in grp}
            for q, c in enumerate(grp):
                kq = int(kc_h[c])
                s['hist'][c, kq:kr[c]] = 0.0
                Yh[q, :kr[c]] = torch.mm(s['hist'][c, :kr[c], :p], Om, out_dtype=torch.float32)
            mom = mc @ Om.float()                                               # [n, r]
            Y = sw[:, :, None] * (Yh * inv_s[:, None, None] - mom[:, None, :])   # [n, kmax, r]
            del Yh
            # Q^T = (L2 L1)^-1 Y^T: Cholesky-QR applied twice (the second pass restores the orthonormality the first loses
            # when Y is ill-conditioned: with Q orthonormal to float32 accuracy R^T R can never exceed B^T B)
            Yt = Y.transpose(1, 2)
            L1, info = torch.linalg.cholesky_ex(Yt.double() @ Y.double())
            ok_chol = bool((info == 0).all())
            if ok_chol:
                Qt = torch.linalg.solve_triangular(L1.to(torch.float32), Yt, upper=False)       # [n, r, kmax]
                L2, info = torch.linalg.cholesky_ex(Qt.double() @ Qt.double().transpose(1, 2))
                ok_chol = bool((info == 0).all())
            if ok_chol:
                Qt = torch.linalg.solve_triangular(L2.to(torch.float32), Qt, upper=False)
            else:
                # (the float32 Gram matrix: its rounding noise separates the zero eigenvalues of an exactly rank-deficient Y^T Y,
                # on which the divide-and-conquer solver does not converge)
                lam, V = torch.linalg.eigh((Yt @ Y).double())                                   # [n, r], [n, r, r]
                keep = lam > lam[:, -1:] * 1e-10                                                # (rank(B) < r: drop the null directions)
                scale = torch.where(keep, lam.clamp_min(1e-300).rsqrt(), torch.zeros_like(lam))
                Qt = (V * scale[:, None, :]).transpose(1, 2).float() @ Yt                       # L^-1/2 V^T Y^T
            # R = Q^T B = (Q^T diag(sqrt w)) H / S - (Q^T sqrt w) m^T: again on the stored rows; the left factor is split into
            # float16 high and low parts (two products, float32 results): its rounding would otherwise be 2^-11 per entry
            A = Qt * sw[:, None, :]                                                             # [n, r, kmax]
            A_hi = A.to(torch.float16)
            A_lo = (A - A_hi.float()).to(torch.float16)
            R = torch.empty(n, r, p, dtype=torch.float32, device=dev)
            for q, c in enumerate(grp):
                Hq = s['hist'][c, :kr[c], :p]
                R[q] = torch.mm(A_hi[q, :, :kr[c]], Hq, out_dtype=torch.float32) + torch.mm(A_lo[q, :, :kr[c]], Hq, out_dtype=torch.float32)
            R *= inv_s[:, None, None]
            R -= A.sum(dim=2)[:, :, None] * mc[:, None, :]
            del A, A_hi, A_lo, Qt, Y
            R *= 0.5 ** 0.5
            cur_mult = s['mult'][idx, ks]                                                       # (a copy)
            # The chain's reference point moves to the weighted mean of what was compressed, ref' = ref + m_c: the pseudo-states are
            # +-R_j and 0 about it, the rows of the states to come are (x - ref') -- as small as the posterior spread once the
            # chain is stationary, which is what keeps float16's 2^-11 relative rounding harmless.  The scale S' puts the largest
            # pseudo-state entry at ~2^10 (float16 holds up to 65504: room for 64 x that before a row saturates).  The sum of the
            # stays the chain has left is re-based with it: sum w (x - ref') = sum w (x - ref) - n_c m_c.
            rmax = R.abs().amax(dim=2).amax(dim=1).double().clamp_min(1e-300)      # (two stages: the fused (1, 2) reduction took 0.65 ms)
            S_new = torch.exp2(torch.floor(torch.log2(1024.0 / rmax))).clamp_(2.0 ** -40, 2.0 ** 40)
            mc64 = mc.double()
            s['x0'][idx] += mc64
            s['sumx'][idx] -= ncs[:, None].double() * mc64
            Rs = (R * S_new.to(torch.float32)[:, None, None]).to(torch.float16)
            s['hist'][idx, 0:2 * r:2, :p] = Rs
            s['hist'][idx, 1:2 * r:2, :p] = -Rs
            s['hist'][idx, 2 * r, :p] = 0.0
            s['hist'][idx, 2 * r + 1, :p] = ((s['cur'][idx] - s['x0'][idx]) * S_new[:, None]).clamp_(-65504.0, 65504.0).to(torch.float16)
            s['hscale'][idx] = S_new
            s['mult'][idx] = 0
            s['mult'][idx, :2 * r] = 1
            s['mult'][idx, 2 * r] = (ncs - 2 * r).to(s['mult'].dtype)
            s['mult'][idx, 2 * r + 1] = cur_mult
            s['kcur'][par, idx] = 2 * r + 1
        if 'ncomp' in s or sel:
            s.setdefault('ncomp', np.zeros(len(kc_h), dtype=np.int64))[[int(c) for c in sel]] += 1
        return len(sel)

    def prepare(self, nmcmc, nchains):
        """Set-up of the adapted phase ahead of a run of `nmcmc` steps of `nchains` chains: the two work buffers of the
        history product (kept by the engine and reused by later runs of the same shape) and the first use of every library
        path the adaptation / compression steps take (lazily initialised solvers and kernels).  `run` calls it itself;
        calling it beforehand keeps that one-time cost (~120 ms at the first adaptation, ~450 ms at the first compression:
        profiles/r02_amcmc_device_50k_steps_bounded_history.json, windows at 4.7 k and 1.7 k steps/s) out of the run.
        Returns (coef, delta)."""
        dev, C, p = self.dev, int(nchains), self.op.p
        kcap, TB = self._kcap(nmcmc), int(self._L.qn_mcmc_hist_block_steps())
        key = (C, kcap, TB, p)
        if getattr(self, '_bufs', None) is not None and self._bufs[0] == key:
            return self._bufs[1], self._bufs[2]
        coef = torch.empty(int(self._L.qn_mcmc_hist_block_coef_bytes(C, kcap)), dtype=torch.uint8, device=dev)
        # (two increment buffers: the next block's is written while the current one is read)
        delta = torch.empty(2 if self.overlap_hist else 1, C, TB, p, dtype=torch.float64, device=dev)
        self._bufs = (key, coef, delta)
        k = torch.zeros(C, dtype=torch.int32, device=dev)
        torch.argsort(k, descending=True).to(torch.int32)
        k.cpu()
        if kcap < nmcmc + 1:
            # the compression's GEMMs, factorisations and solves once at their real sizes (the BLAS / solver libraries pick
            # and load their kernels per shape on first use)
            # the compression itself once per group size it will meet (a straggler alone / a full group of the spread) on a
            # dummy history: the BLAS / solver libraries pick and load kernels per shape on first use, and the caching
            # allocator gets the blocks of the temporaries
            r = max(8, kcap // 8)
            gen = torch.Generator(device=dev)
            gen.manual_seed(1)
            pstride = (p + 3) // 4 * 4
            for n in sorted({1, min(16, max(1, -(-C // 6)))}):
                fake = {'hist': torch.randn(n, kcap, pstride, dtype=torch.float32, device=dev, generator=gen).to(torch.float16),
                        'mult': torch.ones(n, kcap, dtype=torch.int32, device=dev), 'par': 0,
                        'hscale': torch.ones(n, dtype=torch.float64, device=dev), 'x0': torch.zeros(n, p, dtype=torch.float64, device=dev),
                        'sumx': torch.zeros(n, p, dtype=torch.float64, device=dev), 'cur': torch.zeros(n, p, dtype=torch.float64, device=dev)}
                for q4 in range(1, 5):                                          # every row-count bucket a compression can meet
                    kf = min(kcap - 2, q4 * max(1, kcap // 4) - 1)
                    if kf <= 2 * r + 1:
                        continue
                    fake['kcur'] = torch.full((2, n), kf, dtype=torch.int32, device=dev)
                    fake['mult'].fill_(1)
                    self._compress_history(fake, kcap, p, spread=False)
                del fake
            lam, V = torch.linalg.eigh(torch.eye(r, dtype=torch.float64, device=dev)[None] * torch.arange(1, r + 1, device=dev)[None, :, None])
            del lam, V
        idx = torch.zeros(2, dtype=torch.int64, device=dev)
        h = torch.zeros(2, 8, 8, device=dev)
        h[idx, 0:4:2, :4] = h[idx, 0:2, :4]
        torch.cuda.synchronize(dev)
        return coef, delta

    def _run_gen(self, nmcmc, param_ini, store_chain=True, verbose=False, chain_out=None):
        """The run as a generator: yields after every block of at most TB enqueued steps (nothing is awaited)."""
        dev, f64 = self.dev, torch.float64
        cur = torch.as_tensor(param_ini, dtype=f64, device=dev).clone().reshape(-1, self.op.p)
        C, p = cur.shape
        n = self.op.N
        const = (n / 2) * np.log(2 * np.pi) + n * np.log(self.sigma)
        cur_lp = -(0.5 * self.op.sse(cur) / self.sigma ** 2 + const)
        # history of distinct states: one row per accepted move at most -> nmcmc + 1 rows always suffice; capped at
        # max_rows (thinned when it could fill up before the next adaptation)
        kcap, pstride = self._kcap(nmcmc), (p + 3) // 4 * 4
        # (a bounded history also needs the compression's working set: per group of <= 16 chains the sketch / factor arrays
        # [n, kcap, r] float32 x 4 and the compressed rows [n, r, p] float32, and prepare()'s dummy history of one group)
        n_grp = min(16, max(1, -(-C // 6)))
        r_cmp = max(8, kcap // 8)
        extra = (n_grp * (4 * kcap * r_cmp * 4 + r_cmp * p * 4) + n_grp * kcap * pstride * 2) if kcap < nmcmc + 1 else 0
        if C * kcap * pstride * 2 + extra > self.max_history_bytes:
            raise MemoryError(f"state history {C} x {kcap} x {pstride} float16 (+ {extra >> 20} MiB of compression work space) exceeds max_history_bytes="
                              f"{self.max_history_bytes}: lower max_rows or raise the limit")
        # per-chain scalars the accept kernel maintains are double-buffered by step parity ([2, C]; slot `par` is current)
        s = {'cur': cur, 'cur_lp': torch.stack([cur_lp, cur_lp]), 'best': cur.clone(),
             'best_lp': torch.stack([cur_lp, cur_lp]), 'x0': cur.clone(), 'par': 0,
             'chain': (chain_out if chain_out is not None else torch.empty(C, nmcmc + 1, p, dtype=f64, device=dev))
                      if store_chain else None,
             'lps': torch.empty(C, nmcmc + 1, dtype=f64, device=dev),
             'alphas': torch.zeros(C, nmcmc + 1, dtype=f64, device=dev),
             'nacc': torch.zeros(C, dtype=torch.int64, device=dev),
             'hist': torch.empty(C, kcap, pstride, dtype=torch.float16, device=dev),
             'hscale': torch.full((C,), self.hist_scale0, dtype=f64, device=dev),
             'mult': torch.zeros(C, kcap, dtype=torch.int32, device=dev),
             'kcur': torch.zeros(2, C, dtype=torch.int32, device=dev),
             'sumx': torch.zeros(C, p, dtype=f64, device=dev),
             'step': torch.zeros(2, dtype=torch.int64, device=dev)}
        s['hist'][:, 0] = 0.0                                               # row 0 = x_0 - x_0
        s['mult'][:, 0] = 1
        if store_chain:
            s['chain'][:, 0] = cur
        s['lps'][:, 0] = cur_lp
        std0 = torch.sqrt(0.09 * s['x0'].abs())
        prop = torch.empty(C, p, dtype=f64, device=dev)
        state = {'snap': None, 'L': None, 'have_prop': False, 'dbuf': 0, 'ahead': None}
        if self.cov_ini is not None:
            state['L'] = torch.linalg.cholesky(torch.as_tensor(np.asarray(self.cov_ini), dtype=f64, device=dev))
            z = torch.empty(C, p, dtype=f64, device=dev)

        TB = int(self._L.qn_mcmc_hist_block_steps())
        G = TB                                                              # steps per captured graph
        coef = delta = None
        if nmcmc > max(self.t0, self.tadapt):                               # the run will adapt: buffers now, not at the first adaptation
            coef, delta = self.prepare(nmcmc, C)

        fuse = self.fuse_propose

        def run_initial(n, last_fused):
            """n steps with the initial proposal; with fusion every accept but the last also writes the next
            step's proposal (last_fused: the step after these n is an initial-proposal step too)."""
            if state['L'] is not None:                                      # user-supplied initial covariance
                for _ in range(n):
                    self._propose(None, None, 0.0, s, z)
                    prop.copy_(s['cur'] + z @ state['L'].T)
                    self._accept(s, prop, self.op.sse_parts(prop), nmcmc)
                return False
            have = state['have_prop']
            for k in range(n):
                if not have:
                    self._propose(s['cur'], std0, 0.1, s, prop)
                nxt = (1, std0, 0.1, None, 0, 0.0) if fuse and (k + 1 < n or last_fused) else None
                self._accept(s, prop, self.op.sse_parts(prop), nmcmc, nxt)
                have = nxt is not None
            return have

        def block_adapted(nsteps, step_abs=None, more=False):
            # increments of TB consecutive steps in ONE pass over the history (they do not depend on the chain's
            # state), then nsteps <= TB steps; the block starts at the device step counter.  With fusion the accept
            # kernel of step t writes the proposal of step t + 1 (inside the block).
            # step_abs (= the device step counter, known to the host): the block's increments may have been formed
            # ahead on the side stream; more: another block of this window follows -- form ITS increments meanwhile
            snap = state['snap']
            overlap = self.overlap_hist and step_abs is not None
            buf = state['dbuf'] if overlap else 0
            dl = delta[buf]
            if overlap and state['ahead'] is not None:
                torch.cuda.current_stream(dev).wait_event(state['ahead'])
                state['ahead'] = None
            else:
                self._propose_hist_block(s, snap, coef, dl, step_abs)
            if overlap and more:
                if self._side is None:
                    self._side = torch.cuda.Stream(device=dev)
                    # (two events for the whole run, re-recorded: a fresh pair per block was ~1600 Python objects per 50 000
                    # steps, and a collector pause of the enqueuing thread shows up as a 30 ms hole in one window)
                    self._ev_fork, self._ev_join = torch.cuda.Event(), torch.cuda.Event()
                self._ev_fork.record(torch.cuda.current_stream(dev))         # (the other buffer's last readers are enqueued before this)
                self._side.wait_event(self._ev_fork)
                with torch.cuda.stream(self._side):
                    self._propose_hist_block(s, snap, coef, delta[1 - buf], step_abs + TB)
                    self._ev_join.record(self._side)
                    state['ahead'] = self._ev_join
                state['dbuf'] = 1 - buf
            have = False
            for t in range(nsteps):
                if not have:
                    self._apply_delta(s, snap, dl, t, prop)
                nxt = (2, None, 0.0, dl, t + 1, snap['s_iso']) if fuse and t + 1 < nsteps else None
                self._accept(s, prop, self.op.sse_parts(prop), nmcmc, nxt)
                have = nxt is not None

        def capture(fn):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            return g

        graphs = {}
        i = 0
        while i < nmcmc:
            if kcap < nmcmc + 1 and i > 0:
                # room for every step up to the next check (each may append a row); compress the chains that lack it
                if os.environ.get("QN_AMCMC_TIMING"):
                    import time as _time
                    torch.cuda.synchronize(dev); _tc = _time.perf_counter()
                ncomp = self._compress_history(s, min(nmcmc, (i // self.tadapt + 1) * self.tadapt) - i + 1, p, step=i)
                if os.environ.get("QN_AMCMC_TIMING"):
                    torch.cuda.synchronize(dev); print("[timing] step %d: compression of %d chains %.1f ms" % (i, ncomp, 1e3 * (_time.perf_counter() - _tc)), flush=True, file=__import__("sys").stderr)
            _tm = os.environ.get("QN_AMCMC_TIMING")                          # (diagnostic: synchronising timings of the window's phases)
            if _tm:
                import time as _time
                torch.cuda.synchronize(dev); _t0 = _time.perf_counter()
            if i > self.t0 and i % self.tadapt == 0:
                # adaptation (admcmc.py:66-67) = snapshot of the history x_0..x_i: n = i + 1 samples
                scale = self.gamma * 2.4 ** 2 / p
                state['snap'] = {'k': (s['kcur'][s['par']] + 1).clone(), 'w': s['mult'].to(torch.float32).sqrt_(),
                                 'mean': self._sum_of_samples(s) / (i + 1), 's_lr': float(np.sqrt(scale / i)),
                                 # dispatch order of the history product: longest history first
                                 'order': torch.argsort(s['kcur'][s['par']], descending=True).to(torch.int32),
                                 's_iso': float(np.sqrt(scale * 1e-8))}
                graphs = {k: g for k, g in graphs.items() if k[0] != 'adapted'}   # new snapshot tensors: recapture
                if coef is None:
                    coef, delta = self.prepare(nmcmc, C)
            if _tm:
                torch.cuda.synchronize(dev); print("[timing] step %d: snapshot / allocation %.1f ms" % (i, 1e3 * (_time.perf_counter() - _t0)), flush=True, file=__import__("sys").stderr)
            nrun = min(nmcmc, (i // self.tadapt + 1) * self.tadapt) - i     # up to the next adaptation
            adapted = state['snap'] is not None
            nfull, rest = divmod(nrun, G)
            if self.use_graph and nfull > 0 and (adapted or state['L'] is None):   # (torch matmul path: not captured)
                # G steps captured once and replayed (no faster than direct launches at cfg2; kept as an option)
                # a captured block bakes in the parity slots of the double-buffered scalars it starts from (G is even,
                # so a replay leaves the parity where it was): one graph per (regime, starting parity) -- a stretch
                # of an odd number of directly launched steps in between flips the parity
                key = ('adapted' if adapted else 'initial', s['par'])
                if graphs.get(key) is None:
                    graphs[key] = capture((lambda: block_adapted(G)) if adapted
                                          else (lambda: run_initial(G, False)))
                state['have_prop'] = False
                for _ in range(nfull):
                    graphs[key].replay()
                    yield
            else:
                rest = nrun
            if adapted:
                while rest > 0:
                    if _tm and rest == nrun:
                        torch.cuda.synchronize(dev); _t1 = _time.perf_counter()
                    block_adapted(min(rest, TB), None if self.use_graph else i + nrun - rest, rest > TB)
                    if _tm and rest == nrun:
                        torch.cuda.synchronize(dev); print("[timing] step %d: first adapted block of %d steps %.1f ms" % (i, min(rest, TB), 1e3 * (_time.perf_counter() - _t1)), flush=True, file=__import__("sys").stderr)
                    rest -= min(rest, TB)
                    yield
            else:
                # the run of initial-proposal steps ends at an adaptation (new regime) or at the end of the chain
                while rest > 0:
                    nb = min(rest, TB)
                    state['have_prop'] = run_initial(nb, rest > nb)
                    rest -= nb
                    yield
            i += nrun
            if verbose:
                print('%d / %d completed, acceptance rate %lg' % (i, nmcmc, float(s['nacc'].double().mean()) / i))
        s['sumx'] = self._sum_of_samples(s)     # (the run is over: the current state's stay enters the sum)
        self.last_state = s            # (tests / diagnostics: history rows, multiplicities, sum of all samples)
        return {'chain': s['chain'], 'mapparams': s['best'], 'maxpost': s['best_lp'][s['par']].clone(),
                'accrate': s['nacc'].double() / max(nmcmc, 1), 'logpost': s['lps'], 'alphas': s['alphas']}
