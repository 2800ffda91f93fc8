"""MCMC wrapper around a user MLP, all chains' log-posteriors evaluated in one launch.

Mirror of the reference's `NN_MCMC` (quinn/solvers/nn_mcmc.py:15-200): same constructor,
`fit` / `logpost` / `logpostgrad` / `predict_*` signatures and attributes (`samples`,
`cmode`, `pdim`, `lpinfo`).  Build-only keyword extras (defaults = reference behaviour):
`nchains`, `seeds` on `fit`; `device`, `dtype` on the constructor.

logpost(w) = -[ 0.5*SSE(w)/sigma^2 + (N/2) log 2pi + N log sigma ]   (no prior,
nn_mcmc.py:64 -> losses.py:197-200); SSE and dSSE/dw come from the HIP kernels, the scalar
tail is applied here in float64.
"""
import copy
import sys

import numpy as np
from scipy.optimize import minimize

from ..mcmc.admcmc import AMCMC
from ..mcmc.hmc import HMC
from ..mcmc.mala import MALA
from ..ops import BatchedMLP, neg_log_post_from_sse
from ..parallel import dist_info, empty_results, gather_results, run_chains_sharded, shard_bounds
from .quinn import QUiNNBase


class NN_MCMC(QUiNNBase):
    """Attributes: samples `(nmcmc+1, p)` (or `(C, nmcmc+1, p)` for C chains), cmode (MAP
    weights), pdim, lpinfo, verbose."""

    def __init__(self, nnmodel, verbose=True, device=None, dtype="float64", kernels="auto"):
        """kernels (build-only extra; the reference has no counterpart): 'auto' -- the fastest kernel family for the network's
        shape; for 64 / 128 / 256-wide tanh networks in float64 those are the sliced int8-product kernels, whose operands are
        rounded to 2^-47 of their row / activation scale (a norm-wise 47-bit bound: ~1e-14 .. 1e-13 on log-posteriors and
        gradients of ordinary networks, where float64 rounding gives ~1e-16).  'float64' -- plain float64 arithmetic
        throughout (the float64-MFMA fused kernels where they apply, the layer-wise float64 kernels otherwise): what the
        reference's torch CPU path computes, up to summation order."""
        super().__init__(nnmodel, device=device, dtype=dtype)
        if kernels not in ("auto", "float64"):
            raise ValueError("kernels must be 'auto' or 'float64'")
        self.kernels = kernels
        self.verbose = verbose
        self.pdim = sum(p.numel() for p in self.nnmodel.parameters())
        print("Number of parameters:", self.pdim)
        if self.verbose:
            self.print_params(names_only=True)
        self.samples = None
        self.cmode = None
        self.lpinfo = {}
        self._op = None
        self._op_key = None

    # -- device operator bound to lpinfo's dataset -----------------------------------------
    def _operator(self, lpinfo):
        # the cached operator belongs to these very objects (kept referenced here, compared with `is`: an id() alone
        # can be recycled by a new array after the old one is freed)
        key = (lpinfo['xd'], lpinfo['yd'])
        if self._op is None or self._op_key[0] is not key[0] or self._op_key[1] is not key[1]:
            xd = np.asarray(lpinfo['xd'], dtype=np.float64)
            yd = np.asarray(lpinfo['yd'], dtype=np.float64)      # list of (o,) rows -> (N,o)
            self._op = BatchedMLP(self.arch, xd, yd.reshape(xd.shape[0], -1), device=self._device,
                                  dtype=self._dtype)
            if self.kernels == "float64":
                self._op.use_exact_float64()
            self._op_key = key
        return self._op

    @staticmethod
    def _check_ltype(lpinfo):
        if lpinfo['ltype'] != 'classical':
            print('Likelihood type is not recognized. Exiting.')
            sys.exit()

    def logpost_batch(self, W, lpinfo=None):
        """Log-posterior of every row of W `(C,p)`: numpy float64 `(C,)`."""
        lpinfo = self.lpinfo if lpinfo is None else lpinfo
        self._check_ltype(lpinfo)
        op = self._operator(lpinfo)
        sse = op.sse(np.atleast_2d(W)).cpu().numpy()
        return -neg_log_post_from_sse(sse, len(lpinfo['yd']), lpinfo['lparams']['sigma'])

    def logpostgrad_batch(self, W, lpinfo=None):
        """Gradient of the log-posterior for every row of W: numpy float64 `(C,p)`."""
        lpinfo = self.lpinfo if lpinfo is None else lpinfo
        self._check_ltype(lpinfo)
        op = self._operator(lpinfo)
        _, g = op.sse_grad(np.atleast_2d(W))
        sig = np.float64(lpinfo['lparams']['sigma'])
        return -(0.5 * g.double().cpu().numpy() / sig ** 2)

    def logpost(self, modelpars, lpinfo):
        """float: log-posterior of one flat weight vector (reference signature)."""
        return float(self.logpost_batch(np.asarray(modelpars).reshape(1, -1), lpinfo)[0])

    def logpostgrad(self, modelpars, lpinfo):
        """np.ndarray `(p,)`: gradient of the log-posterior (reference signature)."""
        return self.logpostgrad_batch(np.asarray(modelpars).reshape(1, -1), lpinfo)[0]

    # -- fit -------------------------------------------------------------------------------
    def fit(self, xtrn, ytrn, zflag=True, datanoise=0.05, nmcmc=6000, param_ini=None, sampler='amcmc',
            sampler_params=None, *, nchains=1, seeds=None, engine='host', gather='all', gather_chain=None, bfgs_jac=None):
        """Run MCMC over the flat weight vector.

        Args (reference): xtrn `(N,d)`, ytrn `(N,o)`, zflag (BFGS pre-fit of a random start),
            datanoise (likelihood sigma), nmcmc, param_ini `(p,)` [or `(C,p)`], sampler
            ('amcmc' | 'hmc' | 'mala'), sampler_params (dict splatted into the sampler).
        Build-only: nchains (C independent chains in lock-step), seeds (C ints; chain c then
            equals a reference run preceded by np.random.seed(seeds[c])).  With nchains=1 and
            seeds=None the global numpy RNG is used, exactly like the reference.
            engine='device' (samplers 'amcmc', 'hmc' and 'mala'): states, proposal factors and history stay on the
            GPU, no host synchronisation per step (`quinn_amd.mcmc.device_amcmc`); same target and
            adaptation schedule, chains equal the host engine in distribution, not bit for bit.  The device AMCMC keeps at most
            `max_rows` (sampler_params; default 4096) distinct states per chain: on longer runs older states are compressed
            into max_rows / 4 pseudo-states with the same multiplicity total, the same mean and the scatter's dominant
            max_rows / 8 directions, so the adapted proposal covariance is then a low-rank APPROXIMATION of the reference's
            full empirical covariance (exact while rank <= max_rows / 8; `DeviceAMCMC._compress_history`).
            gather (multi-rank runs; chains are block-partitioned over the ranks): 'all' -- every rank ends with all
            chains (one all_gather of the result arrays, from the device buffers in bounded pieces); 'root' -- rank 0
            does, the other ranks keep their own shard; 'none' -- no communication at all.  A gather whose result
            exceeds `quinn_amd.parallel.DEFAULT_MAX_GATHER_BYTES` raises MemoryError before any traffic.
            gather_chain: the same choice for the `chain` array alone (None: as `gather`), e.g. gather='all',
            gather_chain='none' returns MAP / acceptance / log-posterior traces of every chain everywhere and leaves the
            states sharded (at cfg2 they are 43.6 GB per rank).
            bfgs_jac: None -- the `zflag` BFGS pre-fit lets scipy difference the log-posterior, exactly as the reference
            does (nn_mcmc.py:125-127; p + 1 log-posterior evaluations per gradient); 'device' -- the gradient kernel is
            passed as the analytic jacobian (one evaluation per gradient; a different start point than the reference's).
        """
        ntrn_, outdim = ytrn.shape
        assert xtrn.shape[0] == ntrn_
        self.lpinfo = {'model': None, 'xd': xtrn, 'yd': [y for y in ytrn], 'ltype': 'classical',
                       'lparams': {'sigma': datanoise}}
        if seeds is not None:
            seeds = list(seeds)
            nchains = len(seeds)
            rngs = [np.random.RandomState(s) for s in seeds]
        elif nchains > 1:
            raise ValueError("nchains > 1 needs seeds=[...] (one per chain)")
        else:
            rngs = None

        if param_ini is None:
            draw = (lambda r: r.rand(self.pdim)) if rngs else (lambda r: np.random.rand(self.pdim))
            inis = [draw(r) for r in (rngs or [None])]
            if zflag:
                # BFGS pre-fit of the random start (nn_mcmc.py:125-127); finite differences by default, like the reference
                if bfgs_jac not in (None, 'device'):
                    raise ValueError("bfgs_jac is None (scipy differences the log-posterior) or 'device'")
                jac = (lambda x, fcn, lpinfo: -self.logpostgrad(x, lpinfo)) if bfgs_jac == 'device' else None
                inis = [minimize((lambda x, fcn, lpinfo: -fcn(x, lpinfo)), ini, args=(self.logpost, self.lpinfo), jac=jac,
                                 method='BFGS', options={'gtol': 1e-13}).x for ini in inis]
            param_ini = np.stack(inis) if rngs else inis[0]
        param_ini = np.asarray(param_ini, dtype=np.float64)
        if rngs and param_ini.ndim == 1:
            param_ini = np.tile(param_ini, (nchains, 1))

        sampler_params = dict(sampler_params)      # None raises, as in the reference
        if engine == 'device':
            op = self._operator(self.lpinfo)
            ini2 = np.atleast_2d(param_ini)
            ctot = ini2.shape[0]
            rank, world = dist_info()
            lo, hi = shard_bounds(ctot) if world > 1 else (0, ctot)      # chains shard over ranks
            seed0 = seeds[0] if seeds else 0
            if sampler == 'amcmc':
                from ..mcmc.device_amcmc import DeviceAMCMC
                # random streams are keyed by the GLOBAL chain id: results do not depend on the number of ranks
                eng = DeviceAMCMC(op, datanoise, seed=seed0, chain0=lo, **sampler_params)
            elif sampler == 'hmc':
                from ..mcmc.device_hmc import DeviceHMC
                eng = DeviceHMC(op, datanoise, seed=seed0, chain0=lo, **sampler_params)
            elif sampler == 'mala':
                from ..mcmc.device_mala import DeviceMALA
                eng = DeviceMALA(op, datanoise, seed=seed0, chain0=lo, **sampler_params)
            else:
                raise ValueError("engine='device' is implemented for sampler='amcmc', 'hmc' and 'mala'")
            if hi > lo:
                res = eng.run(nmcmc, ini2[lo:hi], verbose=self.verbose and rank == 0)
            else:
                res = empty_results(nmcmc, ini2.shape[1])
            # the single collective, at the end, straight from the device tensors (world == 1: a device->host copy)
            res = gather_results(res, ctot, gather, gather_chain)
            self.mcmc_results = res
            if np.ndim(param_ini) == 1:
                self.mcmc_results = {k: v[0] for k, v in self.mcmc_results.items()}
            self.samples, self.cmode = self.mcmc_results['chain'], self.mcmc_results['mapparams']
            return
        if sampler == 'amcmc':
            mymcmc = AMCMC(**sampler_params)
            mymcmc.setLogPostBatch(self.logpost_batch, None, lpinfo=self.lpinfo)
        elif sampler == 'hmc':
            mymcmc = HMC(**sampler_params)
            mymcmc.setLogPostBatch(self.logpost_batch, self.logpostgrad_batch, lpinfo=self.lpinfo)
        elif sampler == 'mala':
            mymcmc = MALA(**sampler_params)
            mymcmc.setLogPostBatch(self.logpost_batch, self.logpostgrad_batch, lpinfo=self.lpinfo)
        else:
            raise ValueError(f"sampler {sampler!r} is not one of 'amcmc', 'hmc', 'mala'")

        if rngs is not None and dist_info()[1] > 1:
            # chains shard over ranks; one all_gather of the result arrays at the end
            self.mcmc_results = run_chains_sharded(lambda: mymcmc, nmcmc, param_ini, seeds, verbose=False, gather=gather, gather_chain=gather_chain)
        else:
            self.mcmc_results = mymcmc.run(nmcmc=nmcmc, param_ini=param_ini, rngs=rngs, verbose=self.verbose)
        self.samples, self.cmode = self.mcmc_results['chain'], self.mcmc_results['mapparams']

    # -- prediction --------------------------------------------------------------------------
    def get_best_model(self, param):
        """A copy of the module carrying the given flat weight vector (nn_mcmc.py:142-154)."""
        import torch
        mod = copy.deepcopy(self.nnmodel)
        s = 0
        with torch.no_grad():
            for p in mod.parameters():
                n = p.numel()
                p.copy_(torch.as_tensor(np.asarray(param[s:s + n])).view(p.shape).to(p.dtype))
                s += n
        return mod

    def predict_sample(self, x, param):
        """`(N,o)` prediction with one flat weight vector (nn_mcmc.py:168-178)."""
        return self._predict_batch(np.asarray(param).reshape(1, -1), x)[0]

    def predict_MAP(self, x):
        cm = self.cmode if np.ndim(self.cmode) == 1 else self.cmode[0]
        return self.predict_sample(x, cm)

    def _predict_ens_dev(self, x, nens=10, nburn=1000, chain=0):
        samples = self.samples if self.samples.ndim == 2 else self.samples[chain]
        nevery = int((samples.shape[0] - nburn) / nens)
        rows = [nburn + j * nevery for j in range(nens)]
        return self._predict_batch_dev(samples[rows, :], x)

    def predict_ens(self, x, nens=10, nburn=1000, chain=0):
        """`(M,N,o)`: predictions with M thinned post-burn-in samples, rows
        nburn + j*int((len-nburn)/nens) of the chain (nn_mcmc.py:194-199) -- one batched
        forward instead of M sequential ones.  `chain` picks the chain of a multi-chain fit."""
        return self._predict_ens_dev(x, nens, nburn, chain).double().cpu().numpy()
