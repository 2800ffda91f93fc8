"""Deep ensemble: all members trained and evaluated as one batch of weight vectors.

Mirror of the reference's `NN_Ens` (quinn/solvers/nn_ens.py:8-127): members are deep copies of
ONE module (identical initial weights, learner.py:28); member j trains on the rows
`np.random.permutation(ntrn)[:int(ntrn*dfrac)]` (nn_ens.py:63-64).  The reference trains the
members one after another; here every forward / backward / Adam step covers all of them
(`quinn_amd.nns.nnfit.fit_members`), with the random draws (numpy permutations, torch
randperm per epoch) consumed in the reference's member-major order so that trajectories match.
"""
import copy

import numpy as np

from ..ens.learner import Learner
from ..nns.nnfit import fit_members, load_flat_into, draw_perms
from ..parallel import shard_bounds, all_gather_rows, dist_info
from ..ops import flatten_module
from .quinn import QUiNNBase


class NN_Ens(QUiNNBase):
    def __init__(self, nnmodel, nens=1, dfrac=1.0, verbose=False, device=None, dtype="float64"):
        super().__init__(nnmodel, device=device, dtype=dtype)
        self.verbose = verbose
        self.nens = nens
        self.dfrac = dfrac
        self.learners = [Learner(nnmodel) for _ in range(nens)]
        if self.verbose:
            self.print_params(names_only=True)

    def print_params(self, names_only=False):
        for i, learner in enumerate(self.learners):
            print(f"==========  Learner {i+1}/{self.nens}  ============")
            learner.print_params(names_only=names_only)

    def fit(self, xtrn, ytrn, **kwargs):
        """Train every member (keyword arguments as `nnfit`: val, lrate, batch_size, nepochs, wd,
        optimizer, loss_fn, datanoise, lmbd, freq_out, ...; build-only: perm_mode)."""
        ntrn = ytrn.shape[0]
        rows = np.stack([np.random.permutation(ntrn)[:int(ntrn * self.dfrac)] for _ in range(self.nens)])
        val = kwargs.pop('val', None)
        # no validation set: every member validates on its own training subset (nnfit.py:106-109 sees
        # xtrn[ind_this] and copies it)
        xval, yval = (None, None) if val is None else val
        for k in ('freq_plot', 'lhist_suffix', 'gradcheck', 'lossparams'):
            kwargs.pop(k, None)
        if kwargs.pop('priorparams', None) is not None:
            raise NotImplementedError("use NN_RMS for a Gaussian prior")
        w0 = flatten_module(self.learners[0].nnmodel)
        nepochs = kwargs.pop('nepochs', 5000)
        # members shard over ranks (torch.distributed); every rank consumes the random streams of ALL
        # members in the reference's order, trains its block, and ONE all_gather returns the results
        lo, hi = shard_bounds(self.nens)
        perms = None
        if kwargs.get('perm_mode', 'reference') == 'reference':
            perms = draw_perms(self.nens, nepochs, rows.shape[1])[lo:hi]
        res = fit_members(self.arch, np.tile(w0, (hi - lo, 1)), xtrn, ytrn, rows[lo:hi], xval, yval, nepochs,
                          kwargs.pop('batch_size', None), device=self._device, dtype=self._dtype,
                          verbose=self.verbose and dist_info()[0] == 0, perms=perms, **kwargs)
        res = {k: all_gather_rows(v, self.nens) for k, v in res.items()}
        self.fit_results = res
        self._best_w = res['best_w']
        for j, learner in enumerate(self.learners):
            load_flat_into(learner.nnmodel, res['final_w'][j])
            learner._best_model, learner._best_w, learner._pred_op = None, res['best_w'][j], None
            learner.history = [list(r) for r in res['history'][j]]
            if hasattr(learner.nnmodel, 'history'):
                learner.nnmodel.history = learner.history
            learner.trained = True

    def predict_sample(self, x):
        """Prediction of one randomly selected member (nn_ens.py:72-82)."""
        jens = np.random.randint(0, self.nens)
        return self._predict_batch(self._best_w[jens:jens + 1], x)[0]

    def _predict_ens_dev(self, x, nens=None):
        if nens is None:
            nens = self.nens
        if nens > self.nens:
            print(f"Warning: Requested {nens} but only {self.nens} ensemble members available.")
            nens = self.nens
        order = np.random.permutation(nens)
        return self._predict_batch_dev(self._best_w[order], x)

    def predict_ens(self, x, nens=None):
        """`(M,N,o)`: predictions of (a random permutation of) the members (nn_ens.py:85-110)."""
        return self._predict_ens_dev(x, nens).double().cpu().numpy()

    def predict_ens_fromsamples(self, x, nens=1):
        return np.array([self.predict_sample(x) for _ in range(nens)])
