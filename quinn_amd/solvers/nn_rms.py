"""Randomised-MAP-sampling ensemble (anchored ensembling, Pearce et al. 2018).

Mirror of the reference's `NN_RMS` (quinn/solvers/nn_rms.py:10-56): an `NN_Ens` whose members
minimise the negative log-posterior with a Gaussian prior centred on a per-member random anchor
`torch.randn(p) * priorsigma` (nn_rms.py:53).  All members train in one batched run; the anchor
draws and the per-epoch permutations are consumed from torch's global generator in the reference's
member-major order (anchor of member j, then member j's epochs).
"""
import copy

import numpy as np
import torch

from ..nns.nnfit import fit_members, load_flat_into
from ..ops import flatten_module
from ..parallel import shard_bounds, all_gather_rows, dist_info
from .nn_ens import NN_Ens


class NN_RMS(NN_Ens):
    def __init__(self, nnmodel, datanoise=0.1, priorsigma=1.0, **kwargs):
        super().__init__(nnmodel, **kwargs)
        self.datanoise = datanoise
        self.priorsigma = priorsigma
        self.nparams = sum(p.numel() for p in self.nnmodel.parameters())

    def fit(self, xtrn, ytrn, **kwargs):
        ntrn = ytrn.shape[0]
        rows = np.stack([np.random.permutation(ntrn)[:int(ntrn * self.dfrac)] for _ in range(self.nens)])
        val = kwargs.pop('val', None)
        xval, yval = (None, None) if val is None else val      # None: members validate on their own rows
        for k in ('freq_plot', 'lhist_suffix', 'gradcheck', 'lossparams', 'loss_fn', 'datanoise', 'priorparams'):
            kwargs.pop(k, None)
        nepochs = kwargs.pop('nepochs', 5000)
        nsub = rows.shape[1]
        anchors = np.empty((self.nens, self.nparams))
        perms = None
        if kwargs.get('perm_mode', 'reference') == 'reference':
            perms = np.empty((self.nens, nepochs, nsub), dtype=np.int64)
        for j in range(self.nens):                              # reference draw order, member-major
            anchors[j] = (torch.randn(size=(self.nparams,), dtype=torch.float64) * self.priorsigma).numpy()
            if perms is not None:
                for t in range(nepochs):
                    perms[j, t] = torch.randperm(nsub).numpy()
        lo, hi = shard_bounds(self.nens)
        w0 = flatten_module(self.learners[0].nnmodel)
        res = fit_members(self.arch, np.tile(w0, (hi - lo, 1)), xtrn, ytrn, rows[lo:hi], xval, yval, nepochs,
                          kwargs.pop('batch_size', None), loss_fn='logpost', datanoise=self.datanoise,
                          anchors=anchors[lo:hi], prior_sigma=self.priorsigma, device=self._device, dtype=self._dtype,
                          verbose=self.verbose and dist_info()[0] == 0,
                          perms=None if perms is None else perms[lo:hi], **kwargs)
        res = {k: all_gather_rows(v, self.nens) for k, v in res.items()}
        self.fit_results, self._best_w, self.anchors = res, res['best_w'], anchors
        for j, learner in enumerate(self.learners):
            load_flat_into(learner.nnmodel, res['final_w'][j])
            learner._best_model, learner._best_w, learner._pred_op = None, res['best_w'][j], None
            learner.history = [list(r) for r in res['history'][j]]
            learner.trained = True
