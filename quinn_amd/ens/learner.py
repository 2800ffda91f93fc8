"""One ensemble member.

Mirror of the reference's `Learner` (quinn/ens/learner.py:8-93): holds a deep copy of the user
module, `fit` trains it (through the module's own `fit` if it has one, else `nnfit`) and keeps
the best model, `predict` evaluates the best model.  `NN_Ens` does not call `fit` member by
member: it trains all members in one batched run and fills the learners' results in.
"""
import copy
import math

import numpy as np

from ..nns.nnfit import nnfit
from ..nns.tchutils import print_nnparams


class Learner():
    def __init__(self, nnmodel, verbose=False):
        self.nnmodel = copy.deepcopy(nnmodel)
        self.trained = False
        self.verbose = verbose
        self._best_model = None
        self._best_w = None          # set by the batched ensemble trainer; the module is built on first use
        self.history = None
        if self.verbose:
            self.print_params(names_only=True)

    @property
    def best_model(self):
        """The best trained module (a copy of `nnmodel` carrying the best weights)."""
        if self._best_model is None and self._best_w is not None:
            from ..nns.nnfit import load_flat_into
            self._best_model = copy.deepcopy(self.nnmodel)
            load_flat_into(self._best_model, self._best_w)
        return self._best_model

    @best_model.setter
    def best_model(self, module):
        self._best_model = module
        self._pred_op = None

    def print_params(self, names_only=False):
        print_nnparams(self.best_model if self.trained else self.nnmodel, names_only=names_only)

    def init_params(self):
        """Uniform(-1/sqrt(fan_in), 1/sqrt(fan_in)) re-initialisation (learner.py:47-57; never
        called by the reference's own solvers)."""
        for p in self.nnmodel.parameters():
            try:
                stdv = 1. / math.sqrt(p.size(1))
            except IndexError:
                stdv = 1.
            p.data.uniform_(-stdv, stdv)

    def fit(self, xtrn, ytrn, **kwargs):
        if hasattr(self.nnmodel, 'fit') and callable(getattr(self.nnmodel, 'fit')):
            self.best_model = self.nnmodel.fit(xtrn, ytrn, **kwargs)
            self.history = getattr(self.nnmodel, 'history', None)
        else:
            fit_info = nnfit(self.nnmodel, xtrn, ytrn, **kwargs)
            self.best_model = fit_info['best_nnmodel']
            self.history = fit_info['history']
        self.trained = True

    def predict(self, x):
        """numpy `(N,d)` -> numpy `(N,o)` with the best model, evaluated by the device operator
        (`NN_Ens.predict_ens` batches all members into one launch)."""
        assert self.trained
        from ..ops import MLPArch, BatchedMLP, flatten_module
        x = np.asarray(x, dtype=np.float64)
        if getattr(self, "_pred_op", None) is None:
            self._pred_op = BatchedMLP(MLPArch.from_module(self.best_model), x, None)
        return self._pred_op.predict(flatten_module(self.best_model)[None, :], x)[0].double().cpu().numpy()
