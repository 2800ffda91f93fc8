"""Base class of the probabilistic wrappers.

Mirror of the reference's `QUiNNBase` (quinn/solvers/quinn.py:15-104): deep-copies the user
module, `predict_ens` stacks `predict_sample` draws, `predict_mom_sample` forms mean /
variance / covariance over that ensemble.  The matplotlib helpers of the reference
(`predict_plot`, `plot_1d_fits`, quinn.py:106-251) are presentation code outside the hot
path and are not reproduced.
"""
import copy
import sys

import numpy as np

from ..ops import MLPArch, BatchedMLP, flatten_module


def print_nnparams(nnmodel, names_only=False):
    for name, param in nnmodel.named_parameters():
        if names_only:
            print(f"{name}, shape {tuple(param.data.shape)}")
        else:
            print(name, param.data)


class QUiNNBase():
    """Args: nnmodel (torch.nn.Module): the user's MLP (never modified in place)."""

    def __init__(self, nnmodel, device=None, dtype="float64"):
        self.nnmodel = copy.deepcopy(nnmodel)
        self.nens = None
        self.arch = MLPArch.from_module(self.nnmodel)
        self._device = device
        self._dtype = dtype
        self._pred_op = None

    def print_params(self, names_only=False):
        print_nnparams(self.nnmodel, names_only=names_only)

    # -- device operator used for predictions (dataset = the query points) --------------
    def _predict_batch(self, W, x):
        """f_W(x) for a stack of flat weight vectors: numpy (M, N, o)."""
        x = np.asarray(x, dtype=np.float64)
        if self._pred_op is None:
            self._pred_op = BatchedMLP(self.arch, x, None, device=self._device, dtype=self._dtype)
        return self._pred_op.predict(W, x).double().cpu().numpy()

    def predict_sample(self, x):
        raise NotImplementedError

    def predict_ens(self, x, nens=None):
        """`(M, N, o)`: M draws of `predict_sample` (quinn.py:51-70)."""
        if nens is None:
            nens = self.nens
        return np.array([self.predict_sample(x) for _ in range(nens)])

    def predict(self, x):
        return self.predict_mom_sample(x)[0]

    def predict_mom_sample(self, x, msc=0, nsam=1000):
        """Mean `(N,o)`, variance `(N,o)` (ddof=1) and per-output covariance `(N,N,o)` of an
        `nsam`-member predictive ensemble; msc = 0 / 1 / 2 selects how much is computed
        (quinn.py:75-104)."""
        y = self.predict_ens(x, nens=nsam)
        _, nx, nout = y.shape
        ymean = np.mean(y, axis=0)
        if msc == 2:
            ycov = np.empty((nx, nx, nout))
            yvar = np.empty((nx, nout))
            for iout in range(nout):
                ycov[:, :, iout] = np.cov(y[:, :, iout], rowvar=False, ddof=1)
                yvar[:, iout] = np.diag(ycov[:, :, iout])
        elif msc == 1:
            ycov, yvar = None, np.var(y, axis=0, ddof=1)
        elif msc == 0:
            ycov, yvar = None, None
        else:
            print(f"msc={msc}, but needs to be 0,1, or 2. Exiting.")
            sys.exit()
        return ymean, yvar, ycov
