"""Base class of the probabilistic wrappers.

Mirror of the reference's `QUiNNBase` (quinn/solvers/quinn.py:15-104): deep-copies the user
module, `predict_ens` stacks `predict_sample` draws, `predict_mom_sample` forms mean /
variance / covariance over that ensemble.  The matplotlib helpers of the reference
(`predict_plot`, `plot_1d_fits`, quinn.py:106-251) are presentation code outside the hot
path and are not reproduced.
"""
import copy
import ctypes
import sys

import numpy as np
import torch

from .. import _lib
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
    def _predict_batch_dev(self, W, x):
        """f_W(x) for a stack of flat weight vectors: device tensor (M, N, o) in the compute dtype."""
        x = np.asarray(x, dtype=np.float64)
        if self._pred_op is None:
            self._pred_op = BatchedMLP(self.arch, x, None, device=self._device, dtype=self._dtype)
        return self._pred_op.predict(W, x)

    def _predict_batch(self, W, x):
        """f_W(x) for a stack of flat weight vectors: numpy (M, N, o)."""
        return self._predict_batch_dev(W, x).double().cpu().numpy()

    def predict_sample(self, x):
        raise NotImplementedError

    def _predict_ens_dev(self, x, nens):
        """The predictive ensemble as a DEVICE tensor (M, N, o).  Solvers whose members come out of one batched forward
        override this (no host round trip); the default takes whatever `predict_ens` of the solver returns."""
        return torch.as_tensor(self.predict_ens(x, nens=nens))

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
        (quinn.py:75-104).  The ensemble stays on the device: mean / variance by `qn_pred_moments`, the
        covariance as one float64 GEMM of the centred ensemble per output; only the moments are downloaded."""
        if msc not in (0, 1, 2):
            print(f"msc={msc}, but needs to be 0,1, or 2. Exiting.")
            sys.exit()
        dev = torch.device(self._device) if self._device is not None else torch.device("cuda", torch.cuda.current_device())
        y = self._predict_ens_dev(x, nsam).to(dev).contiguous()
        M, nx, nout = y.shape
        mean = torch.empty(nx, nout, dtype=torch.float64, device=dev)
        var = torch.empty(nx, nout, dtype=torch.float64, device=dev) if msc == 1 else None
        qdt = _lib.QN_F32 if y.dtype == torch.float32 else _lib.QN_F64
        if y.dtype not in (torch.float32, torch.float64):
            y, qdt = y.double(), _lib.QN_F64
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().qn_pred_moments(y.data_ptr(), qdt, M, nx * nout, mean.data_ptr(),
                                                  var.data_ptr() if var is not None else None,
                                                  ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "qn_pred_moments")
        ymean = mean.cpu().numpy()
        if msc == 2:
            yc = y.double() - mean                                   # (M, N, o)
            ycov_d = torch.empty(nx, nx, nout, dtype=torch.float64, device=dev)
            for iout in range(nout):
                a = yc[:, :, iout]
                ycov_d[:, :, iout] = (a.T @ a) / (M - 1)            # np.cov(rowvar=False, ddof=1), quinn.py:88-90
            ycov = ycov_d.cpu().numpy()
            yvar = np.stack([np.diag(ycov[:, :, iout]) for iout in range(nout)], axis=1)
        elif msc == 1:
            ycov, yvar = None, var.cpu().numpy()
        else:
            ycov, yvar = None, None
        return ymean, yvar, ycov

    # -- figures (presentation only; same signatures and file names as quinn.py:106-260) ---------------
    @staticmethod
    def _center_and_spread(yens, quantiles):
        """(centre, lower, upper) of an ensemble: mean and +-std, or median and the distances to the quartiles."""
        if quantiles:
            q25, q50, q75 = np.quantile(yens, [0.25, 0.5, 0.75], axis=0)
            return q50, q50 - q25, q75 - q50
        sd = np.std(yens, axis=0)
        return np.mean(yens, axis=0), sd, sd

    def predict_plot(self, xx_list, yy_list, nmc=100, plot_qt=False, labels=None, colors=None, iouts=None, msize=14,
                     figname=None):
        """Predicted-vs-data ('diagonal') figure per output, error bars from an `nmc`-member ensemble;
        saved as `fitdiag_o<iout>.png` unless `figname` is given."""
        import matplotlib
        matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt
        assert len(xx_list) == len(yy_list)
        stats = [self._center_and_spread(self.predict_ens(xx, nens=nmc), plot_qt) for xx in xx_list]
        nset, nout = len(xx_list), stats[0][0].shape[1]
        labels = labels or [f'Set {i + 1}' for i in range(nset)]
        colors = colors or (['b', 'g', 'r', 'c', 'm', 'y'] * nset)[:nset]
        for iout in (range(nout) if iouts is None else iouts):
            plt.figure(figsize=(10, 10))
            lo = min(float(yy[:, iout].min()) for yy in yy_list)
            hi = max(float(yy[:, iout].max()) for yy in yy_list)
            plt.plot([lo, hi], [lo, hi], 'k--', linewidth=1)
            for (mid, dl, du), yy, lab, col in zip(stats, yy_list, labels, colors):
                plt.errorbar(yy[:, iout], mid[:, iout], yerr=[dl[:, iout], du[:, iout]], fmt=col + 'o', markersize=msize / 2,
                             markeredgecolor='w', label=lab)
            plt.xlabel(f'Model output # {iout + 1}')
            plt.ylabel(f'Fit output # {iout + 1}')
            plt.legend()
            plt.savefig(figname or f'fitdiag_o{iout}.png')
            plt.close()

    def plot_1d_fits(self, xx_list, yy_list, domain=None, ngr=111, plot_qt=False, nmc=100, true_model=None,
                     labels=None, colors=None, name_postfix=''):
        """One-dimensional slices of the fit through the middle of the domain, one figure per (input, output),
        saved as `fit_d<idim>_o<iout>_<name_postfix>.png`: data, ensemble members, centre and spread band."""
        import matplotlib
        matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt
        assert len(xx_list) == len(yy_list)
        nset = len(xx_list)
        labels = labels or [f'Set {i + 1}' for i in range(nset)]
        colors = colors or (['b', 'g', 'r', 'c', 'm', 'y'] * nset)[:nset]
        if domain is None:
            xall = np.vstack(xx_list)
            domain = np.stack([xall.min(axis=0), xall.max(axis=0)], axis=1)
        ndim, nout = xx_list[0].shape[1], yy_list[0].shape[1]
        mid_label, band_label = ('Median Pred.', 'Qtile') if plot_qt else ('Mean Pred.', 'St.Dev.')
        for idim in range(ndim):
            unit = np.full((ngr, ndim), 0.5)
            unit[:, idim] = np.linspace(0.0, 1.0, ngr)
            xgrid = unit * (domain[:, 1] - domain[:, 0]) + domain[:, 0]
            yens = self.predict_ens(xgrid, nens=nmc)
            mid, dl, du = self._center_and_spread(yens, plot_qt)
            for iout in range(nout):
                plt.figure(figsize=(12, 8))
                for xx, yy, lab, col in zip(xx_list, yy_list, labels, colors):
                    plt.plot(xx[:, idim], yy[:, iout], col + 'o', markersize=13, markeredgecolor='w', label=lab, zorder=1000)
                if true_model is not None:
                    plt.plot(xgrid[:, idim], true_model(xgrid, 0.0)[:, iout], 'k-', alpha=0.5, label='Truth')
                for member in yens:
                    plt.plot(xgrid[:, idim], member[:, iout], 'm--', linewidth=1, zorder=-10000)
                plt.plot(xgrid[:, idim], mid[:, iout], 'm-', linewidth=5, label=mid_label)
                plt.fill_between(xgrid[:, idim], mid[:, iout] - dl[:, iout], mid[:, iout] + du[:, iout], color='plum',
                                 alpha=0.9, zorder=-1000, label=band_label)
                plt.legend()
                plt.xlabel(f'Input # {idim + 1}')
                plt.ylabel(f'Output # {iout + 1}')
                plt.savefig(f'fit_d{idim}_o{iout}_{name_postfix}.png')
                plt.close()
