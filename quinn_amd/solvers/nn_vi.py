"""Variational-inference (Bayes-by-backprop) wrapper.

Mirror of the reference's `NN_VI` (quinn/solvers/nn_vi.py:14-132): builds a `BNet` around the
user module, `fit` sets `loss_params = [datanoise, nsam, num_batches]` with
`num_batches = ntrn if batch_size == 1 else (ntrn+1)//batch_size` (nn_vi.py:94-100) and runs
`nnfit` with `loss_xy = bmodel.viloss`; `best_model` is the BNet at the best validation loss;
`predict_sample` draws one weight sample from it.
"""
import numpy as np

from ..nns.nnfit import nnfit
from ..nns.tchutils import print_nnparams
from ..vi.bnet import BNet
from .quinn import QUiNNBase


class NN_VI(QUiNNBase):
    def __init__(self, nnmodel, verbose=False, pi=0.5, sigma1=1.0, sigma2=1.0, mu_init_lower=-0.2,
                 mu_init_upper=0.2, rho_init_lower=-5.0, rho_init_upper=-4.0, device=None, dtype="float64",
                 rng="reference"):
        super().__init__(nnmodel, device=device, dtype=dtype)
        self.bmodel = BNet(nnmodel, pi=pi, sigma1=sigma1, sigma2=sigma2, mu_init_lower=mu_init_lower,
                           mu_init_upper=mu_init_upper, rho_init_lower=rho_init_lower,
                           rho_init_upper=rho_init_upper, device=device, dtype=dtype, rng=rng)
        self.device = self.bmodel.device
        self.verbose = verbose
        self.trained = False
        self.best_model = None
        if self.verbose:
            print("=========== Deterministic model parameters ================")
            self.print_params(names_only=True)
            print("=========== Variational model parameters ==================")
            print_nnparams(self.bmodel, names_only=True)
            print("===========================================================")

    def fit(self, xtrn, ytrn, val=None, nepochs=600, lrate=0.01, batch_size=None, freq_out=100, freq_plot=1000,
            wd=0, cooldown=100, factor=0.95, nsam=1, scheduler_lr=None, datanoise=0.05):
        ntrn = xtrn.shape[0]
        assert ntrn == ytrn.shape[0]
        if batch_size is None or batch_size > ntrn:
            batch_size = ntrn
        num_batches = ntrn if batch_size == 1 else (ntrn + 1) // batch_size
        self.bmodel.loss_params = [datanoise, nsam, num_batches]
        fit_info = nnfit(self.bmodel, xtrn, ytrn, val=val, loss_xy=self.bmodel.viloss, lrate=lrate,
                         batch_size=batch_size, nepochs=nepochs, wd=wd, cooldown=cooldown, factor=factor,
                         freq_plot=freq_plot, scheduler_lr=scheduler_lr, freq_out=freq_out)
        self.fit_info = fit_info
        self.best_model = fit_info['best_nnmodel']
        self.trained = True

    def predict_sample(self, x):
        """`(N,o)` numpy prediction with one weight sample from the best variational posterior."""
        assert self.trained
        return self.best_model(np.asarray(x, dtype=np.float64), sample=True).cpu().numpy()

    def _predict_ens_dev(self, x, nens=None):
        if nens is None:
            nens = self.nens
        assert self.trained
        bm = self.best_model
        W, _, _ = bm._sample_kl(bm.mu, bm.rho, bm._draw_eps(nens))
        return bm.op.predict(W, np.asarray(x, dtype=np.float64))

    def predict_ens(self, x, nens=None):
        """`(M,N,o)`: M weight samples pushed through the network in ONE batched forward."""
        return self._predict_ens_dev(x, nens).double().cpu().numpy()
