#!/usr/bin/env python
"""The reference's `examples/ex_ufit.py` call pattern on the MI355X path.

    python examples/ex_ufit.py {amcmc|hmc|vi|ens|rms} [--quick] [--mlp]

Same data generation, same network (`RNet(3, 3, wp_function=Poly(0), ...)`, examples/ex_ufit.py:72-77
there; `--mlp` switches to the commented-out MLP alternative), same solver calls and keyword
arguments as the reference example (examples/ex_ufit.py:40-115); differences: many chains run at
once (`seeds=`), and the matplotlib output is replaced by a printed summary of the predictive
mean / standard deviation.
"""
import sys

import numpy as np
import torch

from quinn_amd.nns.mlp import MLP
from quinn_amd.nns.rnet import RNet, Poly
from quinn_amd.solvers.nn_ens import NN_Ens
from quinn_amd.solvers.nn_rms import NN_RMS
from quinn_amd.solvers.nn_mcmc import NN_MCMC
from quinn_amd.solvers.nn_vi import NN_VI


def scale01ToDom(xx, dom):
    return xx * np.abs(dom[:, 1] - dom[:, 0]) + np.min(dom, axis=1)


def Sine(xx, datanoise=0.0):
    yy = datanoise * np.random.randn(xx.shape[0], 1)
    yy += np.sum(np.sin(xx), axis=1).reshape(-1, 1)
    return yy


def main(meth, quick=False, mlp=False):
    torch.set_default_dtype(torch.double)
    all_uq_options = ['amcmc', 'hmc', 'vi', 'ens', 'rms']
    assert meth in all_uq_options, f'Pick among {all_uq_options}'
    nall, trn_factor, ntst, ndim, datanoise = 15, 0.9, 13, 1, 0.02
    domain = np.tile(np.array([-np.pi, np.pi]), (ndim, 1))
    xall = scale01ToDom(np.random.rand(nall, ndim), domain)
    yall = Sine(xall, datanoise=datanoise)
    np.random.seed(100)
    xtst = scale01ToDom(np.random.rand(ntst, ndim), domain)
    ytst = Sine(xtst, datanoise=datanoise)
    if mlp:
        nnet = MLP(ndim, 1, (11, 11, 11), biasorno=True, activ='tanh')
    else:
        nnet = RNet(3, 3, wp_function=Poly(0), indim=ndim, outdim=1, layer_pre=True, layer_post=True,
                    biasorno=True, nonlin=True, mlp=False, final_layer=None)
    ntrn = int(trn_factor * nall)
    xtrn, xval = xall[:ntrn, :], xall[ntrn:, :]
    ytrn, yval = yall[:ntrn, :], yall[ntrn:, :]
    k = 20 if quick else 1

    if meth == 'amcmc':
        uqnet = NN_MCMC(nnet, verbose=not quick)
        uqnet.fit(xtrn, ytrn, zflag=False, datanoise=datanoise, nmcmc=10000 // k, sampler='amcmc',
                  sampler_params={'gamma': 0.01}, seeds=range(8))
        predict = lambda x: uqnet.predict_ens(x, nens=100 // k * 2, nburn=1000 // k, chain=0)
    elif meth == 'hmc':
        uqnet = NN_MCMC(nnet, verbose=not quick)
        uqnet.fit(xtrn, ytrn, zflag=False, datanoise=datanoise, nmcmc=10000 // k, sampler='hmc',
                  sampler_params={'L': 3, 'epsilon': 0.0025}, seeds=range(8))
        predict = lambda x: uqnet.predict_ens(x, nens=100 // k * 2, nburn=1000 // k, chain=0)
    elif meth == 'vi':
        uqnet = NN_VI(nnet, verbose=not quick)
        uqnet.fit(xtrn, ytrn, val=[xval, yval], datanoise=datanoise, lrate=0.01, batch_size=None, nsam=1,
                  nepochs=5000 // k, freq_out=1000)
        predict = lambda x: uqnet.predict_ens(x, nens=111)
    elif meth == 'ens':
        uqnet = NN_Ens(nnet, nens=3, dfrac=0.8, verbose=not quick)
        uqnet.fit(xtrn, ytrn, val=[xval, yval], lrate=0.01, batch_size=2, nepochs=1000 // k, freq_out=1000)
        predict = lambda x: uqnet.predict_ens(x)
    else:
        uqnet = NN_RMS(nnet, nens=7, dfrac=1.0, verbose=not quick, datanoise=datanoise, priorsigma=0.1)
        uqnet.fit(xtrn, ytrn, val=[xval, yval], lrate=0.01, batch_size=2, nepochs=1000 // k, freq_out=1000)
        predict = lambda x: uqnet.predict_ens(x)

    xgrid = scale01ToDom(np.linspace(0.0, 1.0, 11), domain)[:, np.newaxis]
    y = predict(xgrid)
    ymean, ystd = y.mean(axis=0)[:, 0], y.std(axis=0, ddof=1)[:, 0]
    print(f"{meth}: {y.shape[0]} predictive samples on an 11-point grid")
    for xg, m, s, t in zip(xgrid[:, 0], ymean, ystd, np.sin(xgrid[:, 0])):
        print(f"  x={xg:+.3f}  mean={m:+.4f}  std={s:.4f}  truth={t:+.4f}")
    rmse = float(np.sqrt(np.mean((uqnet.predict_ens(xtst, nens=y.shape[0]).mean(axis=0) - ytst) ** 2))) \
        if meth in ('vi', 'ens', 'rms') else float(np.sqrt(np.mean((predict(xtst).mean(axis=0) - ytst) ** 2)))
    print(f"  test RMSE of the predictive mean: {rmse:.4f}")
    return ymean, ystd, rmse


if __name__ == '__main__':
    torch.manual_seed(0)
    np.random.seed(0)
    main(sys.argv[1], quick='--quick' in sys.argv, mlp='--mlp' in sys.argv)
