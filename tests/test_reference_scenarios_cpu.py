"""CPU: the scenarios of the reference's tests/test_mcmc.py, run on this package's sampler classes with
plain Python log-posterior callables (the samplers are host code; no GPU involved).  Same constructor
arguments, calls and assertions as the reference's tests, so a user of `quinn.mcmc` finds the same contract."""
import numpy as np

from quinn_amd.mcmc.admcmc import AMCMC
from quinn_amd.mcmc.hmc import HMC
from quinn_amd.mcmc.mala import MALA


def gaussian(mean, cov):
    prec = np.linalg.inv(cov)
    return (lambda x: -0.5 * (x - mean) @ prec @ (x - mean)), (lambda x: -prec @ (x - mean))


def test_amcmc_defaults():                                   # test_mcmc.py:25-30
    s = AMCMC()
    assert (s.gamma, s.t0, s.tadapt) == (0.1, 100, 1000)


def test_amcmc_samples_2d_gaussian():                        # test_mcmc.py:33-53
    np.random.seed(42)
    mean = np.array([1.0, 2.0])
    lp, _ = gaussian(mean, np.array([[1.0, 0.3], [0.3, 1.0]]))
    s = AMCMC(gamma=0.5, t0=50, tadapt=100)
    s.setLogPost(lp, None)
    res = s.run(3000, np.zeros(2))
    assert {'chain', 'mapparams', 'accrate'} <= set(res)
    assert res['chain'].shape[1] == 2
    assert np.allclose(res['mapparams'], mean, atol=0.5)


def test_amcmc_acceptance_rate_is_reasonable():              # test_mcmc.py:56-70
    np.random.seed(42)
    lp, _ = gaussian(np.array([0.0]), np.array([[1.0]]))
    s = AMCMC(gamma=0.5)
    s.setLogPost(lp, None)
    res = s.run(2000, np.array([0.0]))
    assert 0.05 < res['accrate'] < 0.95


def test_amcmc_result_shapes():                              # test_mcmc.py:73-90
    np.random.seed(42)
    lp, _ = gaussian(np.zeros(3), np.eye(3))
    s = AMCMC(gamma=0.5)
    s.setLogPost(lp, None)
    res = s.run(500, np.zeros(3))
    assert res['chain'].shape == (501, 3)
    assert res['logpost'].shape == (501,) and res['alphas'].shape == (501,)


def test_hmc_samples_2d_gaussian():                          # test_mcmc.py:93-109
    np.random.seed(42)
    mean = np.array([1.0, 2.0])
    lp, lg = gaussian(mean, np.eye(2))
    s = HMC(epsilon=0.1, L=10)
    s.setLogPost(lp, lg)
    res = s.run(1000, np.zeros(2))
    assert res['chain'].shape[1] == 2
    assert np.allclose(res['mapparams'], mean, atol=0.5)


def test_mala_runs():                                        # test_mcmc.py:112-126
    np.random.seed(42)
    lp, lg = gaussian(np.array([1.0, 2.0]), np.eye(2))
    s = MALA(epsilon=0.1)
    s.setLogPost(lp, lg)
    assert s.run(1000, np.zeros(2))['chain'].shape[1] == 2


def test_maxpost_is_the_largest_stored_logpost():            # test_mcmc.py:129-143
    np.random.seed(42)
    lp, _ = gaussian(np.zeros(2), np.eye(2))
    s = AMCMC(gamma=0.5)
    s.setLogPost(lp, None)
    res = s.run(500, np.zeros(2))
    assert res['maxpost'] >= res['logpost'].max() - 1e-10


def test_amcmc_custom_initial_covariance():                  # test_mcmc.py:146-162
    np.random.seed(42)
    lp, _ = gaussian(np.zeros(2), np.eye(2))
    s = AMCMC(cov_ini=0.01 * np.eye(2), gamma=0.5)
    s.setLogPost(lp, None)
    res = s.run(500, np.zeros(2))
    assert res['chain'].shape == (501, 2)
