"""CPU: the closed form the device AMCMC engine uses for the reference's covariance recursion."""
import numpy as np

from oracle.mcmc_ref import AmcmcState


def test_recursion_equals_sample_covariance_of_history():
    rs = np.random.RandomState(3)
    n, p = 73, 5
    x = (rs.randn(n + 1, p) * 0.2).cumsum(axis=0)
    st = AmcmcState(gamma=0.1, t0=10 ** 9)                     # never adapts: only the recursion runs
    rng = np.random.RandomState(0)
    for i in range(n + 1):
        st.propose(x[i], i, rng)
    np.testing.assert_allclose(st.cov, np.cov(x.T, ddof=1), rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(st.mean, x.mean(axis=0), rtol=1e-13)
    # windowed Gram form (what the device engine accumulates), shifted by x_0
    y = x - x[0]
    S2 = y[:40].T @ y[:40] + y[40:].T @ y[40:]
    s1 = y.sum(axis=0)
    cov = (S2 - np.outer(s1, s1) / (n + 1)) / n
    np.testing.assert_allclose(cov, st.cov, rtol=1e-10, atol=1e-12)


def test_initial_proposal_is_diag_plus_rank_one():
    x0 = np.array([0.5, -2.0, 0.0, 1.5])
    cov = 0.01 + np.diag(0.09 * np.abs(x0))                    # admcmc.py:65
    rs = np.random.RandomState(1)
    z, z0 = rs.randn(400000, 4), rs.randn(400000, 1)
    draws = np.sqrt(0.09 * np.abs(x0)) * z + 0.1 * z0
    np.testing.assert_allclose(np.cov(draws.T), cov, atol=3e-3)


def test_sample_space_draw_has_the_adapted_covariance():
    """delta = sqrt(c/(n-1)) sum_k sqrt(w_k) u_k (x_k - m) + sqrt(c eps) v over the DISTINCT states of a
    chain history has covariance c (cov + eps I) exactly (the identity behind qn_mcmc_propose_hist)."""
    rs = np.random.RandomState(7)
    p, K = 6, 23
    xk = rs.randn(K, p).cumsum(axis=0) * 0.3                    # distinct states
    w = rs.randint(1, 9, K)                                     # multiplicities (rejections repeat a state)
    hist = np.repeat(xk, w, axis=0)                             # the chain as the reference sees it
    n = hist.shape[0]
    cov = np.cov(hist.T, ddof=1)
    m = hist.mean(axis=0)
    A = np.sqrt(w)[:, None] * (xk - m)                          # delta_lr = A^T u / sqrt(n-1)
    np.testing.assert_allclose(A.T @ A / (n - 1), cov, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose((w[:, None] * xk).sum(axis=0) / n, m, rtol=1e-13)
    # and by simulation, with the isotropic floor
    c, eps = 0.37, 1e-2
    u, v = rs.randn(200000, K), rs.randn(200000, p)
    d = np.sqrt(c / (n - 1)) * (u @ A) + np.sqrt(c * eps) * v
    target = c * (cov + eps * np.eye(p))
    assert np.abs(np.cov(d.T) - target).max() < 0.02 * np.abs(target).max()
