"""GPU: the device-resident HMC engine (qn_hmc_begin / qn_hmc_leap / qn_hmc_accept around the batched gradient kernel).
The leapfrog arithmetic and the MH ratio are checked against the reference's formulas (quinn/mcmc/hmc.py:43-66,
mcmc.py:68-75) evaluated with the ORACLE's gradient on the kernel's own momenta; chains are checked for independence
from the way they are split over engines (what sharding over ranks does), for graph replay == direct launches, and in
distribution on a Gaussian posterior."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import mlp_ref
from quinn_amd import _lib
from quinn_amd.mcmc.device_hmc import DeviceHMC
from quinn_amd.ops import BatchedMLP, MLPArch

pytestmark = pytest.mark.gpu


def _problem(seed=0, N=48, d=1):
    rs = np.random.RandomState(seed)
    x = rs.rand(N, d) * 6 - 3
    y = np.sin(x).sum(axis=1, keepdims=True) + 0.1 * rs.randn(N, 1)
    return x, y


def _begin(op, cur, gcur, sigma, eps, chain0, seed, step):
    """qn_hmc_begin through the C ABI -> (mom, q, K_cur) on the host."""
    L = _lib.lib()
    C, p = cur.shape
    nk = L.qn_hmc_parts(p)
    mom, q = torch.empty_like(cur), torch.empty_like(cur)
    kparts = torch.empty(C, nk, dtype=torch.float64, device=cur.device)
    st = torch.zeros(2, dtype=torch.int64, device=cur.device)
    st[0] = step
    _lib.check(L.qn_hmc_begin(cur.data_ptr(), gcur.data_ptr(), sigma, eps, C, chain0, p, seed, st.data_ptr(), mom.data_ptr(),
                              q.data_ptr(), kparts.data_ptr(), None), "qn_hmc_begin")
    torch.cuda.synchronize()
    return mom.cpu().numpy(), q.cpu().numpy(), 0.5 * kparts.cpu().numpy().sum(axis=1)


@pytest.mark.parametrize("dims,L_", [((1, 8, 8, 1), 3), ((2, 16, 1), 1), ((1, 64, 64, 64, 1), 4)])
def test_one_step_equals_the_reference_leapfrog_on_the_kernel_momenta(dims, L_):
    x, y = _problem(N=40, d=dims[0])
    sigma, eps, seed, C = 0.2, 0.003, 1234, 5
    arch = MLPArch(dims, "tanh")
    op = BatchedMLP(arch, x, y)
    rs = np.random.RandomState(3)
    ini = 0.3 * rs.randn(C, arch.nparams)
    eng = DeviceHMC(op, sigma, epsilon=eps, L=L_, seed=seed, chain0=7)
    r = eng.run(1, ini)
    torch.cuda.synchronize()
    # the momenta the engine drew at step 0: rerun the begin kernel on the same (state, gradient, seed, step, chain0)
    cur = torch.as_tensor(ini, device="cuda")
    _, g0 = op.sse_grad(cur)
    mom, q1, kcur = _begin(op, cur, g0, sigma, eps, 7, eng.seed, 0)
    mod = mlp_ref.build_module(mlp_ref.MLPSpec(dims, "tanh"))
    yd = [v for v in y]
    lp = lambda w: mlp_ref.logpost(mod, w, x, yd, sigma)
    lpg = lambda w: mlp_ref.logpostgrad(mod, w, x, yd, sigma)
    for c in range(C):
        z = mom[c] - eps * lpg(ini[c]) / 2                          # undo the half kick: the N(0, I) draw itself
        assert abs(np.sum(np.square(z)) / 2 - kcur[c]) <= 1e-12 * kcur[c]
        # reference sampler (hmc.py:43-66) on this momentum
        qq, pp = ini[c].copy(), z.copy()
        k0 = np.sum(np.square(pp)) / 2
        pp += eps * lpg(qq) / 2
        for jj in range(L_):
            qq += eps * pp
            if jj != L_ - 1:
                pp += eps * lpg(qq)
        pp += eps * lpg(qq) / 2
        k1 = np.sum(np.square(-pp)) / 2
        mh = np.exp((-lp(ini[c]) + k0) - (-lp(qq) + k1))           # mcmc.py:68-72
        np.testing.assert_allclose(r['alphas'][c, 1].item(), mh, rtol=1e-8)
        got = r['chain'][c, 1].cpu().numpy()
        moved = not np.array_equal(got, ini[c])
        if moved:
            np.testing.assert_allclose(got, qq, rtol=1e-10, atol=1e-12)
            assert abs(r['logpost'][c, 1].item() - lp(qq)) <= 1e-10 * abs(lp(qq))
        assert abs(r['logpost'][c, 0].item() - lp(ini[c])) <= 1e-10 * abs(lp(ini[c]))
    assert np.all(r['alphas'][:, 0].cpu().numpy() == 0.0)


def test_momenta_are_standard_normal_and_keyed_by_global_chain_and_step():
    x, y = _problem()
    arch = MLPArch((1, 64, 64, 1), "tanh")
    op = BatchedMLP(arch, x, y)
    C, p = 64, arch.nparams
    cur = torch.zeros(C, p, dtype=torch.float64, device="cuda")
    g = torch.zeros_like(cur)                                        # zero gradient: mom is the draw itself
    z, q, k = _begin(op, cur, g, 0.2, 0.5, 0, 42, 5)
    v = z.ravel()
    n = v.size
    assert abs(v.mean()) < 5 / np.sqrt(n) and abs(v.var() - 1) < 5 * np.sqrt(2 / n) and abs((v ** 4).mean() - 3) < 5 * np.sqrt(96 / n)
    np.testing.assert_allclose(q, 0.5 * z, rtol=0, atol=0)            # first drift: q = cur + eps * mom
    np.testing.assert_allclose(k, 0.5 * (z ** 2).sum(axis=1), rtol=1e-13)
    # chains 10..19 of a launch that starts at chain 10 == chains 10..19 of the launch that starts at 0
    z10, _, _ = _begin(op, cur[:10], g[:10], 0.2, 0.5, 10, 42, 5)
    assert np.array_equal(z10, z[10:20])
    z_other_step, _, _ = _begin(op, cur, g, 0.2, 0.5, 0, 42, 6)
    assert not np.array_equal(z_other_step, z)


def test_chains_do_not_depend_on_how_they_are_split():
    x, y = _problem(N=64)
    arch = MLPArch((1, 16, 16, 1), "tanh")
    op = BatchedMLP(arch, x, y)
    C, nmcmc = 6, 40
    ini = np.stack([np.random.RandomState(50 + c).rand(arch.nparams) for c in range(C)])
    kw = dict(epsilon=0.002, L=3, seed=9)
    whole = DeviceHMC(op, 0.2, chain0=0, **kw).run(nmcmc, ini)
    a = DeviceHMC(op, 0.2, chain0=0, **kw).run(nmcmc, ini[:2])
    b = DeviceHMC(op, 0.2, chain0=2, **kw).run(nmcmc, ini[2:])
    for k in ('chain', 'logpost', 'alphas', 'accrate', 'mapparams', 'maxpost'):
        joined = torch.cat([a[k], b[k]]).cpu().numpy()
        # identical random numbers; only the summation order of a chain's SSE / gradient depends on the batch size
        np.testing.assert_allclose(joined, whole[k].cpu().numpy(), rtol=1e-7, atol=1e-9, err_msg=k)
    moved = lambda r: (r['chain'][:, 1:] != r['chain'][:, :-1]).any(dim=2).cpu().numpy()
    assert np.array_equal(np.concatenate([moved(a), moved(b)]), moved(whole))
    assert 0.3 < whole['accrate'].mean().item() <= 1.0


@pytest.mark.parametrize("engine", ["hmc", "mala"])
def test_chain_groups_on_the_fused_kernels_are_bit_identical(engine):
    """groups=2: two groups of chains on two HIP streams.  The groups' gradient launches split a chain's rows as the launch of all chains does
    (qn_mlp_desc_set_plan_batch), so every chain is the one-group chain bit for bit -- on the int8-slice kernels."""
    from quinn_amd import _lib
    from quinn_amd.mcmc.device_mala import DeviceMALA
    rs = np.random.RandomState(4)
    N, C = 2048, 64
    x = rs.rand(N, 1) * 2 - 1
    y = np.sin(3 * x) + 0.1 * rs.randn(N, 1)
    arch = MLPArch((1, 64, 64, 64, 1), "tanh")
    op = BatchedMLP(arch, x, y)
    assert op.arith(C, want_grad=True) == _lib.ARITH_I8_FUSED
    ini = np.stack([0.3 * np.random.RandomState(60 + c).randn(arch.nparams) for c in range(C)])
    mk = (lambda g, **kw: DeviceHMC(op, 0.1, epsilon=2e-4, L=3, seed=6, groups=g, **kw)) if engine == "hmc" else \
         (lambda g, **kw: DeviceMALA(op, 0.1, epsilon=4e-4, seed=6, groups=g, **kw))
    one = mk(1).run(25, ini)
    two = mk(2).run(25, ini)
    assert mk(None)._ngroups(C) == 1
    for k in ('chain', 'logpost', 'alphas', 'accrate', 'mapparams', 'maxpost'):
        assert torch.equal(one[k], two[k]), k
    acc = one['accrate'].mean().item()
    assert 0.05 < acc < 1.0, acc
    graph = mk(2, use_graph=True).run(25, ini, store_chain=False)
    assert graph['chain'] is None and torch.equal(graph['logpost'], one['logpost'])


def test_graph_replay_equals_direct_launches_bit_for_bit():
    x, y = _problem(N=64)
    arch = MLPArch((1, 16, 16, 1), "tanh")
    op = BatchedMLP(arch, x, y)
    ini = np.stack([np.random.RandomState(70 + c).rand(arch.nparams) for c in range(4)])
    for nmcmc in (11, 12):                                              # odd: the last step is launched directly
        d = DeviceHMC(op, 0.2, epsilon=0.002, L=2, seed=5).run(nmcmc, ini)
        g = DeviceHMC(op, 0.2, epsilon=0.002, L=2, seed=5, use_graph=True).run(nmcmc, ini)
        for k in d:
            assert torch.equal(d[k], g[k]), (nmcmc, k)


def test_float32_operator_runs_the_same_chain_to_float32_accuracy():
    x, y = _problem(N=64)
    arch = MLPArch((1, 16, 16, 1), "tanh")
    ini = np.stack([np.random.RandomState(80 + c).rand(arch.nparams) for c in range(3)])
    r64 = DeviceHMC(BatchedMLP(arch, x, y), 0.3, epsilon=0.002, L=3, seed=2).run(5, ini)
    r32 = DeviceHMC(BatchedMLP(arch, x, y, dtype="float32"), 0.3, epsilon=0.002, L=3, seed=2).run(5, ini)
    np.testing.assert_allclose(r32['logpost'].cpu().numpy(), r64['logpost'].cpu().numpy(), rtol=2e-4)


def test_gaussian_posterior_moments():
    """Linear model y = w x + b: the posterior over (w, b) is Gaussian with the least-squares mean and covariance
    sigma^2 (A^T A)^-1 -- the device chains must reproduce it (the reference's own sampler test is of this kind,
    tests/test_mcmc.py:93-109)."""
    rs = np.random.RandomState(1)
    N, sigma = 50, 0.5
    x = rs.randn(N, 1)
    y = 1.5 * x - 0.7 + sigma * rs.randn(N, 1)
    arch = MLPArch((1, 1), "identity")
    op = BatchedMLP(arch, x, y)
    A = np.hstack([x, np.ones((N, 1))])
    cov = sigma ** 2 * np.linalg.inv(A.T @ A)
    mean = np.linalg.solve(A.T @ A, A.T @ y).ravel()
    C, nmcmc = 256, 400
    ini = mean + 0.1 * rs.randn(C, 2)
    r = DeviceHMC(op, sigma, epsilon=0.03, L=8, seed=11).run(nmcmc, ini)
    ch = r['chain'][:, 100:].cpu().numpy().reshape(-1, 2)
    acc = r['accrate'].mean().item()
    assert 0.7 < acc <= 1.0
    se = np.sqrt(np.diag(cov) / (C * 10))                              # generous: ~10 effective samples per chain
    assert np.all(np.abs(ch.mean(axis=0) - mean) < 5 * se)
    np.testing.assert_allclose(np.cov(ch.T), cov, rtol=0.15, atol=0.1 * np.abs(cov).max())
    lp_map = r['maxpost'].cpu().numpy()
    assert np.all(lp_map >= r['logpost'].cpu().numpy().max(axis=1) - 1e-9)


def test_device_mala_step_equals_the_reference_langevin_step_on_the_kernel_momenta():
    """`DeviceMALA` (one leapfrog step of the HMC kernels) against the reference's `MALA.sampler` formulas
    (quinn/mcmc/mala.py:42-51) and MH ratio (mcmc.py:68-75), evaluated with the ORACLE's gradient on the momenta the kernel
    drew; two network shapes, one of them the BASELINE configs[1] shape (the int8-slice gradient kernel)."""
    from quinn_amd.mcmc.device_mala import DeviceMALA
    for dims, eps in (((1, 8, 8, 1), 0.01), ((1, 64, 64, 64, 1), 0.002)):
        x, y = _problem(N=40, d=dims[0])
        sigma, seed, C = 0.2, 4321, 4
        arch = MLPArch(dims, "tanh")
        op = BatchedMLP(arch, x, y)
        ini = 0.3 * np.random.RandomState(5).randn(C, arch.nparams)
        eng = DeviceMALA(op, sigma, epsilon=eps, seed=seed, chain0=3)
        assert eng.L == 1
        r = eng.run(1, ini)
        torch.cuda.synchronize()
        cur = torch.as_tensor(ini, device="cuda")
        _, g0 = op.sse_grad(cur)
        mom, q1, kcur = _begin(op, cur, g0, sigma, eps, 3, eng.seed, 0)
        mod = mlp_ref.build_module(mlp_ref.MLPSpec(dims, "tanh"))
        yd = [v for v in y]
        lp = lambda w: mlp_ref.logpost(mod, w, x, yd, sigma)
        lpg = lambda w: mlp_ref.logpostgrad(mod, w, x, yd, sigma)
        for c in range(C):
            g_cur = lpg(ini[c])
            pz = mom[c] - eps * g_cur / 2                               # undo the half kick: the N(0, I) draw itself
            prop = ini[c] + 0.5 * eps ** 2 * g_cur + eps * pz           # mala.py:45
            g_prop = lpg(prop)
            k0 = np.sum(np.square(pz)) / 2                              # mala.py:48
            k1 = np.sum(np.square(pz + eps * (g_cur + g_prop) / 2)) / 2   # mala.py:50-51
            np.testing.assert_allclose(q1[c], prop, rtol=1e-12, atol=1e-14)
            mh = np.exp((-lp(ini[c]) + k0) - (-lp(prop) + k1))
            np.testing.assert_allclose(r['alphas'][c, 1].item(), mh, rtol=1e-8)
            got = r['chain'][c, 1].cpu().numpy()
            if not np.array_equal(got, ini[c]):
                np.testing.assert_allclose(got, prop, rtol=1e-10, atol=1e-12)


def test_device_mala_through_the_solver_matches_the_host_sampler_in_distribution():
    """`NN_MCMC.fit(sampler='mala', engine='device')` against the host `MALA` on a small posterior: acceptance rate and the
    stationary level of the log-posterior agree (different random streams: in distribution, not bit for bit)."""
    from quinn_amd.nns.mlp import MLP
    from quinn_amd.solvers.nn_mcmc import NN_MCMC
    x, y = _problem(seed=2, N=64)
    res = {}
    for engine in ("host", "device"):
        torch.manual_seed(0)
        nn = MLP(1, 1, (6,), activ="tanh")
        s = NN_MCMC(nn, verbose=False)
        np.random.seed(11)
        s.fit(x, y, zflag=False, datanoise=0.3, nmcmc=1500, param_ini=0.1 * np.ones(s.pdim), sampler="mala",
              sampler_params={"epsilon": 0.02}, nchains=8, seeds=list(range(50, 58)), engine=engine)
        r = s.mcmc_results
        res[engine] = (np.mean(r["accrate"]), np.mean(np.asarray(r["logpost"])[:, 700:]))
    (ah, lh), (ad, ld) = res["host"], res["device"]
    assert 0.3 < ah < 1.0 and abs(ah - ad) < 0.08, (ah, ad)
    assert abs(lh - ld) < 1.5, (lh, ld)
