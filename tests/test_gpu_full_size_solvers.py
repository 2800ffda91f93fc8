"""GPU: solver-level checks at the sizes of BASELINE configs 3 and 5 (the operator-level full-size checks are in
test_gpu_full_size.py).
  * cfg3 (quinn/vi, S = 128 Monte-Carlo samples, 3 x 128 MLP, p = 33 537, 2-D inputs): one `viloss` evaluation + gradient
    through qn_vi_sample_kl / qn_mlp_sse_fwdbwd / qn_vi_grad on a FIXED epsilon against oracle/vi_ref.py (the restated
    bnet.py:181-232) on a 384-row subsample, so that the oracle's 128 sequential forwards take seconds; on all 8192 rows
    the data term is checked through its additivity over row blocks.
  * cfg5 (HMC, 4 x 256 MLP, N = 32 768, L = 10): one device-HMC proposal of 8 chains against the reference's leapfrog
    (hmc.py:43-66) driven by the SAME momenta, with gradients from the operator."""
import numpy as np
import pytest
import torch

from oracle import mlp_ref, vi_ref
from quinn_amd import _lib
from quinn_amd.mcmc.device_hmc import DeviceHMC
from quinn_amd.nns.mlp import MLP
from quinn_amd.ops import BatchedMLP, MLPArch, neg_log_post_from_sse
from quinn_amd.vi.bnet import BNet

pytestmark = pytest.mark.gpu


def _vi_problem():
    rs = np.random.RandomState(0)
    N = 8192
    x = rs.rand(N, 2) * 2 * np.pi - np.pi
    y = np.sin(x).sum(axis=1, keepdims=True) + 0.05 * rs.randn(N, 1)
    torch.manual_seed(0)
    net = MLP(2, 1, (128, 128, 128), activ='tanh')
    return x, y, net


def test_cfg3_viloss_and_gradient_vs_oracle_on_fixed_eps():
    x, y, net = _vi_problem()
    S, datanoise, nb = 128, 0.05, 1
    bm = BNet(net)
    p = bm.p
    assert p == 33537
    rs = np.random.RandomState(1)
    mu = 0.2 * rs.uniform(-1, 1, p)
    rho = rs.uniform(-5, -4, p)
    eps = rs.randn(S, p)
    with torch.no_grad():
        bm.theta.copy_(torch.as_tensor(np.concatenate([mu, rho]), device=bm.theta.device))
    bm.loss_params = [datanoise, S, nb]

    def device_loss(xs, ys):
        bm._draw_eps = lambda n: torch.as_tensor(eps, device=bm.device)
        if bm.theta.grad is not None:
            bm.theta.grad = None
        loss = bm.viloss(xs, ys)
        loss.backward()
        g = bm.theta.grad.detach().cpu().numpy().copy()
        return loss.item(), g[:p], g[p:]

    # ---- parity on a row subsample (all 128 samples, all 33 537 parameters)
    rows = rs.choice(len(x), 384, replace=False)
    xs, ys = x[rows], y[rows]
    spec = mlp_ref.MLPSpec((2, 128, 128, 128, 1), "tanh")
    ref = vi_ref.viloss(spec, mu, rho, eps, xs, ys, datanoise, nb)
    loss, dmu, drho = device_loss(xs, ys)
    assert abs(loss - ref["loss"]) <= 1e-11 * abs(ref["loss"]), (loss, ref["loss"])
    sc = max(np.abs(ref["dmu"]).max(), np.abs(ref["drho"]).max())
    assert np.abs(dmu - ref["dmu"]).max() <= 1e-9 * sc
    assert np.abs(drho - ref["drho"]).max() <= 1e-9 * sc
    bm._draw_eps = lambda n: torch.as_tensor(eps, device=bm.device)
    lp, lq, nll = bm.sample_elbo(xs, ys, S, likparams=[datanoise])
    assert abs(lp.item() - ref["log_prior"]) <= 1e-12 * abs(ref["log_prior"])
    assert abs(lq.item() - ref["log_q"]) <= 1e-12 * abs(ref["log_q"])
    assert abs(nll.item() - ref["nll"]) <= 1e-11 * abs(ref["nll"])

    # ---- all 8192 rows: the data term of loss and gradient is additive over row blocks; the KL part does not depend
    # on the data.  NLL(B rows) = B log(sigma) + B/2 log(2 pi) + SSE / (2 S o sigma^2)
    full_loss, full_dmu, full_drho = device_loss(x, y)
    kl = (ref["log_q"] - ref["log_prior"]) / nb
    const = lambda B: B * np.log(datanoise) + 0.5 * B * np.log(2 * np.pi)
    data_sum, dmu_sum = 0.0, np.zeros(p)
    blocks = np.array_split(np.arange(len(x)), 4)
    kl_dmu = None
    for blk in blocks:
        l, dm, _ = device_loss(x[blk], y[blk])
        data_sum += l - kl - const(len(blk))
        dmu_sum += dm
    # every block's gradient carries the (data-independent) KL gradient once: remove the extra copies
    _, dm_a, _ = device_loss(x[blocks[0]], y[blocks[0]])
    _, dm_b, _ = device_loss(np.concatenate([x[blocks[0]], x[blocks[0]]]), np.concatenate([y[blocks[0]], y[blocks[0]]]))
    kl_dmu = 2 * dm_a - dm_b                                            # data part doubles, KL part does not
    assert abs((full_loss - kl - const(len(x))) - data_sum) <= 1e-10 * abs(data_sum)
    np.testing.assert_allclose(full_dmu, dmu_sum - 3 * kl_dmu, rtol=0, atol=1e-9 * np.abs(full_dmu).max())
    assert np.isfinite(full_drho).all()


def test_cfg5_one_device_hmc_proposal_equals_the_reference_leapfrog_on_the_same_momenta():
    rs = np.random.RandomState(0)
    N, C, L_, eps, sigma = 32768, 8, 10, 2e-6, 0.02
    x = rs.rand(N, 1) * 2 * np.pi - np.pi
    y = np.sin(x) + sigma * rs.randn(N, 1)
    arch = MLPArch((1, 256, 256, 256, 256, 1), "tanh")
    assert arch.nparams == 198145
    op = BatchedMLP(arch, x, y)
    ini = np.stack([0.05 * np.random.RandomState(1000 + c).randn(arch.nparams) for c in range(C)])
    eng = DeviceHMC(op, sigma, epsilon=eps, L=L_, seed=77, chain0=40)
    r = eng.run(1, ini)
    torch.cuda.synchronize()
    # the momenta of step 0 (qn_hmc_begin on the same state / gradient / seed / chain ids), then hmc.py:43-66 on the host
    Lb = _lib.lib()
    cur = torch.as_tensor(ini, device="cuda")
    _, g0 = op.sse_grad(cur)
    p = arch.nparams
    mom, q = torch.empty_like(cur), torch.empty_like(cur)
    kparts = torch.empty(C, Lb.qn_hmc_parts(p), dtype=torch.float64, device="cuda")
    st = torch.zeros(2, dtype=torch.int64, device="cuda")
    _lib.check(Lb.qn_hmc_begin(cur.data_ptr(), g0.data_ptr(), sigma, eps, C, 40, p, eng.seed, st.data_ptr(), mom.data_ptr(),
                               q.data_ptr(), kparts.data_ptr(), None), "qn_hmc_begin")
    torch.cuda.synchronize()
    lpg = lambda Wc: -(0.5 * op.sse_grad(Wc)[1].cpu().numpy() / sigma ** 2)
    lp = lambda Wc: -neg_log_post_from_sse(op.sse(Wc).cpu().numpy(), N, sigma)
    z = mom.cpu().numpy() - eps * lpg(ini) / 2
    k0 = np.array([np.sum(np.square(z[c])) / 2 for c in range(C)])
    np.testing.assert_allclose(0.5 * kparts.cpu().numpy().sum(axis=1), k0, rtol=1e-12)
    qq, pp = ini.copy(), z.copy()
    pp += eps * lpg(qq) / 2
    for jj in range(L_):
        qq += eps * pp
        if jj != L_ - 1:
            pp += eps * lpg(qq)
    pp += eps * lpg(qq) / 2
    k1 = np.array([np.sum(np.square(pp[c])) / 2 for c in range(C)])
    mh = np.exp((-lp(ini) + k0) - (-lp(qq) + k1))
    np.testing.assert_allclose(r['alphas'][:, 1].cpu().numpy(), mh, rtol=1e-6)
    got = r['chain'][:, 1].cpu().numpy()
    moved = (got != ini).any(axis=1)
    assert np.isfinite(mh).all() and (mh > 0).all() and moved.any()      # (a small step: H nearly conserved)
    np.testing.assert_allclose(got[moved], qq[moved], rtol=1e-9, atol=1e-12)
    assert np.array_equal(got[~moved], ini[~moved])
    np.testing.assert_allclose(r['logpost'][:, 0].cpu().numpy(), lp(ini), rtol=1e-12)
