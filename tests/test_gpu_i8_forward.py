"""GPU: the sliced int8-product forward for 64-wide tanh / relu / identity networks (csrc/qn_fused_i8.hip; relu is the
reference's default activation, quinn/nns/mlp.py:23, its activations sliced with one scale per data row) against the oracle, against the
float64-MFMA fused kernel (QN_PATH_FUSED_DP) and against the layer-wise kernels: float64 tolerance 1e-11 (measured
~1e-14), exceptional weights / inputs through its plain-float64 tile path, and an AMCMC chain whose acceptance
indices do not depend on which of the kernels evaluated the log-posterior."""
import numpy as np
import pytest
import torch

from oracle import mlp_ref
from quinn_amd import _lib
from quinn_amd.ops import BatchedMLP, MLPArch, neg_log_post_from_sse

pytestmark = pytest.mark.gpu


def _data(N, d, o, seed=0):
    rs = np.random.RandomState(seed)
    x = rs.rand(N, d) * 2 * np.pi - np.pi
    y = np.sin(x).sum(axis=1, keepdims=True) * np.ones((1, o)) + 0.02 * rs.randn(N, o)
    return x, y


def _both(op, fn):
    out = {}
    for path in (_lib.PATH_FUSED, _lib.PATH_FUSED_DP, _lib.PATH_GENERIC):
        old = op.set_path(path)
        try:
            out[path] = fn()
        finally:
            op.set_path(old)
    return out[_lib.PATH_FUSED], out[_lib.PATH_FUSED_DP], out[_lib.PATH_GENERIC]


@pytest.mark.parametrize("dims,N,B,wscale", [((1, 64, 64, 64, 1), 4096, 64, 0.1), ((1, 64, 64, 1), 300, 5, 1.0),
                                             ((3, 64, 64, 64, 64, 2), 257, 3, 0.3), ((4, 64, 64, 4), 64, 2, 3.0),
                                             ((2, 64, 64, 64, 1), 1000, 9, 1e-3), ((1, 50, 50, 50, 1), 333, 4, 0.2),
                                             ((6, 64, 64, 64, 1), 777, 5, 0.2), ((8, 64, 64, 3), 129, 3, 0.5), ((5, 40, 40, 1), 300, 4, 0.3)],
                         ids=["cfg2", "2hid", "4hid_d3_o2", "d4_o4_bigw", "tinyw", "padded50", "d6", "d8_o3", "d5_padded40"])
@pytest.mark.parametrize("act", ["tanh", "relu", "identity"])
def test_matches_oracle_and_float64_kernels(dims, N, B, wscale, act):
    x, y = _data(N, dims[0], dims[-1])
    arch = MLPArch(dims, act)
    rs = np.random.RandomState(sum(dims) + N)
    if act != "tanh":
        wscale = min(wscale, 0.3)                                       # (unbounded activations: keep the outputs O(1) .. O(1e3))
    W = wscale * rs.randn(B, arch.nparams)
    W[0, : arch.nparams // 3] *= 1e-6                                   # rows with very different scales in one chain
    op = BatchedMLP(arch, x, y)
    assert op.path(B, N, False) == _lib.PATH_FUSED and op.arith(B, N, False) == _lib.ARITH_I8_FUSED
    (s8, p8), (sd, pd), (sg, pg) = _both(op, lambda: tuple(t.cpu().numpy() for t in op.sse_pred(W)))
    np.testing.assert_allclose(s8, sd, rtol=1e-11)
    np.testing.assert_allclose(s8, sg, rtol=1e-11)
    scale = np.abs(pg).max()
    assert np.abs(p8 - pg).max() <= 1e-11 * scale and np.abs(p8 - pd).max() <= 1e-11 * scale
    # parts (what the device samplers consume) sum to the same SSE
    np.testing.assert_allclose(op.sse_parts(W).cpu().numpy().sum(axis=1), s8, rtol=1e-13)
    mod = mlp_ref.build_module(mlp_ref.MLPSpec(dims, act))
    yd = [v for v in y]
    for b in range(min(B, 3)):
        ref = mlp_ref.logpost(mod, W[b], x, yd, 0.02)
        got = -neg_log_post_from_sse(s8[b], N, 0.02)
        assert abs(got - ref) <= 1e-11 * abs(ref), (b, got, ref)
    err8, errd = np.abs(p8 - pg).max() / scale, np.abs(pd - pg).max() / scale
    print(f"max |pred - layerwise| / max|pred|: int8 slices {err8:.2e}, f64 MFMA {errd:.2e}")


@pytest.mark.parametrize("o", [1, 2])
@pytest.mark.parametrize("act", ["tanh", "relu"])
def test_row_subsets_and_ragged_tail(o, act):
    dims = (2, 64, 64, 64, o)
    x, y = _data(777, 2, o, seed=3)
    arch = MLPArch(dims, act)
    rs = np.random.RandomState(5)
    W = 0.2 * rs.randn(6, arch.nparams)
    idx = rs.randint(0, 777, size=(6, 403))
    op = BatchedMLP(arch, x, y)
    (s8, p8), (sd, pd), (sg, pg) = _both(op, lambda: tuple(t.cpu().numpy() for t in op.sse_pred(W, row_idx=idx)))
    np.testing.assert_allclose(s8, sg, rtol=1e-11)
    np.testing.assert_allclose(p8, pg, rtol=0, atol=1e-11 * np.abs(pg).max())
    # determinism: two launches, same bits
    s8b = op.sse(W, row_idx=idx).cpu().numpy()
    assert np.array_equal(s8b, s8)


@pytest.mark.parametrize("where", ["weight_nan", "weight_inf", "weight_huge", "bias_nan", "x_nan", "x_inf", "y_nan", "w0_inf"])
@pytest.mark.parametrize("o,d", [(1, 1), (3, 1), (1, 7)])                 # (o > 1: the 4-output instance of the int8-slice kernel; d = 7: the 8-column one)
@pytest.mark.parametrize("act", ["tanh", "relu", "identity"])
def test_exceptional_values_follow_the_layerwise_kernels(where, o, d, act):
    dims = (d, 64, 64, 64, o)
    arch = MLPArch(dims, act)
    x, y = _data(200, d, o, seed=1)
    rs = np.random.RandomState(2)
    W = 0.2 * rs.randn(3, arch.nparams)
    off_w1 = 64 * d + 64 + 5 * 64 + 7                                    # an entry of the first hidden matrix
    if where == "weight_nan": W[1, off_w1] = np.nan
    if where == "weight_inf": W[1, off_w1] = np.inf
    if where == "weight_huge": W[1, off_w1] = 1e200
    if where == "bias_nan": W[1, 64 * d + 3] = np.nan
    if where == "w0_inf": W[1, 3 * d + d - 1] = -np.inf
    if where == "x_nan": x[17, d - 1] = np.nan
    if where == "x_inf": x[150, d - 1] = -np.inf
    if where == "y_nan": y[17, 0] = np.nan
    op = BatchedMLP(arch, x, y)
    (s8, p8), (sd, pd), (sg, pg) = _both(op, lambda: tuple(t.cpu().numpy() for t in op.sse_pred(W)))
    assert np.array_equal(np.isnan(s8), np.isnan(sg)) and np.array_equal(np.isnan(p8), np.isnan(pg))
    fin = np.isfinite(pg)
    np.testing.assert_allclose(p8[fin], pg[fin], rtol=1e-10, atol=1e-11)
    ok = np.isfinite(sg)
    np.testing.assert_allclose(s8[ok], sg[ok], rtol=1e-11)
    assert np.array_equal(np.isinf(s8), np.isinf(sg))


def test_mh_acceptance_does_not_depend_on_the_kernel():
    """Host HMC chains (the reference-exact stepper, quinn/mcmc/mcmc.py:65-85) on the cfg2 network with the proposal's
    log-posterior from the int8-slice kernel and from the float64-MFMA kernel: same acceptance indices, states equal to
    1e-9.  (The host AMCMC cannot run at p = 8513: numpy's multivariate_normal factors a p x p matrix per draw.)"""
    from quinn_amd.mcmc.hmc import HMC
    dims = (1, 64, 64, 64, 1)
    arch = MLPArch(dims, "tanh")
    x, y = _data(512, 1, 1, seed=4)
    op = BatchedMLP(arch, x, y)
    sigma, C, nmcmc = 0.05, 4, 120
    res = {}
    for path in (_lib.PATH_FUSED, _lib.PATH_FUSED_DP):
        op.set_path(path)
        mc = HMC(epsilon=2e-4, L=3)
        mc.setLogPostBatch(lambda Wc: -neg_log_post_from_sse(op.sse(Wc).cpu().numpy(), 512, sigma),
                           lambda Wc: -(0.5 * op.sse_grad(Wc)[1].cpu().numpy() / sigma ** 2))
        rngs = [np.random.RandomState(40 + c) for c in range(C)]
        ini = np.stack([0.1 * np.random.RandomState(90 + c).randn(arch.nparams) for c in range(C)])
        res[path] = mc.run(nmcmc, ini, rngs=rngs, verbose=False)
    op.set_path(_lib.PATH_AUTO)
    a, b = res[_lib.PATH_FUSED], res[_lib.PATH_FUSED_DP]
    moved = lambda r: (np.asarray(r['chain'])[:, 1:] != np.asarray(r['chain'])[:, :-1]).any(axis=2)
    assert np.array_equal(moved(a), moved(b))
    assert 0.05 < moved(a).mean() < 1.0, moved(a).mean()
    np.testing.assert_allclose(a['chain'], b['chain'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(a['logpost'], b['logpost'], rtol=1e-11)
    fin = np.isfinite(b['alphas']) & (b['alphas'] < 1e300)
    np.testing.assert_allclose(a['alphas'][fin], b['alphas'][fin], rtol=1e-6)


def test_path_override_is_per_operator():
    """qn_mlp_desc_set_path acts on one descriptor: two operators in one process keep their own choice."""
    dims = (1, 64, 64, 64, 1)
    arch = MLPArch(dims, "tanh")
    x, y = _data(128, 1, 1)
    a, b = BatchedMLP(arch, x, y), BatchedMLP(arch, x, y)
    assert a.set_path(_lib.PATH_GENERIC) == _lib.PATH_AUTO
    assert a.path(8, 128, False) == _lib.PATH_GENERIC and b.path(8, 128, False) == _lib.PATH_FUSED
    assert b.set_path(_lib.PATH_FUSED_DP) == _lib.PATH_AUTO and a.set_path(_lib.PATH_AUTO) == _lib.PATH_GENERIC
    W = 0.1 * np.random.RandomState(0).randn(8, arch.nparams)
    np.testing.assert_allclose(a.sse(W).cpu().numpy(), b.sse(W).cpu().numpy(), rtol=1e-11)


@pytest.mark.parametrize("dims,N,B", [((2, 128, 128, 128, 1), 700, 5), ((1, 256, 256, 256, 256, 1), 300, 3), ((3, 128, 256, 128, 2), 257, 2)],
                         ids=["3x128", "4x256", "mixed_o2"])
def test_layerwise_int8_forward_of_wide_networks(dims, N, B):
    """Hidden widths 128 / 256: the forward part of a GRADIENT evaluation (and, at width 256, of a forward call) runs layer
    by layer as sliced int8 products (k_i8_slice_w / k_i8_first / k_i8_gemm); QN_PATH_GENERIC is the exact float64
    layer-wise reference.  SSE, predictions and gradients (the backward pass consumes the stored float64 activations)."""
    x, y = _data(N, dims[0], dims[-1], seed=7)
    arch = MLPArch(dims, "tanh")
    rs = np.random.RandomState(sum(dims))
    W = rs.randn(B, arch.nparams) / np.sqrt(max(dims))
    idx = rs.randint(0, N, size=(B, N // 2 + 5))
    op = BatchedMLP(arch, x, y)
    res = {}
    for path in (_lib.PATH_AUTO, _lib.PATH_GENERIC):
        op.set_path(path)
        s, g = op.sse_grad(W)
        s2, pr = op.sse_pred(W, row_idx=idx)
        res[path] = (s.cpu().numpy(), g.cpu().numpy(), s2.cpu().numpy(), pr.cpu().numpy())
    op.set_path(_lib.PATH_AUTO)
    a, r = res[_lib.PATH_AUTO], res[_lib.PATH_GENERIC]
    np.testing.assert_allclose(a[0], r[0], rtol=1e-11)
    np.testing.assert_allclose(a[2], r[2], rtol=1e-11)
    assert np.abs(a[1] - r[1]).max() <= 1e-10 * np.abs(r[1]).max()
    assert np.abs(a[3] - r[3]).max() <= 1e-11 * np.abs(r[3]).max()
    mod = mlp_ref.build_module(mlp_ref.MLPSpec(dims, "tanh"))
    ref = mlp_ref.logpost(mod, W[0], x, [v for v in y], 0.1)
    got = -neg_log_post_from_sse(a[0][0], N * 1, 0.1) if dims[-1] == 1 else None
    if got is not None:
        assert abs(got - ref) <= 1e-11 * abs(ref)


@pytest.mark.parametrize("where", ["weight_nan", "weight_inf", "weight_huge", "bias_nan", "x_nan", "x_inf", "w0_inf"])
def test_layerwise_int8_forward_exceptional_values(where):
    dims = (1, 128, 128, 1)
    arch = MLPArch(dims, "tanh")
    x, y = _data(150, 1, 1, seed=1)
    rs = np.random.RandomState(2)
    W = 0.2 * rs.randn(3, arch.nparams)
    off_w1 = 128 + 128 + 5 * 128 + 7                                     # an entry of the hidden matrix
    if where == "weight_nan": W[1, off_w1] = np.nan
    if where == "weight_inf": W[1, off_w1] = np.inf
    if where == "weight_huge": W[1, off_w1] = 1e200
    if where == "bias_nan": W[1, 128 + 128 + 128 * 128 + 3] = np.nan
    if where == "w0_inf": W[1, 3] = -np.inf
    if where == "x_nan": x[17, 0] = np.nan
    if where == "x_inf": x[140, 0] = -np.inf
    op = BatchedMLP(arch, x, y)
    res = {}
    for path in (_lib.PATH_AUTO, _lib.PATH_GENERIC):
        op.set_path(path)
        s, g = op.sse_grad(W)
        res[path] = (s.cpu().numpy(), g.cpu().numpy())
    op.set_path(_lib.PATH_AUTO)
    (sa, ga), (sg, gg) = res[_lib.PATH_AUTO], res[_lib.PATH_GENERIC]
    assert np.array_equal(np.isnan(sa), np.isnan(sg)) and np.array_equal(np.isinf(sa), np.isinf(sg))
    ok = np.isfinite(sg)
    np.testing.assert_allclose(sa[ok], sg[ok], rtol=1e-11)
    assert np.array_equal(np.isnan(ga), np.isnan(gg))
    fin = np.isfinite(gg)
    if fin.any():                                                        # (a NaN input makes every gradient NaN)
        assert np.abs(ga[fin] - gg[fin]).max() <= 1e-9 * max(np.abs(gg[fin]).max(), 1e-300)
