"""GPU: the fused int8-slice forward for 128 / 256-wide networks (csrc/qn_wide_i8.hip: k_i8_wide_fwd for tanh, k_i8_wide_fwd_u with
per-row activation scales for relu -- the reference's default, quinn/nns/mlp.py:23 -- and identity) against the
oracle and against the exact float64 layer-wise kernels (QN_PATH_GENERIC): SSE, predictions and -- through the float64
activations it stashes for the backward pass -- gradients; ragged row counts, per-member row subsets, 1..4 inputs, no
bias, 2..5 hidden layers, padded twins, exceptional weights / inputs (plain-float64 rows), determinism.
Float64 tolerances: 1e-11 on SSE / predictions, 1e-10 of max |g| on gradients."""
import numpy as np
import pytest

from oracle import mlp_ref
from quinn_amd import _lib
from quinn_amd.ops import BatchedMLP, MLPArch, neg_log_post_from_sse

pytestmark = pytest.mark.gpu


def _data(N, d, seed=0):
    rs = np.random.RandomState(seed)
    x = rs.rand(N, d) * 2 * np.pi - np.pi
    y = np.sin(x).sum(axis=1, keepdims=True) + 0.02 * rs.randn(N, 1)
    return x, y


def _run(op, W, idx=None):
    out = {}
    for path in (_lib.PATH_AUTO, _lib.PATH_GENERIC):
        old = op.set_path(path)
        try:
            s, g = op.sse_grad(W, row_idx=idx)
            s2, pr = op.sse_pred(W, row_idx=idx)
            out[path] = tuple(t.cpu().numpy() for t in (s, g, s2, pr))
        finally:
            op.set_path(old)
    return out[_lib.PATH_AUTO], out[_lib.PATH_GENERIC]


def _check(a, r):
    np.testing.assert_allclose(a[0], r[0], rtol=1e-11)
    np.testing.assert_allclose(a[2], r[2], rtol=1e-11)
    assert np.abs(a[1] - r[1]).max() <= 1e-10 * np.abs(r[1]).max()
    assert np.abs(a[3] - r[3]).max() <= 1e-11 * np.abs(r[3]).max()


ACTS = ["tanh", "relu", "identity"]

CASES = [((2, 128, 128, 128, 1), 8192, 6, True), ((1, 256, 256, 256, 256, 1), 1024, 5, True),
         ((1, 128, 128, 1), 63, 3, True), ((3, 128, 128, 128, 128, 128, 1), 130, 2, True),
         ((4, 256, 256, 1), 321, 4, True), ((1, 256, 256, 256, 1), 200, 3, False), ((2, 128, 128, 128, 1), 77, 9, False),
         ((1, 256, 256, 256, 256, 256, 256, 1), 65, 2, True),
         # whole 64-row chunks (the group-scale weight-gradient kernel k_i8_dw_g): a partial last group, a single chunk,
         # slabs of several groups
         ((2, 128, 128, 128, 1), 448, 5, True), ((1, 256, 256, 256, 1), 64, 3, True), ((3, 256, 256, 256, 1), 2368, 2, False),
         # 5..8 inputs (round 4: a third width of the first layer's LDS image and input registers)
         ((6, 128, 128, 128, 1), 300, 3, True), ((8, 256, 256, 1), 130, 2, True), ((5, 128, 128, 1), 77, 4, False), ((7, 256, 256, 256, 1), 1000, 2, True)]


@pytest.mark.parametrize("act", ACTS)
@pytest.mark.parametrize("dims,N,B,bias", CASES, ids=[f"{c[0][1]}x{len(c[0]) - 2}_d{c[0][0]}_N{c[1]}{'' if c[3] else '_nobias'}" for c in CASES])
def test_wide_forward_matches_layerwise_float64_and_oracle(dims, N, B, bias, act):
    x, y = _data(N, dims[0], seed=len(dims))
    arch = MLPArch(dims, act, bias=bias)
    rs = np.random.RandomState(N + B)
    W = rs.randn(B, arch.nparams) / np.sqrt(dims[1])
    W[0, : arch.nparams // 2] *= 1e-5                        # rows with very different scales in one vector
    W[-1] *= 8.0                                             # saturated activations
    op = BatchedMLP(arch, x, y)
    assert op.arith(B, N, False) == op.arith(B, N, True) == _lib.ARITH_I8_WIDE
    a, r = _run(op, W)
    _check(a, r)
    if bias:
        mod = mlp_ref.build_module(mlp_ref.MLPSpec(dims, act))
        for b in range(min(B, 2)):
            ref = mlp_ref.logpost(mod, W[b], x, [v for v in y], 0.05)
            got = -neg_log_post_from_sse(a[0][b], N, 0.05)
            assert abs(got - ref) <= 1e-11 * abs(ref)
            gref = mlp_ref.logpostgrad(mod, W[b], x, [v for v in y], 0.05)
            ggot = -(0.5 * a[1][b] / 0.05 ** 2)
            assert np.abs(ggot - gref).max() <= 1e-9 * np.abs(gref).max()


@pytest.mark.parametrize("act", ACTS)
@pytest.mark.parametrize("dims", [(2, 128, 128, 128, 1), (1, 256, 256, 256, 1)])
def test_row_subsets_ragged_tail_and_determinism(dims, act):
    N, B = 1000, 7
    x, y = _data(N, dims[0], seed=3)
    arch = MLPArch(dims, act)
    rs = np.random.RandomState(5)
    W = 0.1 * rs.randn(B, arch.nparams)
    idx = rs.randint(0, N, size=(B, 333))                    # 333 rows per member: 5 full 64-row iterations + 13 rows
    op = BatchedMLP(arch, x, y)
    a, r = _run(op, W, idx)
    _check(a, r)
    a2, _ = _run(op, W, idx)
    for u, v in zip(a, a2):
        assert np.array_equal(u, v)                          # bitwise reproducible
    # additivity over disjoint row sets
    full = op.sse(W).cpu().numpy()
    h1 = op.sse(W, row_idx=np.tile(np.arange(0, 500, dtype=np.int32), (B, 1))).cpu().numpy()
    h2 = op.sse(W, row_idx=np.tile(np.arange(500, N, dtype=np.int32), (B, 1))).cpu().numpy()
    np.testing.assert_allclose(h1 + h2, full, rtol=1e-12)


@pytest.mark.parametrize("act", ACTS)
@pytest.mark.parametrize("h", [128, 256])
@pytest.mark.parametrize("N,B", [(1, 1), (16, 1), (64, 3), (65, 2), (129, 300)])
def test_tiny_row_counts_and_many_or_single_vectors(h, N, B, act):
    """One row, one vector, one more row than an iteration takes, more vectors than workgroups per chip."""
    dims = (1, h, h, 1)
    x, y = _data(N, 1, seed=N)
    arch = MLPArch(dims, act)
    W = 0.3 * np.random.RandomState(B).randn(B, arch.nparams)
    op = BatchedMLP(arch, x, y)
    a, r = _run(op, W)
    _check(a, r)


@pytest.mark.parametrize("act", ACTS)
def test_padded_twin_takes_the_wide_kernel(act):
    dims = (1, 100, 100, 100, 1)                             # runs on its 128-wide zero-padded twin
    x, y = _data(500, 1, seed=9)
    arch = MLPArch(dims, act)
    W = 0.2 * np.random.RandomState(1).randn(4, arch.nparams)
    op = BatchedMLP(arch, x, y)
    a, r = _run(op, W)
    _check(a, r)


@pytest.mark.parametrize("act", ACTS)
@pytest.mark.parametrize("where", ["weight_nan", "weight_inf", "weight_huge", "bias_nan", "bias_big", "x_nan", "x_inf", "x_big", "w0_inf", "wl_nan"])
@pytest.mark.parametrize("h,N", [(128, 150), (256, 150), (128, 192), (256, 320)])       # (whole 64-row chunks: k_i8_dw_g)
def test_exceptional_values_follow_the_layerwise_kernels(where, h, N, act):
    dims = (1, h, h, h, 1)
    arch = MLPArch(dims, act)
    x, y = _data(N, 1, seed=1)
    rs = np.random.RandomState(2)
    W = 0.2 * rs.randn(3, arch.nparams)
    off_w1 = h + h + 5 * h + 7                               # an entry of the first hidden matrix
    if where == "weight_nan": W[1, off_w1] = np.nan
    if where == "weight_inf": W[1, off_w1] = np.inf
    if where == "weight_huge": W[1, off_w1] = 1e200
    if where == "bias_nan": W[1, h + h + h * h + 3] = np.nan
    if where == "bias_big": W[1, h + h + h * h + 3] = 3e7       # (>= 2^20: outside the contract of the relu / identity kernels)
    if where == "x_big": x[33, 0] = 1e40                         # (>= 2^100)
    if where == "w0_inf": W[1, 3] = -np.inf
    if where == "wl_nan": W[1, arch.nparams - 5] = np.nan
    if where == "x_nan": x[17, 0] = np.nan
    if where == "x_inf": x[140, 0] = -np.inf
    op = BatchedMLP(arch, x, y)
    res = {}
    for path in (_lib.PATH_AUTO, _lib.PATH_GENERIC):
        op.set_path(path)
        s, g = op.sse_grad(W)
        s2, pr = op.sse_pred(W)
        res[path] = tuple(t.cpu().numpy() for t in (s, g, s2, pr))
    op.set_path(_lib.PATH_AUTO)
    (sa, ga, sa2, pa), (sg, gg, sg2, pg) = res[_lib.PATH_AUTO], res[_lib.PATH_GENERIC]
    for u, v in ((sa, sg), (sa2, sg2), (pa, pg)):
        assert np.array_equal(np.isnan(u), np.isnan(v)) and np.array_equal(np.isinf(u), np.isinf(v))
        ok = np.isfinite(v)
        if ok.any():
            np.testing.assert_allclose(u[ok], v[ok], rtol=1e-11, atol=1e-11 * np.abs(v[ok]).max())
    assert np.array_equal(np.isnan(ga), np.isnan(gg))
    fin = np.isfinite(gg)
    if fin.any():
        assert np.abs(ga[fin] - gg[fin]).max() <= 1e-9 * max(np.abs(gg[fin]).max(), 1e-300)


def test_forward_only_calls_at_width_128_leave_the_streaming_kernel():
    arch = MLPArch((2, 128, 128, 128, 1), "tanh")
    x, y = _data(256, 2)
    op = BatchedMLP(arch, x, y)
    assert op.path(8, 256, False) == _lib.PATH_GENERIC       # layer-wise family = the fused int8-slice forward
    op.set_path(_lib.PATH_FUSED)
    assert op.path(8, 256, False) == _lib.PATH_FUSED         # the float64-MFMA streaming kernel stays selectable
    W = 0.1 * np.random.RandomState(0).randn(8, arch.nparams)
    s_f = op.sse(W).cpu().numpy()
    op.set_path(_lib.PATH_AUTO)
    np.testing.assert_allclose(op.sse(W).cpu().numpy(), s_f, rtol=1e-11)


TINY = [((2, 64, 64, 64, 64, 1), 700, 3, False), ((1, 64, 64, 64, 1), 130, 2, True),           # fused 64-wide forward
        ((2, 128, 128, 128, 128, 128, 1), 1754, 1, False), ((3, 256, 256, 256, 1), 300, 4, False),   # wide forward / backward / dW
        ((3, 128, 128, 128, 1), 200, 3, True), ((6, 128, 128, 128, 128, 1), 257, 2, False),    # d = 6: layer-wise int8 forward
        ((2, 100, 100, 100, 1), 150, 2, False),                                                # padded twin
        ((2, 128, 128, 128, 1), 512, 2, False), ((1, 256, 256, 256, 1), 320, 3, False)]        # whole chunks: group-scale dW


@pytest.mark.parametrize("dims,N,B,bias", TINY, ids=[f"{c[0][1]}x{len(c[0]) - 2}_d{c[0][0]}_N{c[1]}{'' if c[3] else '_nobias'}" for c in TINY])
@pytest.mark.parametrize("act", ACTS)
@pytest.mark.parametrize("wscale", [1e-3, 3e-2])
def test_tiny_activations_keep_relative_accuracy(dims, N, B, bias, wscale, act):
    """Weights ~ 1e-3 (and no bias): activations shrink layer by layer (1e-2, 1e-4, ... 1e-10).  The int8-slice kernels
    slice activations with a fixed scale (absolute error 2^-47); rows whose activations are all below 2^-7 in some layer
    must take the plain-float64 path so predictions and gradients keep their RELATIVE accuracy (found by
    tools/fuzz_wide.py: 2.7e-6 on predictions, 4.7e-8 on gradients before the guard)."""
    x, y = _data(N, dims[0], seed=11)
    arch = MLPArch(dims, act, bias=bias)
    W = wscale * np.random.RandomState(N).randn(B, arch.nparams)
    W[-1, : arch.nparams // 3] *= 30.0                       # one vector whose first layers are of ordinary size
    op = BatchedMLP(arch, x, y)
    a, r = _run(op, W)
    _check(a, r)
    # the bar per vector (the one with larger weights must not hide the others)
    for b in range(B):
        assert np.abs(a[3][b] - r[3][b]).max() <= 1e-11 * np.abs(r[3][b]).max()
        assert np.abs(a[1][b] - r[1][b]).max() <= 1e-10 * np.abs(r[1][b]).max()
