"""GPU: randomized cross-check of the two kernel families (fused MFMA vs layer-wise) over shapes the
fused family supports: hidden width 16/32/64, 1-4 hidden layers, d, o in 1..4, every activation, with
and without bias, ragged row counts, per-member row subsets; plus hidden widths that the library zero-pads to
those (any width <= 64, non-uniform).  float64; SSE 1e-12, gradient 1e-10."""
import numpy as np
import pytest
import torch

from quinn_amd import _lib
from quinn_amd.ops import MLPArch, BatchedMLP

pytestmark = pytest.mark.gpu


def _cases(n=36, seed=123):
    rs = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        H = int(rs.choice([16, 32, 64]))
        NH = int(rs.randint(1, 5))
        if H == 64 and NH > 3:
            continue
        d, o = int(rs.randint(1, 5)), int(rs.randint(1, 5))
        act = str(rs.choice(["tanh", "relu", "identity"]))
        out.append(((d,) + (H,) * NH + (o,), act, bool(rs.rand() < 0.8), int(rs.choice([1, 7, 64, 65, 200, 513])),
                    int(rs.randint(1, 7)), bool(rs.rand() < 0.4)))
    return out


def _ragged_cases():
    """Hidden widths that are not one common 16 / 32 / 64: the library zero-pads them for the fused kernels."""
    shapes = [(1, 11, 11, 11, 1), (2, 50, 3), (1, 20, 40, 10, 1), (3, 64, 32, 2), (1, 3, 1), (4, 33, 17, 64, 4),
              (1, 64, 64, 63, 1), (2, 1, 1, 2)]
    rs = np.random.RandomState(7)
    out = []
    for k, dims in enumerate(shapes):
        act = ["tanh", "relu", "identity"][k % 3]
        out.append((dims, act, k % 4 != 3, int(rs.choice([1, 37, 130, 400])), int(rs.randint(1, 6)), k % 2 == 1))
    return out


@pytest.mark.parametrize("case", _cases() + _ragged_cases(),
                         ids=lambda c: f"{c[0]}-{c[1]}-b{int(c[2])}-N{c[3]}-B{c[4]}-idx{int(c[5])}")
def test_fused_equals_generic(case):
    dims, act, bias, N, B, use_idx = case
    rs = np.random.RandomState(sum(dims) * 1000 + N * 7 + B)           # deterministic per case
    arch = MLPArch(dims, act, bias)
    x, y = rs.randn(N, dims[0]), rs.randn(N, dims[-1])
    W = rs.randn(B, arch.nparams) / np.sqrt(max(dims))
    idx = rs.randint(0, N, size=(B, max(1, N // 2 + 3))) if use_idx else None
    op = BatchedMLP(arch, x, y)
    L = _lib.lib()
    res = {}
    for path in (_lib.PATH_GENERIC, _lib.PATH_FUSED):
        old = op.set_path(path)
        try:
            s, g = op.sse_grad(W, row_idx=idx)
            s2, pr = op.sse_pred(W, row_idx=idx)
        finally:
            op.set_path(old)
        res[path] = (s.cpu().numpy(), g.cpu().numpy(), s2.cpu().numpy(), pr.cpu().numpy())
    a, b = res[_lib.PATH_GENERIC], res[_lib.PATH_FUSED]
    np.testing.assert_allclose(b[0], a[0], rtol=1e-12)
    np.testing.assert_allclose(b[2], a[2], rtol=1e-12)
    np.testing.assert_allclose(b[2], b[0], rtol=1e-12)
    assert np.abs(b[1] - a[1]).max() <= 1e-10 * max(np.abs(a[1]).max(), 1e-300)
    np.testing.assert_allclose(b[3], a[3], rtol=1e-11, atol=1e-12)


def _stream_cases():
    """Hidden width 128: forward-only streaming kernel (one 128 KB matrix buffer in LDS, restaged per layer)."""
    out = []
    rs = np.random.RandomState(5)
    for k, (NH, d, o) in enumerate([(1, 1, 1), (2, 2, 1), (3, 2, 1), (4, 4, 4), (3, 3, 2), (6, 1, 3)]):
        act = ["tanh", "relu", "identity"][k % 3]
        out.append(((d,) + (128,) * NH + (o,), act, k != 3, int(rs.choice([1, 127, 129, 700])), int(rs.randint(1, 5)),
                    k % 2 == 1))
    return out


@pytest.mark.parametrize("case", _stream_cases(),
                         ids=lambda c: f"{c[0]}-{c[1]}-b{int(c[2])}-N{c[3]}-B{c[4]}-idx{int(c[5])}")
def test_streaming_forward_equals_generic(case):
    dims, act, bias, N, B, use_idx = case
    rs = np.random.RandomState(sum(dims) + N * 7 + B)
    arch = MLPArch(dims, act, bias)
    x, y = rs.randn(N, dims[0]), rs.randn(N, dims[-1])
    W = rs.randn(B, arch.nparams) / np.sqrt(128)
    idx = rs.randint(0, N, size=(B, max(1, N // 2 + 3))) if use_idx else None
    op = BatchedMLP(arch, x, y)
    L = _lib.lib()
    Nb = N if idx is None else idx.shape[1]
    # (uniform 128-wide networks with one output, <= 4 inputs and >= 2 hidden layers take the int8-slice kernels of the layer-wise
    # family under PATH_AUTO -- tanh since round 2, relu / identity since round 4; the streaming kernel stays selectable)
    wide = dims[-1] == 1 and dims[0] <= 4 and len(dims) >= 4
    assert op.path(B, Nb, False) == (_lib.PATH_GENERIC if wide else _lib.PATH_FUSED)
    assert op.arith(B, Nb, False) == (_lib.ARITH_I8_WIDE if wide else _lib.ARITH_PLAIN)
    assert op.path(B, Nb, True) == _lib.PATH_GENERIC
    res = {}
    for path in (_lib.PATH_GENERIC, _lib.PATH_FUSED):
        old = op.set_path(path)
        try:
            s = op.sse(W, row_idx=idx)
            s2, pr = op.sse_pred(W, row_idx=idx)
        finally:
            op.set_path(old)
        res[path] = (s.cpu().numpy(), s2.cpu().numpy(), pr.cpu().numpy())
    a, b = res[_lib.PATH_GENERIC], res[_lib.PATH_FUSED]
    np.testing.assert_allclose(b[0], a[0], rtol=1e-12)
    np.testing.assert_allclose(b[1], a[1], rtol=1e-12)
    np.testing.assert_allclose(b[2], a[2], rtol=1e-11, atol=1e-12)
    # non-finite weights in a streamed matrix switch the block to the NaN-propagating tanh
    if act == "tanh" and len(dims) > 3:
        W2 = W.copy()
        W2[0, arch.nparams // 2] = np.inf
        old = op.set_path(_lib.PATH_GENERIC)
        try:
            ref = op.sse(W2, row_idx=idx).cpu().numpy()
        finally:
            op.set_path(old)
        got = op.sse(W2, row_idx=idx).cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=1e-12, equal_nan=True)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("dims", [(2, 100, 100, 1), (1, 70, 128, 90, 1), (3, 200, 300, 2), (1, 65, 1), (4, 129, 64, 4),
                                  (1, 100, 100, 100, 5), (6, 50, 50, 7), (1, 40, 64, 33, 1), (2, 64, 64, 64, 64, 2)])
def test_wide_ragged_widths_run_on_the_padded_twin(dims, dtype):
    """Hidden widths above 64 that are no multiples of 64: the default path pads them (to 128, or to multiples of 64)
    and runs the MFMA kernels on the twin; the forced layer-wise path keeps the exact widths (VALU kernels)."""
    rs = np.random.RandomState(sum(dims))
    arch = MLPArch(dims, "tanh")
    N, B = 333, 3
    x, y = rs.randn(N, dims[0]), rs.randn(N, dims[-1])
    W = rs.randn(B, arch.nparams) / np.sqrt(max(dims))
    idx = rs.randint(0, N, size=(B, 150))
    op = BatchedMLP(arch, x, y, dtype=dtype)
    L = _lib.lib()
    res = {}
    for path in (_lib.PATH_GENERIC, _lib.PATH_AUTO):
        old = op.set_path(path)
        try:
            s, g = op.sse_grad(W)
            s2, pr = op.sse_pred(W, row_idx=idx)
        finally:
            op.set_path(old)
        res[path] = (s.cpu().numpy(), g.double().cpu().numpy(), s2.cpu().numpy(), pr.double().cpu().numpy())
    a, b = res[_lib.PATH_GENERIC], res[_lib.PATH_AUTO]
    rt, gt = (1e-11, 1e-10) if dtype == "float64" else (2e-4, 2e-3)
    np.testing.assert_allclose(b[0], a[0], rtol=rt)
    np.testing.assert_allclose(b[2], a[2], rtol=rt)
    assert np.abs(b[1] - a[1]).max() <= gt * np.abs(a[1]).max()
    np.testing.assert_allclose(b[3], a[3], rtol=rt * 10, atol=rt * 10)
    assert b[1].shape == (B, arch.nparams)


@pytest.mark.parametrize("where", ["weight_nan", "weight_inf", "weight_huge", "bias_nan", "x_nan", "x_inf", "y_nan"])
@pytest.mark.parametrize("grad", [False, True])
def test_non_finite_inputs_follow_the_reference(where, grad):
    """The fused kernels switch to the NaN-propagating tanh when a chain's weights or a tile's inputs are
    not bounded: results equal the layer-wise kernels' and the oracle's (NaN where torch gives NaN)."""
    from oracle import mlp_ref
    dims, N, B = (1, 64, 64, 64, 1), 200, 3
    rs = np.random.RandomState(11)
    x = rs.uniform(-3, 3, (N, 1)); y = np.sin(x) + 0.1 * rs.randn(N, 1)
    arch = MLPArch(dims, "tanh")
    W = 0.3 * rs.randn(B, arch.nparams)
    off_w1 = 64 + 64 + 5 * 64 + 7           # an entry of the first hidden->hidden matrix
    if where == "weight_nan": W[1, off_w1] = np.nan
    if where == "weight_inf": W[1, off_w1] = np.inf
    if where == "weight_huge": W[1, off_w1] = 1e200
    if where == "bias_nan": W[1, 64 + 3] = np.nan
    if where == "x_nan": x[17, 0] = np.nan
    if where == "x_inf": x[17, 0] = -np.inf
    if where == "y_nan": y[17, 0] = np.nan
    op = BatchedMLP(arch, x, y)
    out = {}
    for path in (_lib.PATH_GENERIC, _lib.PATH_FUSED):
        old = op.set_path(path)
        try:
            if grad:
                s, g = op.sse_grad(W)
                out[path] = (s.cpu().numpy(), g.cpu().numpy())
            else:
                s, pr = op.sse_pred(W)
                out[path] = (s.cpu().numpy(), pr.cpu().numpy())
        finally:
            op.set_path(old)
    sg, ag = out[_lib.PATH_GENERIC]
    sf, af = out[_lib.PATH_FUSED]
    assert np.array_equal(np.isnan(sg), np.isnan(sf))
    np.testing.assert_allclose(sf, sg, rtol=1e-11)
    assert np.array_equal(np.isnan(ag), np.isnan(af))
    fin = np.isfinite(ag)
    if fin.any():
        np.testing.assert_allclose(af[fin], ag[fin], rtol=1e-8, atol=1e-9 * max(1.0, np.abs(ag[fin]).max()))
    mod = mlp_ref.build_module(mlp_ref.MLPSpec(dims, "tanh"))
    for b in range(B):
        ref = mlp_ref.sse(mod, W[b], x, y)
        assert np.isnan(ref) == np.isnan(sf[b])
        if np.isfinite(ref):
            np.testing.assert_allclose(sf[b], ref, rtol=1e-11)


@pytest.mark.parametrize("dims,N,B", [((1, 64, 64, 64, 1), 4096, 64), ((2, 32, 32, 1), 700, 5), ((1, 50, 50, 1), 300, 3),
                                      ((1, 128, 128, 128, 1), 1000, 4), ((3, 256, 256, 2), 200, 2)])
def test_partial_sums_add_up_to_the_sse_bit_for_bit(dims, N, B):
    """qn_mlp_sse_fwd_parts: the forward without its final summation launch; summing a row left to right gives exactly
    what qn_mlp_sse_fwd returns (the accept kernel of the device sampler does that sum itself)."""
    rs = np.random.RandomState(N)
    arch = MLPArch(dims, "tanh")
    x, y = rs.randn(N, dims[0]), rs.randn(N, dims[-1])
    op = BatchedMLP(arch, x, y)
    W = op.weights(rs.randn(B, arch.nparams) / np.sqrt(max(dims)))
    parts = op.sse_parts(W)
    full = op.sse(W)
    assert parts.shape[0] == B and parts.shape[1] >= 1
    acc = torch.zeros(B, dtype=torch.float64, device=parts.device)
    for i in range(parts.shape[1]):
        acc = acc + parts[:, i]
    assert torch.equal(acc, full)
    if dims == (1, 64, 64, 64, 1):
        assert parts.shape[1] == 8                       # cfg2: 8 row splits per chain


@pytest.mark.parametrize("act", ["tanh", "relu", "identity"])
@pytest.mark.parametrize("dims", [(6, 64, 64, 64, 1), (10, 32, 32, 10), (16, 16, 16), (3, 64, 64, 7), (5, 50, 40, 2),
                                  (8, 11, 11, 11, 12), (1, 32, 32, 9)])
def test_wide_first_and_last_layer_in_the_fused_forward(dims, act):
    """Up to 16 inputs / outputs (every activation since round 4): the fused FORWARD kernel takes them (inputs / targets beyond 4 are read where
    they are used instead of being prefetched); the gradient runs on the fused kernel up to 16 inputs and
    16 outputs as well (round 4), beyond that on the layer-wise kernels."""
    rs = np.random.RandomState(sum(dims))
    arch = MLPArch(dims, act)
    N, B = 777, 4
    x, y = rs.randn(N, dims[0]), rs.randn(N, dims[-1])
    W = rs.randn(B, arch.nparams) / np.sqrt(max(dims))
    idx = rs.randint(0, N, size=(B, 300))
    op = BatchedMLP(arch, x, y)
    grad_fused = True                                # (up to 16 inputs and 16 outputs since round 4)
    assert op.path(B, N, False) == _lib.PATH_FUSED and op.path(B, N, True) == (_lib.PATH_FUSED if grad_fused else _lib.PATH_GENERIC)
    L = _lib.lib()
    res = {}
    for path in (_lib.PATH_GENERIC, _lib.PATH_AUTO):
        old = op.set_path(path)
        try:
            s1 = op.sse(W)
            s2, pr = op.sse_pred(W, row_idx=idx)
        finally:
            op.set_path(old)
        res[path] = (s1.cpu().numpy(), s2.cpu().numpy(), pr.cpu().numpy())
    a, b = res[_lib.PATH_GENERIC], res[_lib.PATH_AUTO]
    np.testing.assert_allclose(b[0], a[0], rtol=1e-12)
    np.testing.assert_allclose(b[1], a[1], rtol=1e-12)
    np.testing.assert_allclose(b[2], a[2], rtol=1e-11, atol=1e-12)
    # non-finite input in a late column of x: NaN-propagating tanh, same result as the layer-wise path
    x2 = x.copy(); x2[5, dims[0] - 1] = np.nan
    op2 = BatchedMLP(arch, x2, y)
    old = op2.set_path(_lib.PATH_GENERIC)
    try:
        ref = op2.sse(W).cpu().numpy()
    finally:
        op2.set_path(old)
    np.testing.assert_allclose(op2.sse(W).cpu().numpy(), ref, rtol=1e-12, equal_nan=True)


def _d8_cases():
    """5..8 inputs: the gradient kernel's DP = 8 instances (every width / depth the kernel has, tanh and the unbounded activations)
    and the relu / identity forward's."""
    rs = np.random.RandomState(808)
    out = []
    for H, NH in [(16, 1), (16, 2), (16, 3), (16, 4), (32, 1), (32, 2), (32, 3), (32, 4), (64, 1), (64, 2), (64, 3), (50, 2), (11, 3)]:
        for act in ("tanh", "relu", "identity"):
            d, o = int(rs.randint(5, 9)), int(rs.randint(1, 5))
            out.append(((d,) + (H,) * NH + (o,), act, bool(rs.rand() < 0.8), int(rs.choice([1, 63, 64, 65, 200, 513])),
                        int(rs.randint(1, 7)), bool(rs.rand() < 0.4)))
    # 9..16 inputs: the gradient kernel's DP = 16 instances (hidden widths 16 / 32 / 64 and their zero-padded twins)
    for H, NH in [(16, 1), (16, 2), (16, 3), (16, 4), (32, 1), (32, 2), (32, 3), (32, 4), (11, 3), (20, 2), (64, 1), (64, 2), (64, 3), (50, 2)]:
        for act in ("tanh", "relu", "identity"):
            d, o = int(rs.randint(9, 17)), int(rs.randint(1, 5))
            out.append(((d,) + (H,) * NH + (o,), act, bool(rs.rand() < 0.8), int(rs.choice([1, 63, 64, 65, 200, 513])),
                        int(rs.randint(1, 7)), bool(rs.rand() < 0.4)))
    # 5..16 outputs: the gradient kernel's OM = 16 instances (4 or 16 input columns)
    for H, NH in [(16, 1), (16, 2), (16, 3), (16, 4), (32, 1), (32, 2), (32, 3), (32, 4), (64, 1), (64, 2), (64, 3), (11, 2), (40, 2)]:
        for act in ("tanh", "relu", "identity"):
            d = int(rs.choice([1, 2, 4, 7, 12, 16])) if (H, NH) != (64, 3) else int(rs.choice([1, 2, 4]))   # (3 x 64 with 16 x 16: over the LDS)
            o = int(rs.randint(5, 17))
            out.append(((d,) + (H,) * NH + (o,), act, bool(rs.rand() < 0.8), int(rs.choice([1, 63, 64, 65, 200, 513])),
                        int(rs.randint(1, 7)), bool(rs.rand() < 0.4)))
    return out


@pytest.mark.parametrize("case", _d8_cases(), ids=lambda c: f"{c[0]}-{c[1]}-b{int(c[2])}-N{c[3]}-B{c[4]}-idx{int(c[5])}")
def test_five_to_eight_inputs_on_the_fused_kernels(case):
    """Networks with 5..16 inputs or 5..16 outputs: gradient and forward of every activation
    run on the fused float64-MFMA kernels (k_fused_bwd_f64<H, NH, 8 | 16, UNB>, k_fused_fwd_f64<.., 8 | 16>; qn_fused_d8.hip) and
    agree with the layer-wise kernels."""
    dims, act, bias, N, B, use_idx = case
    rs = np.random.RandomState(sum(dims) * 1000 + N * 7 + B)
    arch = MLPArch(dims, act, bias)
    x, y = rs.randn(N, dims[0]), rs.randn(N, dims[-1])
    W = rs.randn(B, arch.nparams) / np.sqrt(max(dims))
    idx = rs.randint(0, N, size=(B, max(1, N // 2 + 3))) if use_idx else None
    op = BatchedMLP(arch, x, y)
    assert op.path(B, N, True) == _lib.PATH_FUSED and op.path(B, N, False) == _lib.PATH_FUSED
    res = {}
    for path in (_lib.PATH_GENERIC, _lib.PATH_AUTO):
        old = op.set_path(path)
        try:
            s, g = op.sse_grad(W, row_idx=idx)
            s2, pr = op.sse_pred(W, row_idx=idx)
        finally:
            op.set_path(old)
        res[path] = (s.cpu().numpy(), g.cpu().numpy(), s2.cpu().numpy(), pr.cpu().numpy())
    a, b = res[_lib.PATH_GENERIC], res[_lib.PATH_AUTO]
    np.testing.assert_allclose(b[0], a[0], rtol=1e-12)
    np.testing.assert_allclose(b[2], a[2], rtol=1e-12)
    assert np.abs(b[1] - a[1]).max() <= 1e-10 * max(np.abs(a[1]).max(), 1e-300)
    np.testing.assert_allclose(b[3], a[3], rtol=1e-11, atol=1e-12)
    # a not-finite input in the LAST column (beyond the four the DP = 4 instances hold): same NaN pattern as the layer-wise kernels
    if N > 5:
        x2 = x.copy(); x2[5, dims[0] - 1] = np.inf
        op2 = BatchedMLP(arch, x2, y)
        out = {}
        for path in (_lib.PATH_GENERIC, _lib.PATH_AUTO):
            old = op2.set_path(path)
            try:
                s, g = op2.sse_grad(W)
            finally:
                op2.set_path(old)
            out[path] = (s.cpu().numpy(), g.cpu().numpy())
        np.testing.assert_array_equal(np.isnan(out[_lib.PATH_AUTO][0]), np.isnan(out[_lib.PATH_GENERIC][0]))
        np.testing.assert_array_equal(np.isfinite(out[_lib.PATH_AUTO][1]), np.isfinite(out[_lib.PATH_GENERIC][1]))


def _dispatch_cases(n=120, seed=2024):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        nh = int(rs.randint(0, 5))
        wmax = int(rs.choice([8, 20, 64, 70, 128, 200, 300]))
        hid = tuple(int(rs.randint(1, wmax + 1)) for _ in range(nh))
        if nh and rs.rand() < 0.4:
            hid = (hid[0],) * nh                                     # uniform widths
        d, o = int(rs.choice([1, 2, 3, 4, 5, 8, 12, 16, 20])), int(rs.choice([1, 2, 4, 5, 9, 16, 17]))
        out.append(((d,) + hid + (o,), str(rs.choice(["tanh", "tanh", "relu", "identity"])), bool(rs.rand() < 0.8),
                    int(rs.choice([1, 2, 63, 64, 65, 130, 333])), int(rs.randint(1, 5)), bool(rs.rand() < 0.3)))
    return out


@pytest.mark.parametrize("case", _dispatch_cases(), ids=lambda c: f"{c[0]}-{c[1]}-b{int(c[2])}-N{c[3]}-B{c[4]}-idx{int(c[5])}")
def test_default_dispatch_equals_exact_layerwise_kernels(case):
    """Random shapes through every dispatch rule (fused / streaming / padded twins / wide first and last layer / layer-wise):
    the default path gives what the exact-width layer-wise kernels give."""
    dims, act, bias, N, B, use_idx = case
    rs = np.random.RandomState(sum(dims) * 7 + N + B)
    arch = MLPArch(dims, act, bias)
    x, y = rs.randn(N, dims[0]), rs.randn(N, dims[-1])
    W = rs.randn(B, arch.nparams) / np.sqrt(max(dims))
    idx = rs.randint(0, N, size=(B, max(1, N // 2 + 1))) if use_idx else None
    op = BatchedMLP(arch, x, y)
    L = _lib.lib()
    res = {}
    for path in (_lib.PATH_GENERIC, _lib.PATH_AUTO):
        old = op.set_path(path)
        try:
            s, g = op.sse_grad(W, row_idx=idx)
            s2, pr = op.sse_pred(W, row_idx=idx)
            s3 = op.sse(W, row_idx=idx)
        finally:
            op.set_path(old)
        res[path] = (s.cpu().numpy(), g.cpu().numpy(), s2.cpu().numpy(), pr.cpu().numpy(), s3.cpu().numpy())
    a, b = res[_lib.PATH_GENERIC], res[_lib.PATH_AUTO]
    for k in (0, 2, 4):
        np.testing.assert_allclose(b[k], a[k], rtol=1e-11)
    assert np.abs(b[1] - a[1]).max() <= 1e-10 * max(np.abs(a[1]).max(), 1e-300)
    np.testing.assert_allclose(b[3], a[3], rtol=1e-10, atol=1e-11)


@pytest.mark.parametrize("dims,act,where,val", [
    ((4, 33, 1), "tanh", "x", -np.inf),              # zero-padded twin (33 -> 64): padded units meet an infinite input
    ((1, 8, 8, 1), "tanh", "x", np.inf),             # twin 8 -> 16
    ((2, 100, 100, 1), "tanh", "x", np.inf),         # twin 100 -> 128 (int8-slice kernels)
    ((3, 50, 50, 2), "relu", "w", 1e200),            # relu activations overflow on a twin
    ((2, 64, 64, 1), "relu", "w", np.nan),           # relu(NaN) = NaN, not 0
    ((2, 16, 16, 1), "relu", "w", np.inf),           # relu backward: a select, not a product with 0 / 1
    ((1, 8, 8, 1), "identity", "w", -np.inf),        # ragged tail of an unbounded activation
])
def test_not_finite_values_on_twins_and_unbounded_activations_follow_torch(dims, act, where, val):
    """Cases the randomised sweep (tests/fuzz_all.py: run_exceptional) found: NaN / +Inf / -Inf pattern of SSE and predictions
    as the reference's torch ops, gradient finite in the same places (an entry that is +-Inf there may be NaN)."""
    from oracle import mlp_ref
    N, B = 45, 3
    rs = np.random.RandomState(5)
    x = rs.uniform(-1, 1, (N, dims[0])); y = rs.randn(N, dims[-1])
    arch = MLPArch(dims, act)
    W = 0.5 * rs.randn(B, arch.nparams) / np.sqrt(dims[1])
    if where == "x":
        x[7, 0] = val
    else:
        W[1, dims[0] * dims[1] + dims[1] + 5] = val                       # an entry of the second layer's matrix
    op = BatchedMLP(arch, x, y)
    mod = mlp_ref.build_module(mlp_ref.MLPSpec(dims, act))
    cls = lambda v: np.where(np.isnan(v), 3, np.where(np.isposinf(v), 1, np.where(np.isneginf(v), 2, 0)))
    with np.errstate(all="ignore"):
        ref_s = np.array([mlp_ref.sse(mod, W[b], x, y) for b in range(B)])
        ref_p = np.stack([mlp_ref.forward_flat(mod, W[b], x) for b in range(B)])
        ref_g = np.stack([-2.0 * mlp_ref.logpostgrad(mod, W[b], x, [v for v in y], 1.0) for b in range(B)])
    for path in (_lib.PATH_AUTO, _lib.PATH_GENERIC):
        old = op.set_path(path)
        try:
            s, g = op.sse_grad(W)
            s2, pr = op.sse_pred(W)
        finally:
            op.set_path(old)
        s, g, s2, pr = (t.cpu().numpy() for t in (s, g, s2, pr))
        assert np.array_equal(cls(s), cls(ref_s)) and np.array_equal(cls(s2), cls(ref_s))
        assert np.array_equal(cls(pr.reshape(ref_p.shape)), cls(ref_p))
        cg, cr = cls(g), cls(ref_g)
        cg = np.where((cg == 3) & ((cr == 1) | (cr == 2)), cr, cg)
        assert np.array_equal(cg, cr)
        for b in range(B):
            fin = np.isfinite(ref_g[b])
            if fin.any():
                assert np.abs(g[b][fin] - ref_g[b][fin]).max() <= 1e-9 * max(np.abs(ref_g[b][fin]).max(), 1e-300)
            if np.isfinite(ref_s[b]):
                np.testing.assert_allclose(s[b], ref_s[b], rtol=1e-11)
