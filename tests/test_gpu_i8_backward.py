"""GPU: the sliced int8-product forward + BACKWARD kernel for 64-wide tanh and relu networks (csrc/qn_fused_bwd_i8.hip; the gradient the
reference gets from autograd, quinn/nns/nnwrap.py:128-150 through quinn/solvers/nn_mcmc.py:73-98) against the oracle, against
the float64-MFMA fused kernel (QN_PATH_FUSED_DP) and against the layer-wise kernels (QN_PATH_GENERIC): SSE rtol 1e-11,
gradient 1e-10 of max |g| (measured ~1e-13); chains outside the fast path's contract (flagged, recomputed by the float64 kernel);
bitwise determinism; dataset additivity at the BASELINE size.  (The MH-acceptance test of tests/test_gpu_i8_forward.py steps HMC
chains with this kernel's gradient on one side and the float64 kernel's on the other.)"""
import numpy as np
import pytest
import torch

from oracle import mlp_ref
from quinn_amd import _lib
from quinn_amd.ops import BatchedMLP, MLPArch

pytestmark = pytest.mark.gpu


def _data(N, d, seed=0, noise=0.02):
    rs = np.random.RandomState(seed)
    x = rs.rand(N, d) * 2 * np.pi - np.pi
    y = np.sin(x).sum(axis=1, keepdims=True) + noise * rs.randn(N, 1)
    return x, y


def _weights(arch, B, wscale, seed):
    rs = np.random.RandomState(seed)
    parts = []
    for a_, b_ in zip(arch.dims[:-1], arch.dims[1:]):
        parts.append(wscale * rs.randn(B, b_ * a_) / np.sqrt(a_) * (3.0 if a_ > 4 else 1.0))
        parts.append(wscale * rs.randn(B, b_))
    return np.concatenate(parts, axis=1)


def _three(op, W, row_idx=None):
    out = []
    for path in (_lib.PATH_FUSED, _lib.PATH_FUSED_DP, _lib.PATH_GENERIC):
        old = op.set_path(path)
        try:
            s, g = op.sse_grad(W, row_idx=row_idx)
            out.append((s.cpu().numpy(), g.cpu().numpy()))
        finally:
            op.set_path(old)
    return out


@pytest.mark.parametrize("dims,N,B,wscale", [((1, 64, 64, 64, 1), 4096, 64, 0.1), ((1, 64, 64, 1), 300, 5, 1.0), ((2, 64, 64, 64, 1), 1000, 7, 2.0),
                                             ((1, 64, 64, 64, 1), 64, 1, 0.3), ((2, 64, 64, 1), 333, 5, 0.5), ((1, 50, 50, 50, 1), 500, 4, 0.5),
                                             ((1, 64, 64, 64, 1), 77, 300, 0.3),
                                             # 3 and 4 inputs: the 4-column image of W0 and the tanh table that ends at n = 1264
                                             ((3, 64, 64, 64, 1), 1000, 6, 0.4), ((4, 64, 64, 64, 1), 333, 5, 2.5), ((4, 64, 64, 1), 130, 9, 0.5),
                                             ((3, 40, 40, 40, 1), 257, 3, 0.7)],
                         ids=["cfg2", "2hid", "d2_bigw", "one_chain", "d2_2hid", "padded50", "many_chains", "d3", "d4_bigw", "d4_2hid",
                              "d3_padded40"])
@pytest.mark.parametrize("act", ["tanh", "relu"])
def test_gradient_matches_oracle_and_float64_kernels(dims, N, B, wscale, act):
    x, y = _data(N, dims[0])
    arch = MLPArch(dims, act)
    if act == "relu":
        wscale = min(wscale, 0.5)
    W = _weights(arch, B, wscale, sum(dims) + N)
    op = BatchedMLP(arch, x, y)
    assert op.path(B, N, True) == _lib.PATH_FUSED and op.arith(B, N, True) == _lib.ARITH_I8_FUSED
    (s8, g8), (sd, gd), (sg, gg) = _three(op, W)
    gmax = np.abs(gg).max(axis=1, keepdims=True)
    np.testing.assert_allclose(s8, sd, rtol=1e-11)
    np.testing.assert_allclose(s8, sg, rtol=1e-11)
    e8d, e8g, edg = (np.abs(g8 - gd) / gmax).max(), (np.abs(g8 - gg) / gmax).max(), (np.abs(gd - gg) / gmax).max()
    print(f"max |g - g_ref| / max|g|: int8 slices vs f64 MFMA {e8d:.2e}, vs layer-wise {e8g:.2e}; f64 MFMA vs layer-wise {edg:.2e}")
    assert e8d <= 1e-10 and e8g <= 1e-10
    mod = mlp_ref.build_module(mlp_ref.MLPSpec(dims, act))
    yd = [v for v in y]
    for b in range(min(B, 3)):
        gref = -2.0 * mlp_ref.logpostgrad(mod, W[b], x, yd, 1.0)                     # d SSE / d w
        assert np.abs(g8[b] - gref).max() <= 1e-10 * np.abs(gref).max(), b
        assert abs(s8[b] / mlp_ref.sse(mod, W[b], x, y) - 1) <= 1e-11


@pytest.mark.parametrize("act", ["tanh", "relu"])
def test_row_subsets_ragged_tail_and_determinism(act):
    dims = (2, 64, 64, 64, 1)
    x, y = _data(777, 2, seed=3)
    arch = MLPArch(dims, act)
    rs = np.random.RandomState(5)
    W = _weights(arch, 6, 0.4, 11)
    idx = rs.randint(0, 777, size=(6, 403))
    op = BatchedMLP(arch, x, y)
    (s8, g8), _, (sg, gg) = _three(op, W, row_idx=idx)
    np.testing.assert_allclose(s8, sg, rtol=1e-11)
    assert (np.abs(g8 - gg) / np.abs(gg).max(axis=1, keepdims=True)).max() <= 1e-10
    s2, g2 = op.sse_grad(W, row_idx=idx)
    assert np.array_equal(s2.cpu().numpy(), s8) and np.array_equal(g2.cpu().numpy(), g8)       # two launches, same bits


@pytest.mark.parametrize("where", ["weight_nan", "weight_inf", "weight_huge", "weight_2e25", "bias_nan", "x_nan", "x_inf", "y_nan", "y_huge", "w0_inf", "tiny_weights"])
@pytest.mark.parametrize("d", [1, 3])                                   # (3 inputs: the 4-column W0 image, the shorter tanh table)
@pytest.mark.parametrize("act", ["tanh", "relu"])
def test_chains_outside_the_contract_are_recomputed_in_float64(where, d, act):
    """One chain (or the data) breaks the fast path's contract: the flagged chains come from k_fused_bwd_f64 -- NaN / Inf
    pattern and finite values of the layer-wise kernels -- and the OTHER chains' results do not change by a bit."""
    dims = (d, 64, 64, 64, 1)
    arch = MLPArch(dims, act)
    x, y = _data(200, d, seed=1)
    W = _weights(arch, 3, 0.3, 2)
    clean = BatchedMLP(arch, x, y)
    s0, g0 = (t.cpu().numpy() for t in clean.sse_grad(W))
    off_w1 = 64 * d + 64 + 5 * 64 + 7                                    # an entry of the first hidden matrix
    data_case = where in ("x_nan", "x_inf", "y_nan", "y_huge")
    if where == "weight_nan": W[1, off_w1] = np.nan
    if where == "weight_inf": W[1, off_w1] = np.inf
    if where == "weight_huge": W[1, off_w1] = 1e200
    if where == "weight_2e25": W[1, off_w1] = 2.0 ** 25                  # finite, but beyond what a sliced matrix may hold (2^20)
    if where == "bias_nan": W[1, 64 * d + 3] = np.nan
    if where == "w0_inf": W[1, 3] = -np.inf
    if where == "tiny_weights": W[1] *= 1e-4                             # activations all tiny: fixed-scale digits lose relative accuracy
    if where == "x_nan": x[17, 0] = np.nan
    if where == "x_inf": x[150, 0] = -np.inf
    if where == "y_nan": y[17, 0] = np.nan
    if where == "y_huge": y[17, 0] = 1e200
    op = BatchedMLP(arch, x, y)
    (s8, g8), _, (sg, gg) = _three(op, W)
    assert np.array_equal(np.isnan(s8), np.isnan(sg)) and np.array_equal(np.isinf(s8), np.isinf(sg))
    ok = np.isfinite(sg)
    np.testing.assert_allclose(s8[ok], sg[ok], rtol=1e-11)
    for b in range(3):
        fin = np.isfinite(gg[b])
        assert np.array_equal(np.isnan(g8[b]), np.isnan(gg[b])) or where in ("weight_inf", "w0_inf", "x_inf", "y_huge"), b   # (+-Inf entries may come out NaN: DESIGN 4.2)
        if fin.any():
            scale = np.abs(gg[b][fin]).max()
            assert np.abs(g8[b][fin] - gg[b][fin]).max() <= 1e-9 * scale, (b, where)
    if not data_case:
        for b in (0, 2):                                                 # the untouched chains: bit for bit what they were
            assert s8[b] == s0[b] and np.array_equal(g8[b], g0[b]), b


def test_additivity_over_the_dataset_at_full_size():
    """BASELINE configs[1] size: the gradient over the 4096 rows equals the sum of the gradients over two halves of the rows
    (a size-independent property: no oracle run needed), and the SSE likewise."""
    dims = (1, 64, 64, 64, 1)
    arch = MLPArch(dims, "tanh")
    x, y = _data(4096, 1, seed=7)
    W = 0.1 * np.random.RandomState(8).randn(64, arch.nparams)
    full = BatchedMLP(arch, x, y)
    assert full.path(64, 4096, True) == _lib.PATH_FUSED
    s, g = (t.cpu().numpy() for t in full.sse_grad(W))
    sa, ga = (t.cpu().numpy() for t in BatchedMLP(arch, x[:1500], y[:1500]).sse_grad(W))
    sb, gb = (t.cpu().numpy() for t in BatchedMLP(arch, x[1500:], y[1500:]).sse_grad(W))
    np.testing.assert_allclose(s, sa + sb, rtol=1e-12)
    gmax = np.abs(g).max(axis=1, keepdims=True)
    assert (np.abs(g - (ga + gb)) / gmax).max() <= 1e-11
    s2, g2 = (t.cpu().numpy() for t in full.sse_grad(W))
    assert np.array_equal(s, s2) and np.array_equal(g, g2)


def test_gradient_inside_a_hip_graph_equals_direct_launches():
    dims = (1, 64, 64, 64, 1)
    arch = MLPArch(dims, "tanh")
    x, y = _data(512, 1, seed=9)
    op = BatchedMLP(arch, x, y)
    Wt = op.weights(0.2 * np.random.RandomState(3).randn(16, arch.nparams))
    s_out = torch.empty(16, dtype=torch.float64, device=op.device)
    g_out = torch.empty(16, arch.nparams, dtype=torch.float64, device=op.device)
    s0, g0 = op.sse_grad(Wt)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=op.device)
    side.wait_stream(torch.cuda.current_stream(op.device))
    with torch.cuda.stream(side):
        op.sse_grad(Wt, out=(s_out, g_out))
    torch.cuda.current_stream(op.device).wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        op.sse_grad(Wt, out=(s_out, g_out))
    s_out.zero_(); g_out.zero_()
    gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(s_out, s0) and torch.equal(g_out, g0)


def test_exact_float64_option_selects_the_plain_float64_kernels():
    """`BatchedMLP.use_exact_float64()` / `NN_MCMC(kernels='float64')`: bit for bit the float64-MFMA kernels (64-wide) or
    the layer-wise float64 kernels (128-wide: no float64 fused gradient there), and closer to the oracle than the default."""
    from quinn_amd.nns.mlp import MLP
    from quinn_amd.solvers.nn_mcmc import NN_MCMC
    for dims, ref_path in (((1, 64, 64, 64, 1), _lib.PATH_FUSED_DP), ((2, 128, 128, 1), _lib.PATH_GENERIC)):
        arch = MLPArch(dims, "tanh")
        x, y = _data(300, dims[0], seed=4)
        W = _weights(arch, 3, 0.4, 21)
        exact, ref = BatchedMLP(arch, x, y), BatchedMLP(arch, x, y)
        assert exact.use_exact_float64() == ref_path
        ref.set_path(ref_path)
        (s0, g0), (s1, g1) = exact.sse_grad(W), ref.sse_grad(W)
        assert torch.equal(s0, s1) and torch.equal(g0, g1)
        assert torch.equal(exact.sse(W), ref.sse(W))
    torch.manual_seed(0)
    nn = MLP(1, 1, (64, 64), activ="tanh")
    x, y = _data(200, 1, seed=6)
    mod = mlp_ref.build_module(mlp_ref.MLPSpec((1, 64, 64, 1), "tanh"))
    errs = {}
    for kern in ("auto", "float64"):
        s = NN_MCMC(nn, verbose=False, kernels=kern)
        lpinfo = {'model': nn, 'xd': x, 'yd': [v for v in y], 'ltype': 'classical', 'lparams': {'sigma': 0.1}}
        w = 0.3 * np.random.RandomState(1).randn(s.pdim)
        errs[kern] = abs(s.logpost(w, lpinfo) / mlp_ref.logpost(mod, w, x, [v for v in y], 0.1) - 1)
    assert errs["float64"] <= 1e-14 and errs["auto"] <= 1e-11, errs
    with pytest.raises(ValueError):
        NN_MCMC(nn, verbose=False, kernels="fast")
