"""GPU parity of the batched operator (through the C ABI) against the golden fixtures and
the CPU oracle.  float64: rtol 1e-11 on SSE / log-posterior, 1e-10 (of max |grad|) on
gradients; float32: 2e-4 / 2e-3.  Both kernel families are exercised where supported."""
import numpy as np
import pytest
import torch

from conftest import load_golden, spec_of
from oracle import mlp_ref
from quinn_amd import _lib
from quinn_amd.ops import MLPArch, BatchedMLP, neg_log_post_from_sse

pytestmark = pytest.mark.gpu

TOL = {"float64": (1e-11, 1e-10), "float32": (2e-4, 2e-3)}


def paths_for(op, B, Nb, grad):
    """Kernel families to test for this shape: generic always, fused when supported."""
    L = _lib.lib()
    out = [_lib.PATH_GENERIC]
    old = op.set_path(_lib.PATH_AUTO)
    if op.path(B, Nb, grad) == _lib.PATH_FUSED:
        out += [_lib.PATH_FUSED, _lib.PATH_FUSED_DP]      # (64-wide tanh forwards: sliced int8 products / float64 MFMA)
    op.set_path(old)
    return out


class forced:
    def __init__(self, op, path):
        self.op, self.path = op, path

    def __enter__(self):
        self.old = self.op.set_path(self.path)

    def __exit__(self, *a):
        self.op.set_path(self.old)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("ci", range(5))
def test_g1_golden_logpost_grad_pred(ci, dtype):
    g = load_golden(f"g1_logpost_{ci}.npz")
    spec = spec_of(g)
    arch = MLPArch(spec.dims, spec.activ)
    op = BatchedMLP(arch, g["x"], g["y"], dtype=dtype)
    rt, gt = TOL[dtype]
    n, sigma = g["x"].shape[0], float(g["sigma"])
    for path in paths_for(op, 8, n, True):
        with forced(op, path):
            sse, grad = op.sse_grad(g["W"])
            sse2, pred = op.sse_pred(g["W"])
        lp = -neg_log_post_from_sse(sse.cpu().numpy(), n, sigma)
        np.testing.assert_allclose(lp, g["logpost"], rtol=rt, err_msg=f"path {path}")
        np.testing.assert_allclose(sse2.cpu().numpy(), sse.cpu().numpy(), rtol=rt)
        gl = -(0.5 * grad.double().cpu().numpy() / sigma ** 2)
        scale = np.abs(g["grad"]).max(axis=1, keepdims=True)
        assert np.max(np.abs(gl - g["grad"]) / scale) < gt, f"path {path}"
        np.testing.assert_allclose(pred.double().cpu().numpy(), g["pred"], rtol=gt, atol=gt)


CASES = [  # dims, activ, bias, N, B
    ((1, 16, 16, 1), "tanh", True, 256, 3),          # cfg1 shape
    ((1, 64, 64, 64, 1), "tanh", True, 300, 5),      # cfg2 shape, ragged N
    ((2, 128, 128, 128, 1), "tanh", True, 130, 2),   # cfg3 shape
    ((1, 256, 256, 256, 256, 1), "tanh", True, 70, 2),  # cfg4/5 shape
    ((3, 5, 7, 2), "relu", True, 33, 4),
    ((2, 8, 1), "identity", False, 17, 3),
    ((4, 3), "tanh", True, 9, 2),                    # single Linear layer
    ((1, 32, 32, 1), "relu", False, 1, 2),           # one data row
    ((2, 128, 128, 1), "tanh", True, 777, 1),        # MFMA GEMM layer path with split-K dW (ragged rows)
    ((1, 64, 128, 64, 2), "relu", True, 300, 3),     # non-uniform widths, all multiples of 64
    ((6, 64, 64, 64, 1), "tanh", True, 200, 3),      # 5..8 inputs: fused float64 kernels' DP = 8 instances (qn_fused_d8.hip)
    ((8, 32, 32, 2), "relu", True, 129, 3),
    ((5, 20, 20, 20, 1), "identity", False, 65, 2),  # (and on a zero-padded twin)
    ((12, 32, 32, 1), "tanh", True, 150, 3),         # 9..16 inputs: DP = 16 instances
    ((16, 16, 16, 16, 2), "relu", True, 77, 2),
    ((10, 64, 64, 64, 1), "relu", True, 100, 2),
    ((3, 32, 32, 8), "tanh", True, 120, 3),          # 5..16 outputs: the gradient kernel's OM = 16 instances (qn_fused_o16.hip)
    ((12, 16, 16, 16, 16), "relu", True, 90, 2),
    ((2, 64, 64, 6), "identity", True, 70, 2),
]


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("case", CASES, ids=[str(c[0]) + c[1] for c in CASES])
def test_random_shapes_vs_oracle(case, dtype):
    dims, activ, bias, N, B = case
    rs = np.random.RandomState(sum(dims) % 1000)
    arch = MLPArch(dims, activ, bias)
    spec = mlp_ref.MLPSpec(dims, activ, bias)
    mod = mlp_ref.build_module(spec)
    x = rs.randn(N, dims[0])
    y = rs.randn(N, dims[-1])
    W = rs.randn(B, arch.nparams) / np.sqrt(max(dims))
    op = BatchedMLP(arch, x, y, dtype=dtype)
    rt, gt = TOL[dtype]
    yd = [v for v in y]
    ref_lp = np.array([mlp_ref.logpost(mod, w, x, yd, 0.5) for w in W])
    ref_g = np.array([mlp_ref.logpostgrad(mod, w, x, yd, 0.5) for w in W])
    for path in paths_for(op, B, N, True):
        with forced(op, path):
            sse, grad = op.sse_grad(W)
            sse_f = op.sse(W)
        np.testing.assert_allclose(sse_f.cpu().numpy(), sse.cpu().numpy(), rtol=rt)
        lp = -neg_log_post_from_sse(sse.cpu().numpy(), N, 0.5)
        np.testing.assert_allclose(lp, ref_lp, rtol=rt, err_msg=f"path {path}")
        gl = -(0.5 * grad.double().cpu().numpy() / 0.25)
        scale = np.abs(ref_g).max(axis=1, keepdims=True) + 1e-300
        assert np.max(np.abs(gl - ref_g) / scale) < gt, f"path {path}"


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_row_idx_minibatches(dtype):
    """Per-member row subsets (ensemble minibatches): each member's SSE/grad equals the oracle's
    on exactly its rows, including repeated rows."""
    dims, N, B, Nb = (2, 16, 16, 2), 50, 4, 13
    rs = np.random.RandomState(5)
    arch = MLPArch(dims, "tanh")
    mod = mlp_ref.build_module(mlp_ref.MLPSpec(dims, "tanh"))
    x, y = rs.randn(N, 2), rs.randn(N, 2)
    W = 0.4 * rs.randn(B, arch.nparams)
    idx = rs.randint(0, N, size=(B, Nb))
    op = BatchedMLP(arch, x, y, dtype=dtype)
    rt, gt = TOL[dtype]
    for path in paths_for(op, B, Nb, True):
        with forced(op, path):
            sse, grad = op.sse_grad(W, row_idx=idx)
        for b in range(B):
            xb, yb = x[idx[b]], y[idx[b]]
            ref = -mlp_ref.logpost(mod, W[b], xb, [v for v in yb], 1.0)
            got = neg_log_post_from_sse(sse[b].item(), Nb, 1.0)
            assert abs(got - ref) <= rt * abs(ref)
            gref = -mlp_ref.logpostgrad(mod, W[b], xb, [v for v in yb], 1.0)
            gg = 0.5 * grad[b].double().cpu().numpy()
            assert np.max(np.abs(gg - gref)) <= gt * np.max(np.abs(gref))


def test_linearity_and_determinism_full_size():
    """cfg2 at full size (64 x 4096 x 3x64): size-independent properties.  (i) SSE over the
    dataset = SSE over its two halves summed; (ii) identical inputs -> bitwise identical
    outputs; (iii) duplicated weight vectors -> identical results in every slot."""
    arch = MLPArch((1, 64, 64, 64, 1), "tanh")
    x, y = mlp_ref.synthetic_data(4096, 1, 0.02, seed=0)
    W = np.stack([0.1 * np.random.RandomState(1000 + c).randn(arch.nparams) for c in range(64)])
    W[17] = W[3]
    op = BatchedMLP(arch, x, y)
    s1, g1 = op.sse_grad(W)
    s2, g2 = op.sse_grad(W)
    assert torch.equal(s1, s2) and torch.equal(g1, g2)
    assert s1[17] == s1[3] and torch.equal(g1[17], g1[3])
    lo = BatchedMLP(arch, x[:2048], y[:2048])
    hi = BatchedMLP(arch, x[2048:], y[2048:])
    sl, gl = lo.sse_grad(W)
    sh, gh = hi.sse_grad(W)
    np.testing.assert_allclose((sl + sh).cpu().numpy(), s1.cpu().numpy(), rtol=1e-12)
    np.testing.assert_allclose((gl + gh).cpu().numpy(), g1.cpu().numpy(), rtol=1e-9, atol=1e-9 * g1.abs().max().item())
    # spot-check 2 chains against the oracle at full size
    mod = mlp_ref.build_module(mlp_ref.MLPSpec(arch.dims, "tanh"))
    for b in (0, 63):
        ref = mlp_ref.sse(mod, W[b], x, y)
        assert abs(s1[b].item() - ref) <= 1e-11 * ref
