"""GPU: BASELINE.json configs 3-5 at FULL size, through size-independent properties (the oracle
would need minutes per evaluation here): dataset additivity of SSE and gradient, run-to-run bitwise
determinism, replicated units give identical results, a directional finite-difference check of the
gradient, and oracle spot checks of single units."""
import numpy as np
import pytest
import torch

from oracle import mlp_ref
from quinn_amd.ops import MLPArch, BatchedMLP

pytestmark = pytest.mark.gpu


def _weights(arch, B, scale=0.1):
    W = np.stack([scale * np.random.RandomState(1000 + c).randn(arch.nparams) for c in range(min(B, 8))])
    return np.tile(W, ((B + 7) // 8, 1))[:B]            # 8 distinct vectors, replicated


def _check(arch, N, B, d, spot=1):
    x, y = mlp_ref.synthetic_data(N, d, 0.02, seed=0)
    W = _weights(arch, B)
    op = BatchedMLP(arch, x, y)
    s1, g1 = op.sse_grad(W)
    s2, g2 = op.sse_grad(W)
    assert torch.equal(s1, s2) and torch.equal(g1, g2)                       # deterministic
    assert torch.equal(s1[:8], s1[8:16]) and torch.equal(g1[0], g1[8])       # replicas agree bitwise
    sf = op.sse(W)
    np.testing.assert_allclose(sf.cpu().numpy(), s1.cpu().numpy(), rtol=1e-12)
    h = N // 2
    lo, hi = BatchedMLP(arch, x[:h], y[:h]), BatchedMLP(arch, x[h:], y[h:])
    sl, gl = lo.sse_grad(W[:8])
    sh, gh = hi.sse_grad(W[:8])
    np.testing.assert_allclose((sl + sh).cpu().numpy(), s1[:8].cpu().numpy(), rtol=1e-12)
    gsum, gref = (gl + gh).cpu().numpy(), g1[:8].cpu().numpy()
    assert np.abs(gsum - gref).max() <= 1e-10 * np.abs(gref).max()
    # directional derivative: (sse(w + e v) - sse(w - e v)) / 2e  ==  g . v
    v = np.random.RandomState(7).randn(arch.nparams); v /= np.linalg.norm(v)
    e = 1e-6
    sp = op.sse(W[:1] + e * v).item(); sm = op.sse(W[:1] - e * v).item()
    fd, an = (sp - sm) / (2 * e), float(g1[0].cpu().numpy() @ v)
    assert abs(fd - an) <= 1e-5 * max(1.0, abs(an))
    mod = mlp_ref.build_module(mlp_ref.MLPSpec(arch.dims, arch.activ))
    for b in range(spot):
        ref = mlp_ref.sse(mod, W[b], x, y)
        assert abs(s1[b].item() - ref) <= 1e-11 * ref


def test_cfg3_vi_shape_full_size():        # 128 MC samples, 3x128, N=8192, d=2
    _check(MLPArch((2, 128, 128, 128, 1), "tanh"), 8192, 128, 2)


def test_cfg3_shape_full_size_relu():      # the reference's default activation on the same shape (row-scale int8 kernels, round 4)
    from quinn_amd import _lib
    arch = MLPArch((2, 128, 128, 128, 1), "relu")
    x, y = mlp_ref.synthetic_data(64, 2, 0.02, seed=0)
    assert BatchedMLP(arch, x, y).arith(128, 8192, True) == _lib.ARITH_I8_WIDE
    _check(arch, 8192, 128, 2)


def test_cfg2_shape_full_size_relu():      # 64 chains, 3x64, N=4096: the fused int8-slice kernels with row scales
    _check(MLPArch((1, 64, 64, 64, 1), "relu"), 4096, 64, 1, spot=2)


def test_cfg4_ensemble_shape_full_size():  # 512 members, 4x256, N=16384 (workspace is chunked over members)
    _check(MLPArch((1, 256, 256, 256, 256, 1), "tanh"), 16384, 512, 1)


def test_cfg5_hmc_shape_full_size():       # 256 chains, 4x256, N=32768
    _check(MLPArch((1, 256, 256, 256, 256, 1), "tanh"), 32768, 256, 1)
