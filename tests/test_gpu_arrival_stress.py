"""GPU: the last-arriver protocol of the fused kernels (csrc/qn_fused_args.h: qn_arrive_tagged / qn_sse_finish -- a chain's row-split
workgroups store their partial sums, bump a tagged counter with relaxed agent-scope atomics, and the last one to arrive adds the
partials left to right) under many launches, row splits from 2 to 32 per chain and batches from one chain to more chains than the
chip has XCD slots: the in-kernel sum must equal the left-to-right sum of the partials that `qn_mlp_sse_fwd_parts` hands out
(the same kernel without the arrival step) BIT FOR BIT in every launch, and the gradient step's SSE / gradient (slab reduced by
the flagged-chain pass behind its own arrival counter) must not change from launch to launch.  (Round-3 advisor finding: the
ordering rests on gfx950's sc1 write-through stores; a reordering would show up as a wrong or stale sum, not as a fault.)"""
import numpy as np
import pytest
import torch

from quinn_amd import _lib
from quinn_amd.ops import BatchedMLP, MLPArch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("act", ["tanh", "relu"])
@pytest.mark.parametrize("B", [1, 3, 17, 64, 200])
def test_in_kernel_sse_sum_equals_the_partials_left_to_right(B, act):
    rs = np.random.RandomState(B)
    N = 4096
    x = rs.rand(N, 1) * 2 - 1
    y = np.sin(3 * x) + 0.05 * rs.randn(N, 1)
    arch = MLPArch((1, 64, 64, 64, 1), act)
    op = BatchedMLP(arch, x, y)
    assert op.arith(B) == _lib.ARITH_I8_FUSED
    W = torch.as_tensor(0.2 * rs.randn(B, arch.nparams), device=op.device)
    parts = op.sse_parts(W)
    assert parts.shape[1] == min(32, -(-512 // B)) or parts.shape[1] >= 2            # 32 / 32 / 31 / 8 / 3 row shares per chain
    ref = parts[:, 0].clone()
    for j in range(1, parts.shape[1]):
        ref = ref + parts[:, j]
    for rep in range(300):
        s = op.sse(W)
        if rep % 50 == 0:                                   # (other launches in between: the counters' tags must not collide)
            op.sse(W[: max(1, B // 2)])
        assert torch.equal(s, ref), (rep, (s - ref).abs().max().item())


@pytest.mark.parametrize("B", [2, 64, 150])
def test_gradient_step_is_bitwise_stable_over_many_launches(B):
    rs = np.random.RandomState(100 + B)
    N = 2048
    x = rs.rand(N, 2) * 2 - 1
    y = np.sin(x.sum(axis=1, keepdims=True)) + 0.05 * rs.randn(N, 1)
    arch = MLPArch((2, 64, 64, 64, 1), "tanh")
    op = BatchedMLP(arch, x, y)
    assert op.arith(B, want_grad=True) == _lib.ARITH_I8_FUSED
    W = torch.as_tensor(0.2 * rs.randn(B, arch.nparams), device=op.device)
    W[0, 64 + 64 + 7] = 3e7                                  # one chain outside the int8 contract: the flagged second pass runs too
    s0, g0 = op.sse_grad(W)
    s0, g0 = s0.clone(), g0.clone()
    fwd = op.sse(W)
    torch.testing.assert_close(s0, fwd, rtol=1e-11, atol=0)
    for rep in range(200):
        s, g = op.sse_grad(W)
        assert torch.equal(s, s0) and torch.equal(g, g0), rep
