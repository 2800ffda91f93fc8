"""CPU: argument validation of every C-ABI entry point (no GPU needed: each call must reject its bad arguments with
QN_EINVAL / QN_EUNSUPPORTED and a message before any HIP call), descriptor life cycle, size queries.  Also the body of
the host-side AddressSanitizer run (tools/asan_host.sh builds the library with the HOST code instrumented and runs this
file and test_abi_exports.py under it; GPU sanitizers are not available on this pool)."""
import ctypes

import pytest

from quinn_amd import _lib

EINVAL, EUNSUPPORTED = -1, -4
vp = ctypes.c_void_p


@pytest.fixture(scope="module")
def L():
    _lib.build()
    return _lib.lib()


def _desc(L, dims, act=1, bias=1):
    h = vp()
    arr = (ctypes.c_int * len(dims))(*dims)
    assert L.qn_mlp_desc_create(arr, len(dims), act, bias, ctypes.byref(h)) == 0
    return h


def test_descriptor_validation_and_sizes(L):
    h = vp()
    one = (ctypes.c_int * 1)(3)
    assert L.qn_mlp_desc_create(one, 1, 1, 1, ctypes.byref(h)) == EINVAL and b"ndims" in L.qn_last_error()
    assert L.qn_mlp_desc_create(None, 3, 1, 1, ctypes.byref(h)) == EINVAL
    bad = (ctypes.c_int * 3)(1, 0, 1)
    assert L.qn_mlp_desc_create(bad, 3, 1, 1, ctypes.byref(h)) != 0
    for dims in [(1, 1), (2, 7, 1), (1, 64, 64, 64, 1), (3, 50, 50, 2), (2, 128, 128, 128, 1), (1, 256, 256, 256, 256, 1),
                 (16, 64, 64, 16), (5, 300, 17)]:
        for bias in (0, 1):
            d = _desc(L, dims, bias=bias)
            p = L.qn_mlp_num_params(d)
            assert p == sum(a * b + (b if bias else 0) for a, b in zip(dims[:-1], dims[1:]))
            for B, Nb in [(1, 1), (3, 100), (64, 4096)]:
                for grad in (0, 1):
                    for dt in (0, 1):
                        assert L.qn_workspace_bytes(d, B, Nb, grad, dt) > 0
                        assert L.qn_mlp_path(d, B, Nb, grad, dt) in (1, 2)
                        assert L.qn_mlp_sse_parts(d, B, Nb, dt) >= 1
            assert L.qn_workspace_bytes(d, 0, 10, 0, 0) == 0
            # per-descriptor kernel-family override
            assert L.qn_mlp_desc_set_path(d, 1) == 0 and L.qn_mlp_path(d, 8, 64, 0, 0) == 1
            assert L.qn_mlp_desc_set_path(d, 99) == 1 and L.qn_mlp_desc_set_path(d, 0) == 1
            # row split of the fused kernels planned for a larger batch than the launch's (chain groups): more chains -> fewer
            # (or as many) row shares per chain; a negative value only queries
            n1 = L.qn_mlp_sse_parts(d, 8, 4096, 0)
            assert L.qn_mlp_desc_set_plan_batch(d, 64) == 0 and L.qn_mlp_desc_set_plan_batch(d, -1) == 64
            assert 1 <= L.qn_mlp_sse_parts(d, 8, 4096, 0) <= n1 and L.qn_mlp_sse_parts(d, 64, 4096, 0) == L.qn_mlp_sse_parts(d, 8, 4096, 0)
            assert L.qn_mlp_desc_set_plan_batch(d, 0) == 64 and L.qn_mlp_sse_parts(d, 8, 4096, 0) == n1
            assert L.qn_mlp_desc_destroy(d) == 0
    assert L.qn_mlp_num_params(None) == -1 and L.qn_mlp_desc_set_path(None, 0) == EINVAL
    assert L.qn_mlp_desc_set_plan_batch(None, 4) == EINVAL


def test_residual_network_descriptor(L):
    coef = (ctypes.c_double * 8)(1, 0, 1, 0.25, 1, 0.5, 1, 0.75)
    h = vp()
    assert L.qn_rnet_desc_create(1, 3, 1, 4, 2, coef, 1, 1, 1, 1, 0, ctypes.byref(h)) == 0
    assert L.qn_mlp_num_params(h) == 3 + 3 + 3 + 1 + 2 * 9 + 2 * 3
    assert L.qn_workspace_bytes(h, 4, 13, 1, 0) > 0
    # which tensors enter a step: 8 entries; one marked unused must have coefficient 0 (entry 1 has, entry 3 has not)
    ub = ctypes.c_ubyte * 8
    assert L.qn_rnet_desc_set_uses(h, ub(1, 0, 1, 1, 1, 1, 1, 1), 8) == 0
    assert L.qn_rnet_desc_set_uses(h, ub(1, 1, 1, 0, 1, 1, 1, 1), 8) == EINVAL
    assert L.qn_rnet_desc_set_uses(h, ub(1, 1, 1, 1, 1, 1, 1, 1), 7) == EINVAL
    assert L.qn_rnet_desc_set_uses(h, None, 8) == EINVAL and L.qn_rnet_desc_set_uses(None, ub(), 8) == EINVAL
    m = _desc(L, (1, 4, 1))
    assert L.qn_rnet_desc_set_uses(m, ub(1, 1, 1, 1, 1, 1, 1, 1), 8) == EINVAL                    # not a residual network
    assert L.qn_mlp_desc_destroy(m) == 0
    assert L.qn_mlp_desc_destroy(h) == 0
    assert L.qn_rnet_desc_create(1, 3, 1, 0, 2, coef, 1, 1, 1, 1, 0, ctypes.byref(h)) != 0        # no steps
    assert L.qn_rnet_desc_create(2, 3, 1, 4, 2, coef, 1, 1, 0, 1, 0, ctypes.byref(h)) != 0        # no pre layer but indim != rdim


def test_compute_entry_points_reject_bad_arguments(L):
    d = _desc(L, (1, 16, 16, 1))
    one = ctypes.c_void_p(8)            # a non-null placeholder pointer; never dereferenced: validation fails first
    assert L.qn_mlp_sse_fwd(d, 0, None, one, one, None, 4, 10, 10, one, None, one, 1 << 20, None) == EINVAL
    assert L.qn_mlp_sse_fwd(d, 7, one, one, one, None, 4, 10, 10, one, None, one, 1 << 20, None) == EINVAL     # dtype
    assert L.qn_mlp_sse_fwd(d, 0, one, one, one, None, 0, 10, 10, one, None, one, 1 << 20, None) == EINVAL     # B = 0
    assert L.qn_mlp_sse_fwd(d, 0, one, one, one, None, 4, 10, 5, one, None, one, 1 << 20, None) == EINVAL      # Nb != N without row_idx
    assert b"row_idx" in L.qn_last_error()
    assert L.qn_mlp_sse_fwdbwd(d, 0, one, one, one, None, 70000, 10, 10, one, None, one, one, 1 << 20, None) == EINVAL
    assert L.qn_mlp_sse_fwd_parts(None, 0, one, one, one, None, 4, 10, 10, one, one, 1 << 20, None) != 0
    assert L.qn_vi_sample_kl(None, one, one, 2, 5, 0.5, 1.0, 1.0, 0, one, one, one, None) != 0
    assert L.qn_vi_grad(one, one, one, None, 2, 5, 0.5, 1.0, 1.0, 1.0, 1.0, 0, one, one, None) != 0
    assert L.qn_adam_batched(None, one, one, one, one, 2, 5, 0, 1.0, 0.0, 0.9, 0.999, 1e-8, 1, None) != 0
    assert L.qn_mcmc_propose(None, None, 0.0, 0, 0, 5, 1, one, one, None) == EINVAL
    assert L.qn_mcmc_propose(one, None, 0.0, 2, 0, 5, 1, one, one, None) == EINVAL                               # cur without sd
    assert L.qn_mcmc_propose_hist(one, one, one, one, one, None, 1.0, 1.0, 2, 0, 5, 5, 4, 1, one, one, None) == EINVAL  # odd pstride
    assert L.qn_mcmc_apply_delta(one, one, 64, 1.0, 2, 0, 5, 1, one, one, None) == EINVAL                        # t >= TB
    assert L.qn_mcmc_hist_block_steps() == 64
    assert L.qn_hmc_parts(0) == EINVAL and L.qn_hmc_parts(8513) == 9 and L.qn_hmc_parts(10 ** 7) == 64
    assert L.qn_hmc_begin(one, one, 0.0, 0.1, 2, 0, 5, 1, one, one, one, one, None) == EINVAL                    # sigma = 0
    assert L.qn_hmc_leap(one, 3, 0.1, 0.1, 0, 2, 5, one, one, None, None) == EINVAL                              # dtype
    assert L.qn_hmc_leap(one, 0, 0.1, 0.1, 1, 2, 5, one, one, None, None) == EINVAL                              # last without K parts
    assert L.qn_hmc_accept(one, one, one, one, one, 0.1, 10, 2, 0, 5, 3, 1, one, one, one, one, one, None, one, one, one,
                           one, 2, None) == EINVAL                                                              # parity
    assert L.qn_pred_moments(one, 0, 1, 10, one, one, None) == EINVAL                                            # variance of one member
    assert L.qn_pred_moments(None, 0, 5, 10, one, None, None) == EINVAL
    assert L.qn_debug_tanh(None, one, 4, None) != 0
    assert L.qn_mlp_desc_destroy(d) == 0
