"""GPU: accuracy of the device float64 tanh (quinn_amd/csrc/qn_math.h) against a
high-precision reference (numpy longdouble / mpmath-free: tanh via expm1 in float128)."""
import ctypes

import numpy as np
import pytest
import torch

from quinn_amd import _lib

pytestmark = pytest.mark.gpu


def _ref_tanh(x):
    xl = x.astype(np.longdouble)
    e = np.expm1(-2 * np.abs(xl))
    return (np.sign(xl) * (-e / (2 + e))).astype(np.longdouble)


@pytest.mark.parametrize("fn", ["qn_debug_tanh", "qn_debug_tanh_finite"])
def test_tanh_f64_ulp_error(fn):
    """Both variants (qn_math.h): the NaN-propagating one and the one the fused kernels use when every weight
    and input is bounded.  Identical values for every non-NaN input."""
    rs = np.random.RandomState(0)
    xs = np.concatenate([rs.uniform(-20, 20, 200000), rs.uniform(-1, 1, 200000), rs.uniform(-1e-3, 1e-3, 50000),
                         10.0 ** rs.uniform(-300, -3, 20000), np.array([0.0, -0.0, 19.0, 19.07, 25.0, -40.0, 1e300,
                                                                           -1e300, np.inf, -np.inf, 5e-324, 0.5493061443340549])])
    x = torch.tensor(xs, device="cuda")
    y = torch.empty_like(x)
    L = _lib.lib()
    _lib.check(getattr(L, fn)(x.data_ptr(), y.data_ptr(), x.numel(), None), fn)
    torch.cuda.synchronize()
    y2 = torch.empty_like(x)
    _lib.check(L.qn_debug_tanh(x.data_ptr(), y2.data_ptr(), x.numel(), None), "qn_debug_tanh")
    assert torch.equal(y, y2)
    got = y.cpu().numpy()
    ref = _ref_tanh(xs)
    ulp = np.spacing(np.abs(ref.astype(np.float64)))
    err = np.abs(got.astype(np.longdouble) - ref) / np.maximum(ulp, 5e-324)
    assert err.max() < 3.0, (err.max(), xs[np.argmax(err)])
    assert np.mean(err) < 0.7
    assert got[np.where(xs == np.inf)[0][0]] == 1.0 and got[np.where(xs == -np.inf)[0][0]] == -1.0
    assert np.signbit(got[np.where(xs == 0.0)[0][1]])          # tanh(-0.0) = -0.0
    if fn == "qn_debug_tanh":
        z = torch.tensor([np.nan], device="cuda", dtype=torch.float64)
        w = torch.empty_like(z)
        _lib.check(L.qn_debug_tanh(z.data_ptr(), w.data_ptr(), 1, None), "qn_debug_tanh")
        assert torch.isnan(w).all()
    # against torch's CPU tanh (what the reference computes with): a few ulp at most
    tref = torch.tanh(torch.tensor(xs)).numpy()
    fin = np.isfinite(xs)
    assert np.max(np.abs(got[fin] - tref[fin]) / ulp[fin]) < 4.0


@pytest.mark.parametrize("nansafe", [1, 0])
def test_table_assisted_tanh_ulp_error(nansafe):
    """The tanh of the fused kernels (qn_tanh_f64_tab): table of tanh(n/16) in LDS + odd polynomial for the rest."""
    rs = np.random.RandomState(1)
    grid = np.arange(0, 321) / 16.0                                      # the table nodes and the bin edges around them
    xs = np.concatenate([rs.uniform(-20, 20, 200000), rs.uniform(-1, 1, 200000), rs.uniform(-1e-3, 1e-3, 50000),
                         10.0 ** rs.uniform(-300, -3, 20000), grid, -grid, grid + 1 / 32, grid[1:] - 1 / 32,
                         np.nextafter(grid + 1 / 32, 0), np.nextafter(grid + 1 / 32, 40),
                         np.array([0.0, -0.0, 19.0, 19.07, 25.0, -40.0, 1e300, -1e300, np.inf, -np.inf, 5e-324])])
    x = torch.tensor(xs, device="cuda")
    y = torch.empty_like(x)
    L = _lib.lib()
    _lib.check(L.qn_debug_tanh_table(x.data_ptr(), y.data_ptr(), x.numel(), nansafe, None), "qn_debug_tanh_table")
    torch.cuda.synchronize()
    got = y.cpu().numpy()
    ref = _ref_tanh(xs)
    ulp = np.spacing(np.abs(ref.astype(np.float64)))
    err = np.abs(got.astype(np.longdouble) - ref) / np.maximum(ulp, 5e-324)
    assert err.max() < 3.0, (err.max(), xs[np.argmax(err)])
    assert np.mean(err) < 0.7
    assert got[np.where(xs == np.inf)[0][0]] == 1.0 and got[np.where(xs == -np.inf)[0][0]] == -1.0
    assert np.signbit(got[np.where(xs == 0.0)[0][-1]])                  # tanh(-0.0) = -0.0
    assert (np.abs(got) <= 1.0).all() and (np.sign(got) == np.sign(xs)).all()
    if nansafe:
        z = torch.tensor([np.nan, 1.0], device="cuda", dtype=torch.float64)
        w = torch.empty_like(z)
        _lib.check(L.qn_debug_tanh_table(z.data_ptr(), w.data_ptr(), 2, 1, None), "qn_debug_tanh_table")
        assert torch.isnan(w[0]) and abs(w[1].item() - np.tanh(1.0)) < 1e-15
    # monotone on a fine grid across many table bins
    t = torch.linspace(0.0, 4.0, 200001, dtype=torch.float64, device="cuda")
    u = torch.empty_like(t)
    _lib.check(L.qn_debug_tanh_table(t.data_ptr(), u.data_ptr(), t.numel(), nansafe, None), "qn_debug_tanh_table")
    d = (u[1:] - u[:-1]).cpu().numpy()
    assert (d >= -2.3e-16).all()                                         # never decreases by more than ~1 ulp


def test_absolute_accuracy_tanh_of_the_int8_slice_kernel():
    """qn_tanh_f64_tab64 (table of tanh(n/64), no residual correction): absolute error far below the 2^-47 half step to
    which the int8-slice forward rounds its activations; exact at 0 and +-inf, odd, bounded by 1."""
    rs = np.random.RandomState(3)
    grid = np.arange(0, 1281) / 64.0
    xs = np.concatenate([rs.uniform(-20, 20, 300000), rs.uniform(-1, 1, 200000), rs.uniform(-1e-3, 1e-3, 50000),
                         10.0 ** rs.uniform(-300, -3, 20000), grid, -grid, grid + 1 / 128, np.nextafter(grid + 1 / 128, 40),
                         np.array([0.0, -0.0, 19.0, 19.07, 25.0, -40.0, 1e300, -1e300, np.inf, -np.inf, 5e-324])])
    x = torch.tensor(xs, device="cuda")
    y = torch.empty_like(x)
    L = _lib.lib()
    _lib.check(L.qn_debug_tanh_table(x.data_ptr(), y.data_ptr(), x.numel(), 2, None), "qn_debug_tanh_table")
    torch.cuda.synchronize()
    got = y.cpu().numpy()
    ref = _ref_tanh(xs)
    err = np.abs(got.astype(np.longdouble) - ref).astype(np.float64)
    assert err.max() < 2.0 ** -51, (err.max(), xs[np.argmax(err)])
    # (the contract is ABSOLUTE: the polynomial's truncation is 2^-46 relative to a small argument, ~100 ulp of the result)
    big = np.abs(ref) > 0.5
    ulp = np.spacing(np.abs(ref.astype(np.float64)))
    assert np.max(err[big] / ulp[big]) < 4.0
    assert got[np.where(xs == np.inf)[0][0]] == 1.0 and got[np.where(xs == -np.inf)[0][0]] == -1.0
    assert (np.abs(got) <= 1.0).all() and (np.sign(got) == np.sign(xs)).all()


def test_float32_tanh_is_relatively_accurate_from_1e_minus_30_to_saturation():
    """The float32 kernels' tanh through a (1, 1, 1) network with unit weights: relative error <= 1e-6 over 30 decades
    (the exp-based form alone is only absolutely accurate: 100 % error below 1e-7; tests/fuzz_all.py found it)."""
    from quinn_amd.ops import BatchedMLP, MLPArch
    mag = np.logspace(-30, 1.2, 4000)
    x = np.concatenate([mag, -mag, [0.0, 0.2999, 0.3, 0.3001]])[:, None].astype(np.float32).astype(np.float64)
    op = BatchedMLP(MLPArch((1, 1, 1), "tanh", bias=False), x, np.zeros_like(x), dtype="float32")
    got = op.predict(np.ones((1, 2))).double().cpu().numpy().reshape(-1)
    ref = np.tanh(x.reshape(-1))
    nz = ref != 0
    assert np.abs(got[nz] / ref[nz] - 1).max() <= 1e-6
    assert np.all(got[~nz] == 0)
