"""CPU: the numpy <-> torch helpers keep the reference's contract (quinn/nns/tchutils.py:11-41; the reference's own
test of them is tests/test_nnwrap.py's round trips through tch/npy)."""
import numpy as np
import torch

from quinn_amd.nns.tchutils import npy, tch


def test_tch_copies_to_float64_and_npy_round_trips():
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    t = tch(a)
    assert t.dtype == torch.float64 and t.shape == (2, 3) and not t.requires_grad
    a[0, 0] = 99.0                                          # a copy, not a view of the caller's array
    assert t[0, 0].item() == 0.0
    assert np.array_equal(npy(t), np.arange(6, dtype=np.float64).reshape(2, 3))
    assert tch([1.0, 2.0]).dtype == torch.float64 and tch([[1, 2], [3, 4]]).dtype == torch.int64
    g = tch(np.ones(3), rgrad=True)
    assert g.requires_grad and npy((g * 2).sum() * torch.ones(2)).tolist() == [6.0, 6.0]
    assert torch.get_default_dtype() == torch.get_default_dtype()   # importing the module leaves the global default alone
