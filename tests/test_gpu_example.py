"""GPU: the reference's ex_ufit.py call pattern (examples/ex_ufit.py) runs for every accelerated
solver and returns finite predictive moments."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


@pytest.mark.parametrize("mlp", [False, True], ids=["rnet", "mlp"])
@pytest.mark.parametrize("meth", ["amcmc", "hmc", "vi", "ens", "rms"])
def test_ex_ufit_call_pattern(meth, mlp):
    import ex_ufit
    torch.manual_seed(0)
    np.random.seed(0)
    old = torch.get_default_dtype()
    try:
        ymean, ystd, rmse = ex_ufit.main(meth, quick=True, mlp=mlp)
    finally:
        torch.set_default_dtype(old)
    assert ymean.shape == (11,) and np.isfinite(ymean).all() and np.isfinite(ystd).all() and np.isfinite(rmse)
    if meth not in ("ens", "rms"):
        assert ystd.max() > 0.0
