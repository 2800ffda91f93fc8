"""GPU: the reference's ex_ufit.py call pattern (examples/ex_ufit.py) runs for every accelerated
solver and returns finite predictive moments."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


@pytest.mark.parametrize("meth", ["amcmc", "hmc", "vi", "ens"])
def test_ex_ufit_call_pattern(meth):
    import ex_ufit
    torch.manual_seed(0)
    np.random.seed(0)
    ymean, ystd, rmse = ex_ufit.main(meth, quick=True)
    assert ymean.shape == (11,) and np.isfinite(ymean).all() and np.isfinite(ystd).all() and np.isfinite(rmse)
    if meth != "ens":
        assert ystd.max() > 0.0
