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


def test_plot_helpers_write_the_reference_file_names(tmp_path, monkeypatch):
    """`plot_1d_fits` / `predict_plot` (quinn.py:106-260), the last two calls of examples/ex_ufit.py:143-145."""
    pytest.importorskip("matplotlib")
    from quinn_amd.nns.mlp import MLP
    from quinn_amd.solvers.nn_ens import NN_Ens
    monkeypatch.chdir(tmp_path)
    rs = np.random.RandomState(0)
    x = rs.rand(20, 1) * 2 - 1
    y = np.sin(3 * x)
    ens = NN_Ens(MLP(1, 1, (8,), activ='tanh'), nens=3)
    ens.fit(x, y, nepochs=30, lrate=0.01, freq_out=1000)
    ens.plot_1d_fits([x[:15], x[15:]], [y[:15], y[15:]], nmc=3, labels=['Training', 'Validation'],
                     true_model=lambda xx, noise: np.sin(3 * xx), name_postfix='ens')
    ens.predict_plot([x[:15], x[15:]], [y[:15], y[15:]], nmc=3, plot_qt=False, labels=['Training', 'Validation'])
    assert (tmp_path / "fit_d0_o0_ens.png").stat().st_size > 0
    assert (tmp_path / "fitdiag_o0.png").stat().st_size > 0
