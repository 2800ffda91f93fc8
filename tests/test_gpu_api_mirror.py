"""GPU: the small API mirrors (NNWrap, nn_p, NegLogPost, MLP.fit / predict) behave like the
reference's (shapes / values against golden G1 and the reference's own test expectations)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from quinn_amd.nns.losses import NegLogPost
from quinn_amd.nns.mlp import MLP
from quinn_amd.nns.nnwrap import NNWrap, nn_p

pytestmark = pytest.mark.gpu


def _net(g):
    dims = [int(v) for v in g["dims"]]
    return MLP(dims[0], dims[-1], tuple(dims[1:-1]), activ=str(g["activ"]))


def test_nnwrap_loss_grad_predict_match_golden():
    g = load_golden("g1_logpost_3.npz")               # 2 inputs, 2 outputs
    net = _net(g)
    w = NNWrap(net)
    loss = NegLogPost(net, g["x"].shape[0], float(g["sigma"]), None)
    yd = [v for v in g["y"]]
    for k in range(3):
        val = w.calc_loss(g["W"][k], loss, g["x"], yd)
        assert isinstance(val, float) and abs(-val - g["logpost"][k]) <= 1e-11 * abs(g["logpost"][k])
        gr = w.calc_lossgrad(g["W"][k], loss, g["x"], yd)
        assert gr.shape == g["W"][k].shape                               # test_nnwrap.py:69-83
        np.testing.assert_allclose(-gr, g["grad"][k], rtol=1e-8, atol=1e-10 * np.abs(g["grad"][k]).max())
        np.testing.assert_allclose(nn_p(g["W"][k], g["x"], net), g["pred"][k], rtol=1e-11, atol=1e-12)
    # flatten -> unflatten round trip preserves predictions (test_nnwrap.py:22-38)
    flat = w.p_flatten().detach().numpy().reshape(-1)
    y0 = w(g["x"])
    w.p_unflatten(flat)
    assert np.array_equal(w(g["x"]), y0)
    with torch.no_grad():
        np.testing.assert_allclose(y0, net(torch.tensor(g["x"])).numpy(), rtol=1e-12, atol=1e-13)
    # NegLogPost.forward on the module's current weights; perfect target beats a bad one (test_losses.py)
    good = loss(torch.tensor(g["x"]), torch.tensor(y0)).item()
    bad = loss(torch.tensor(g["x"]), torch.tensor(y0 + 1.0)).item()
    assert np.isfinite(good) and good < bad
    # Gaussian prior term: anchor is the minimum over w of the prior part
    pri = NegLogPost(net, g["x"].shape[0], float(g["sigma"]), {'sigma': 0.5, 'anchor': torch.tensor(flat)})
    v0, g0 = pri.value_and_grad(flat, g["x"], y0, want_grad=True)
    v1, _ = pri.value_and_grad(flat + 0.1, g["x"], y0)
    assert v0 < v1 and g0.shape == flat.shape


def test_mlp_fit_and_predict():
    rs = np.random.RandomState(0)
    x = rs.rand(50, 1) * 2 - 1
    y = np.sin(3 * x)
    net = MLP(1, 1, (8, 8), activ='tanh')
    torch.manual_seed(0)
    best = net.fit(x, y, lrate=0.02, nepochs=150, batch_size=25, freq_out=1000)
    assert net.trained and len(net.history) == 300 and best is net.best_model
    assert net.history[-1][3] < net.history[0][3]                     # test_nnfit.py: loss decreases
    assert net.predict(x).shape == (50, 1)


def test_more_weight_vectors_than_one_launch_takes():
    """B > 65535 (grid limit of one launch) is split by the operator."""
    from quinn_amd.ops import MLPArch, BatchedMLP
    rs = np.random.RandomState(0)
    x, y = rs.rand(5, 1), rs.rand(5, 1)
    arch = MLPArch((1, 4, 1), "tanh")
    op = BatchedMLP(arch, x, y)
    W = rs.randn(70000, arch.nparams)
    s = op.sse(W).cpu().numpy()
    pick = [0, 65534, 65535, 65536, 69999]
    ref = op.sse(W[pick]).cpu().numpy()
    assert np.array_equal(s[pick], ref) and np.isfinite(s).all()


def test_empty_batch_and_argument_errors():
    from quinn_amd.ops import MLPArch, BatchedMLP
    from quinn_amd._lib import QuinnAmdError
    arch = MLPArch((2, 8, 1), "tanh")
    x, y = np.random.RandomState(0).rand(6, 2), np.zeros((6, 1))
    op = BatchedMLP(arch, x, y)
    s, g = op.sse_grad(np.zeros((0, arch.nparams)))
    assert s.shape == (0,) and g.shape == (0, arch.nparams)
    assert op.predict(np.zeros((0, arch.nparams))).shape == (0, 6, 1)
    with pytest.raises(ValueError):
        op.sse(np.zeros((3, arch.nparams + 1)))                  # wrong parameter count
    with pytest.raises((QuinnAmdError, ValueError, RuntimeError)):
        BatchedMLP(arch, np.zeros((0, 2)), np.zeros((0, 1))).sse(np.zeros((1, arch.nparams)))   # no data rows
