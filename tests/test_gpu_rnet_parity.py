"""GPU: the residual network of the reference (quinn/nns/rnet.py) through the HIP path -- the
operator against the reference's fixtures and the live oracle, and the solvers (NN_MCMC / NN_Ens /
NN_VI) on the network of examples/ex_ufit.py against the reference's own runs.
float64: rtol 1e-11 on SSE / log-posterior, 1e-10 (of max |grad|) on gradients; float32: 2e-4 / 2e-3."""
import numpy as np
import pytest
import torch

from conftest import load_golden, spec_of
from oracle import mlp_ref, mcmc_ref
from oracle.rnet_ref import RNetSpec
from quinn_amd.nns import rnet as R
from quinn_amd.nns.nnfit import load_flat_into
from quinn_amd.ops import MLPArch, RNetArch, BatchedMLP, neg_log_post_from_sse
from quinn_amd.solvers.nn_ens import NN_Ens
from quinn_amd.solvers.nn_mcmc import NN_MCMC
from quinn_amd.solvers.nn_vi import NN_VI

pytestmark = pytest.mark.gpu

TOL = {"float64": (1e-11, 1e-10), "float32": (2e-4, 2e-3)}


@pytest.fixture(autouse=True)
def _double_default():
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.double)              # examples/ex_ufit.py:25
    yield
    torch.set_default_dtype(old)


def _net_from_spec(s):
    wp = {"const": R.Const, "lin": R.Lin, "quad": R.Quad, "cubic": R.Cubic}.get(s.wp_kind)
    wp = wp() if wp else R.Poly(s.wp_arg) if s.wp_kind == "poly" else (R.NonPar(s.wp_arg) if s.wp_arg else None)
    return R.RNet(s.rdim, s.nlayers, wp_function=wp, indim=s.indim or None, outdim=s.outdim or None,
                  biasorno=s.bias, nonlin=s.nonlin, mlp=s.mlp, layer_pre=s.layer_pre, layer_post=s.layer_post)


def _net(g):
    return _net_from_spec(spec_of(g))


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("ci", range(6))
def test_g10_golden_logpost_grad_pred(ci, dtype):
    g = load_golden(f"g10_rnet_logpost_{ci}.npz")
    arch = MLPArch.from_module(_net(g))
    assert isinstance(arch, RNetArch)
    op = BatchedMLP(arch, g["x"], g["y"], dtype=dtype)
    rt, gt = TOL[dtype]
    n, sigma = g["x"].shape[0], float(g["sigma"])
    sse, grad = op.sse_grad(g["W"])
    sse2, pred = op.sse_pred(g["W"])
    lp = -neg_log_post_from_sse(sse.cpu().numpy(), n, sigma)
    np.testing.assert_allclose(lp, g["logpost"], rtol=rt)
    np.testing.assert_allclose(sse2.cpu().numpy(), sse.cpu().numpy(), rtol=rt)
    gl = -(0.5 * grad.double().cpu().numpy() / sigma ** 2)
    scale = np.abs(g["grad"]).max(axis=1, keepdims=True)
    assert np.max(np.abs(gl - g["grad"]) / scale) < gt
    np.testing.assert_allclose(pred.double().cpu().numpy(), g["pred"], rtol=gt, atol=gt)


SPECS = [  # spec, N, B
    (RNetSpec(64, 2, "nonpar", 0, 3, 2, layer_pre=True, layer_post=True), 700, 3),     # wide, ragged rows
    (RNetSpec(20, 7, "poly", 3, 1, 1, layer_pre=True, layer_post=True), 1000, 5),
    (RNetSpec(9, 15, "nonpar", 0), 257, 2),                                            # 16 steps, d = r = o
    (RNetSpec(3, 3, "poly", 0, 1, 1, layer_pre=True, layer_post=True), 1, 4),          # one data row
    (RNetSpec(7, 1, "const", 0, 7, 2, layer_post=True, bias=False), 90, 2),            # post only
    (RNetSpec(5, 2, "lin", 0, 2, 5, layer_pre=True, mlp=True), 64, 3),                 # pre only, plain layers
]


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("case", SPECS, ids=[f"r{c[0].rdim}L{c[0].nlayers}{c[0].wp_kind}" for c in SPECS])
def test_random_rnets_vs_oracle(case, dtype):
    spec, N, B = case
    rs = np.random.RandomState(spec.rdim * 31 + spec.nlayers)
    x = rs.uniform(-2, 2, (N, spec.d))
    y = rs.randn(N, spec.o)
    W = 0.4 * rs.randn(B, spec.nparams)
    arch = MLPArch.from_module(_net_from_spec(spec))
    assert arch.nparams == spec.nparams
    op = BatchedMLP(arch, x, y, dtype=dtype)
    mod = mlp_ref.build_module(spec)
    rt, gt = TOL[dtype]
    sse, grad = op.sse_grad(W)
    _, pred = op.sse_pred(W)
    for b in range(B):
        ref_g = -mlp_ref.logpostgrad(mod, W[b], x, [v for v in y], 1.0) * 2.0      # d SSE / dw at sigma = 1
        np.testing.assert_allclose(float(sse[b]), mlp_ref.sse(mod, W[b], x, y), rtol=rt)
        assert np.max(np.abs(grad[b].double().cpu().numpy() - ref_g)) / np.abs(ref_g).max() < gt
        np.testing.assert_allclose(pred[b].double().cpu().numpy(), mlp_ref.forward_flat(mod, W[b], x), rtol=gt,
                                   atol=gt)


FUSED_SPECS = [  # small residual networks the single-launch kernels take (qn_rnet.hip): spec, N, B
    (RNetSpec(3, 3, "poly", 0, 1, 1, layer_pre=True, layer_post=True), 300, 5),        # examples/ex_ufit.py
    (RNetSpec(5, 3, "nonpar", 0, 2, 1, layer_pre=True, layer_post=True), 77, 3),
    (RNetSpec(3, 4, "lin", 0), 130, 2),                                               # no pre / post layer
    (RNetSpec(4, 7, "nonpar", 0, 2, 3, layer_pre=True, layer_post=True), 513, 2),      # 8 steps
    (RNetSpec(8, 2, "poly", 2, 4, 4, layer_pre=True, layer_post=True), 65, 4),         # widest: smaller blocks
    (RNetSpec(6, 2, "cubic", 0, 2, 2, layer_pre=True, layer_post=True, bias=False, nonlin=False), 40, 3),
    (RNetSpec(4, 3, "quad", 0, 2, 1, mlp=True, layer_pre=True, layer_post=True), 1, 2),  # one row, plain layers
    (RNetSpec(2, 1, "const", 0, 2, 4, layer_post=True), 33, 70),                       # post only, many chains
]


@pytest.mark.parametrize("use_idx", [False, True])
@pytest.mark.parametrize("case", FUSED_SPECS, ids=[f"r{c[0].rdim}L{c[0].nlayers}{c[0].wp_kind}" for c in FUSED_SPECS])
def test_single_launch_kernels_equal_layerwise_and_oracle(case, use_idx):
    from quinn_amd import _lib
    spec, N, B = case
    rs = np.random.RandomState(spec.rdim * 17 + spec.nlayers + N)
    x = rs.uniform(-2, 2, (N, spec.d))
    y = rs.randn(N, spec.o)
    W = 0.4 * rs.randn(B, spec.nparams)
    idx = rs.randint(0, N, size=(B, max(1, N // 2 + 1))).astype(np.int32) if use_idx else None
    op = BatchedMLP(MLPArch.from_module(_net_from_spec(spec)), x, y)
    L = _lib.lib()
    assert op.path(B, N, True) == _lib.PATH_FUSED
    res = {}
    for path in (_lib.PATH_GENERIC, _lib.PATH_FUSED):
        old = op.set_path(path)
        try:
            s, g = op.sse_grad(W, row_idx=idx)
            s2, pr = op.sse_pred(W, row_idx=idx)
        finally:
            op.set_path(old)
        res[path] = (s.cpu().numpy(), g.cpu().numpy(), s2.cpu().numpy(), pr.cpu().numpy())
    a, b = res[_lib.PATH_GENERIC], res[_lib.PATH_FUSED]
    np.testing.assert_allclose(b[0], a[0], rtol=1e-12)
    np.testing.assert_allclose(b[2], b[0], rtol=1e-13)
    assert np.abs(b[1] - a[1]).max() <= 1e-10 * max(np.abs(a[1]).max(), 1e-300)
    np.testing.assert_allclose(b[3], a[3], rtol=1e-11, atol=1e-12)
    if idx is None:
        mod = mlp_ref.build_module(spec)
        for k in range(min(B, 2)):
            np.testing.assert_allclose(b[0][k], mlp_ref.sse(mod, W[k], x, y), rtol=1e-11)
            ref_g = -mlp_ref.logpostgrad(mod, W[k], x, [v for v in y], 1.0) * 2.0
            assert np.max(np.abs(b[1][k] - ref_g)) / np.abs(ref_g).max() < 1e-10
    # run-to-run bitwise determinism of the fused gradient
    s3, g3 = op.sse_grad(W, row_idx=idx)
    assert np.array_equal(g3.cpu().numpy(), b[1]) and np.array_equal(s3.cpu().numpy(), b[0])


@pytest.mark.parametrize("spec", [RNetSpec(3, 3, "poly", 0, 1, 10, layer_pre=True, layer_post=True),      # ex_ufit.py with Sine10
                                  RNetSpec(5, 2, "nonpar", 0, 12, 7, layer_pre=True, layer_post=True),
                                  RNetSpec(8, 1, "lin", 0, 16, 16, layer_pre=True, layer_post=True)],
                         ids=["sine10", "d12o7", "d16o16"])
def test_single_launch_forward_with_up_to_16_inputs_and_outputs(spec):
    """More than 4 inputs / outputs (`Sine10`, examples/ex_ufit.py:54): the single-launch FORWARD kernel takes up to 16; the
    gradient of such a network runs on the layer-wise kernels."""
    from quinn_amd import _lib
    rs = np.random.RandomState(spec.d * 31 + spec.o)
    N, B = 210, 3
    x = rs.uniform(-2, 2, (N, spec.d))
    y = rs.randn(N, spec.o)
    W = 0.4 * rs.randn(B, spec.nparams)
    idx = rs.randint(0, N, size=(B, 64)).astype(np.int32)
    op = BatchedMLP(MLPArch.from_module(_net_from_spec(spec)), x, y)
    assert op.path(B, N, False) == _lib.PATH_FUSED and op.path(B, N, True) == _lib.PATH_GENERIC
    L = _lib.lib()
    res = {}
    for path in (_lib.PATH_GENERIC, _lib.PATH_AUTO):
        old = op.set_path(path)
        try:
            s1 = op.sse(W)
            s2, pr = op.sse_pred(W, row_idx=idx)
        finally:
            op.set_path(old)
        res[path] = (s1.cpu().numpy(), s2.cpu().numpy(), pr.cpu().numpy())
    a, b = res[_lib.PATH_GENERIC], res[_lib.PATH_AUTO]
    np.testing.assert_allclose(b[0], a[0], rtol=1e-12)
    np.testing.assert_allclose(b[1], a[1], rtol=1e-12)
    np.testing.assert_allclose(b[2], a[2], rtol=1e-11, atol=1e-12)
    mod = mlp_ref.build_module(spec)
    for k in range(2):
        np.testing.assert_allclose(b[0][k], mlp_ref.sse(mod, W[k], x, y), rtol=1e-11)
    s, g = op.sse_grad(W)                                   # layer-wise kernels
    ref_g = -mlp_ref.logpostgrad(mod, W[0], x, [v for v in y], 1.0) * 2.0
    assert np.max(np.abs(g[0].cpu().numpy() - ref_g)) / np.abs(ref_g).max() < 1e-10


def test_minibatch_rows_per_member():
    """row_idx gathers (the ensemble trainer's minibatches) on an RNet."""
    spec = RNetSpec(6, 2, "quad", 0, 2, 1, layer_pre=True, layer_post=True)
    rs = np.random.RandomState(5)
    x, y = rs.uniform(-2, 2, (50, 2)), rs.randn(50, 1)
    W = 0.4 * rs.randn(3, spec.nparams)
    rows = np.stack([rs.permutation(50)[:13] for _ in range(3)]).astype(np.int32)
    op = BatchedMLP(MLPArch.from_module(_net_from_spec(spec)), x, y)
    sse, grad = op.sse_grad(W, row_idx=rows)
    mod = mlp_ref.build_module(spec)
    for b in range(3):
        xb, yb = x[rows[b]], y[rows[b]]
        np.testing.assert_allclose(float(sse[b]), mlp_ref.sse(mod, W[b], xb, yb), rtol=1e-11)
        ref_g = -mlp_ref.logpostgrad(mod, W[b], xb, [v for v in yb], 1.0) * 2.0
        assert np.max(np.abs(grad[b].cpu().numpy() - ref_g)) / np.abs(ref_g).max() < 1e-10


def test_amcmc_chain_on_ex_ufit_network():
    g = load_golden("g10_rnet_amcmc.npz")
    solver = NN_MCMC(_net(g), verbose=False)
    np.random.seed(int(g["seed"]))
    solver.fit(g["x"], g["y"], zflag=False, datanoise=float(g["sigma"]), nmcmc=int(g["nmcmc"]), sampler='amcmc',
               sampler_params={'gamma': float(g["gamma"]), 't0': int(g["t0"]), 'tadapt': int(g["tadapt"])})
    acc = (solver.samples[1:] != solver.samples[:-1]).any(axis=1)
    assert np.array_equal(acc, (g["chain"][1:] != g["chain"][:-1]).any(axis=1))     # acceptance indices: exact
    assert solver.mcmc_results["accrate"] == float(g["accrate"])
    np.testing.assert_allclose(solver.samples, g["chain"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(solver.mcmc_results["logpost"], g["logpost"], rtol=1e-9)
    # bit-exact against the oracle stepping the same chain on this host (host LAPACK in the proposal)
    spec = spec_of(g)
    mod = mlp_ref.build_module(spec)
    yd = [v for v in g["y"]]
    rng = np.random.RandomState(int(g["seed"]))
    ini = rng.rand(spec.nparams)
    ref = mcmc_ref.run_chain(lambda w: mlp_ref.logpost(mod, w, g["x"], yd, float(g["sigma"])),
                             mcmc_ref.AmcmcState(gamma=float(g["gamma"]), t0=int(g["t0"]), tadapt=int(g["tadapt"])),
                             int(g["nmcmc"]), ini, rng)
    assert np.array_equal(solver.samples, ref["chain"])
    ymap = solver.predict_MAP(g["x"])
    np.testing.assert_allclose(ymap, mlp_ref.forward_flat(mod, ref["mapparams"], g["x"]), rtol=1e-10, atol=1e-12)


def test_hmc_chain_and_multichain():
    g = load_golden("g10_rnet_hmc.npz")
    solver = NN_MCMC(_net(g), verbose=False)
    sp = {'epsilon': float(g["epsilon"]), 'L': int(g["L"])}
    solver.fit(g["x"], g["y"], zflag=False, datanoise=float(g["sigma"]), nmcmc=int(g["nmcmc"]), sampler='hmc',
               sampler_params=sp, seeds=[int(g["seed"]), 5, 6])
    chain = solver.samples[0]
    acc = (chain[1:] != chain[:-1]).any(axis=1)
    assert np.array_equal(acc, (g["chain"][1:] != g["chain"][:-1]).any(axis=1))
    np.testing.assert_allclose(chain, g["chain"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(solver.mcmc_results["logpost"][0], g["logpost"], rtol=1e-9)
    assert solver.samples.shape == (3,) + g["chain"].shape


def test_ensemble_trajectories():
    g = load_golden("g10_rnet_ens.npz")
    net = _net(g)
    load_flat_into(net, g["w0"])
    ens = NN_Ens(net, nens=int(g["nens"]), dfrac=float(g["dfrac"]), verbose=False)
    np.random.seed(int(g["np_seed"]))
    torch.manual_seed(int(g["torch_seed"]))
    ens.fit(g["x"], g["y"], val=[g["xval"], g["yval"]], lrate=float(g["lrate"]), batch_size=int(g["batch_size"]),
            nepochs=int(g["nepochs"]), freq_out=1000)
    hist = np.array([l.history for l in ens.learners])
    np.testing.assert_allclose(hist, g["history"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(ens.fit_results["best_w"], g["best"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(ens.fit_results["final_w"], g["final"], rtol=1e-9, atol=1e-11)
    np.random.seed(int(g["predict_seed"]))
    np.testing.assert_allclose(ens.predict_ens(g["xg"]), g["yens"], rtol=1e-9, atol=1e-11)


def test_vi_fit_trajectory_and_init_order():
    g = load_golden("g10_rnet_vifit.npz")
    torch.manual_seed(int(g["torch_seed"]))
    net = _net(g)                                      # consumes the generator like the reference's RNet()
    from quinn_amd.ops import flatten_module
    assert np.array_equal(flatten_module(net), g["w_net"])
    vi = NN_VI(net, verbose=False)
    np.testing.assert_array_equal(vi.bmodel.mu.cpu().numpy(), g["mu0"])
    np.testing.assert_array_equal(vi.bmodel.rho.cpu().numpy(), g["rho0"])
    torch.set_rng_state(torch.from_numpy(g["gen_state"]))
    vi.fit(g["x"], g["y"], val=[g["xval"], g["yval"]], datanoise=float(g["datanoise"]), lrate=float(g["lrate"]),
           batch_size=int(g["batch_size"]), nsam=int(g["nsam"]), nepochs=int(g["nepochs"]), freq_out=1000)
    np.testing.assert_allclose(np.array(vi.fit_info["history"]), g["history"], rtol=1e-8, atol=1e-9)
    p = vi.bmodel.p
    th = vi.bmodel.theta.detach().cpu().numpy()
    np.testing.assert_allclose(th[:p], g["mu_final"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(th[p:], g["rho_final"], rtol=1e-8, atol=1e-10)
    assert vi.fit_info["best_epoch"] == int(g["best_epoch"])


def test_module_predict_and_fit_api():
    """RNet.predict / RNet.fit (nnbase.py:61-115; tests/test_mlp.py:159-170) run on the device operator."""
    torch.manual_seed(3)
    net = R.RNet(4, 3, indim=2, outdim=1, layer_pre=True, layer_post=True)
    x = np.random.RandomState(0).rand(10, 2)
    y = net.predict(x)
    assert isinstance(y, np.ndarray) and y.shape == (10, 1)
    with torch.no_grad():
        np.testing.assert_allclose(y, net(torch.from_numpy(x)).numpy(), rtol=1e-12, atol=1e-14)
    assert net.numpar() == 2 * 4 + 4 + 4 + 1 + 4 * 16 + 4 * 4


def test_descriptor_argument_checks():
    with pytest.raises(Exception):
        BatchedMLP(RNetArch(2, 3, 1, 2, ((1.0,), (1.0,)), layer_pre=False, layer_post=True), np.zeros((1, 2)), None)
    with pytest.raises(Exception):
        BatchedMLP(RNetArch(3, 3, 3, 17, tuple((1.0,) for _ in range(17))), np.zeros((1, 3)), None)


@pytest.mark.parametrize("sampler,sp", [("amcmc", {'gamma': 0.1, 't0': 50, 'tadapt': 100}), ("hmc", {'epsilon': 0.01, 'L': 3})])
def test_device_engines_run_on_the_ex_ufit_network(sampler, sp):
    """engine='device' (states, proposals and history on the GPU) with the residual network of examples/ex_ufit.py."""
    rs = np.random.RandomState(3)
    x = rs.rand(14, 1) * 2 * np.pi - np.pi
    y = np.sin(x) + 0.02 * rs.randn(14, 1)
    torch.manual_seed(5)
    solver = NN_MCMC(R.RNet(3, 3, wp_function=R.Poly(0), indim=1, outdim=1, layer_pre=True, layer_post=True), verbose=False)
    solver.fit(x, y, zflag=False, datanoise=0.2, nmcmc=400, sampler=sampler, sampler_params=sp, seeds=list(range(6)),
               engine='device')
    r = solver.mcmc_results
    assert r['chain'].shape == (6, 401, 22) and np.isfinite(r['logpost']).all()
    assert (np.asarray(r['accrate']) > 0.02).all()
    assert (r['logpost'][:, -100:].mean(axis=1) > r['logpost'][:, 0]).all()      # climbed from the random start
    assert solver.predict_ens(x, nens=5, nburn=100, chain=0).shape == (5, 14, 1)


EXC = [  # spec, where the value goes, value
    (RNetSpec(3, 0, "cubic", 0), "x", np.inf),                                           # no pre layer: out_0 = x (a copy, not I . x)
    (RNetSpec(2, 0, "const", 0, 0, 4, layer_post=True), "x", -np.inf),
    (RNetSpec(5, 2, "lin", 0), "x", np.inf),
    (RNetSpec(8, 5, "nonpar", 3, mlp=True), "ww1", -np.inf),                             # NonPar: a tensor only enters its own steps
    (RNetSpec(4, 15, "nonpar", 16, 5, 0, layer_pre=True), "ww9", np.nan),
    (RNetSpec(4, 0, "quad", 0, 4, 0, bias=False, nonlin=False, layer_pre=True), "ww2", np.nan),   # t = 0: the reference multiplies ww_2 by 0
    (RNetSpec(3, 2, "poly", 2, 0, 1, layer_post=True), "ww1", np.inf),
]


@pytest.mark.parametrize("case", EXC, ids=[f"r{c[0].rdim}L{c[0].nlayers}{c[0].wp_kind}_{c[1]}_{c[2]}" for c in EXC])
def test_not_finite_values_follow_the_reference_ops(case):
    """NaN / +Inf / -Inf pattern of SSE and predictions as torch's on both kernel families; gradient: finite entries in the
    same places and equal (an entry that is +-Inf in the reference may be NaN: DESIGN 4.2).  Found by tests/fuzz_all.py."""
    from quinn_amd import _lib
    spec, where, val = case
    rs = np.random.RandomState(7)
    N, B = 37, 2
    x = rs.uniform(-1, 1, (N, spec.d)); y = rs.randn(N, spec.o)
    W = 0.3 * rs.randn(B, spec.nparams)
    if where == "x":
        x[5, 0] = val
    else:                                                       # an entry of parameter tensor ww_k
        k = int(where[2:])
        off = sum(int(np.prod(s)) for s in spec.param_shapes()[:(2 if spec.layer_pre else 0) + (2 if spec.layer_post else 0) + k])
        W[0, off + 1] = val
    op = BatchedMLP(MLPArch.from_module(_net_from_spec(spec)), x, y)
    mod = mlp_ref.build_module(spec)
    cls = lambda v: np.where(np.isnan(v), 3, np.where(np.isposinf(v), 1, np.where(np.isneginf(v), 2, 0)))
    with np.errstate(all="ignore"):
        ref_s = np.array([mlp_ref.sse(mod, W[b], x, y) for b in range(B)])
        ref_p = np.stack([mlp_ref.forward_flat(mod, W[b], x) for b in range(B)])
        ref_g = np.stack([-2.0 * mlp_ref.logpostgrad(mod, W[b], x, [v for v in y], 1.0) for b in range(B)])
    for path in (_lib.PATH_AUTO, _lib.PATH_GENERIC):
        op.set_path(path)
        s, g = op.sse_grad(W)
        s2, pr = op.sse_pred(W)
        s, g, s2, pr = (t.cpu().numpy() for t in (s, g, s2, pr))
        assert np.array_equal(cls(s), cls(ref_s)) and np.array_equal(cls(s2), cls(ref_s))
        assert np.array_equal(cls(pr.reshape(ref_p.shape)), cls(ref_p))
        cg, cr = cls(g), cls(ref_g)
        cg = np.where((cg == 3) & ((cr == 1) | (cr == 2)), cr, cg)
        assert np.array_equal(cg, cr)
        fin = np.isfinite(ref_g)
        for b in range(B):
            if fin[b].any():
                assert np.abs(g[b][fin[b]] - ref_g[b][fin[b]]).max() <= 1e-9 * max(np.abs(ref_g[b][fin[b]]).max(), 1e-300)
        fp = np.isfinite(ref_p)
        np.testing.assert_allclose(pr.reshape(ref_p.shape)[fp], ref_p[fp], rtol=1e-10, atol=1e-12)
