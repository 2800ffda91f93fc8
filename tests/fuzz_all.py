#!/usr/bin/env python3
"""Randomised parity sweep of the whole MLP operator (QN_PATH_AUTO: whichever kernel family the dispatcher picks) against
the oracle (oracle/mlp_ref.py: the reference's torch float64 module + autograd): random depths, uniform / ragged / odd
widths, 1..16 inputs, 1..4 outputs, every activation, bias on / off, row subsets, weight scales.  Test infrastructure
(imports oracle/).  usage: tests/fuzz_all.py [ncases] [seed]"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mlp_ref
from quinn_amd import _lib
from quinn_amd.ops import MLPArch, BatchedMLP
PATHS = {_lib.PATH_GENERIC: "generic", _lib.PATH_FUSED: "fused", _lib.PATH_FUSED_DP: "fused_dp"}


def _sensitivity(mod, w, xb, yb, rs):
    """How much the oracle's own SSE / gradient / predictions move when every weight changes by one unit in the last place
    (relative, max norm): the floor below which two correct float64 implementations cannot be told apart.  Deep saturated
    networks amplify rounding by 1e6 and more.  Measured PER CHAIN.  The bars of the sweeps are max(stated bar, K x this):
    K = 20 for the residual networks (plain float64 kernels), K = K_SLICED = 512 for the MLP operator, whose default
    (QN_PATH_AUTO) kernels for 64 / 128 / 256-wide tanh networks round operands to 2^-47 of their row / activation scale --
    64 units in the last place of float64 (2^-47 / 2^-53) -- times 8 for the accumulation over the layers."""
    w2 = w * (1.0 + 2.0 ** -52 * rs.choice([-1.0, 1.0], size=w.shape))
    with np.errstate(all="ignore"):
        s0, s1 = mlp_ref.sse(mod, w, xb, yb), mlp_ref.sse(mod, w2, xb, yb)
        p0, p1 = mlp_ref.forward_flat(mod, w, xb), mlp_ref.forward_flat(mod, w2, xb)
        g0 = mlp_ref.logpostgrad(mod, w, xb, [v for v in yb], 1.0); g1 = mlp_ref.logpostgrad(mod, w2, xb, [v for v in yb], 1.0)
    rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(a).max(), 1e-300))
    return abs(s1 / s0 - 1) if s0 else 0.0, rel(g0, g1), rel(p0, p1)


K_SLICED = 512          # 64 ulp (47-bit operands) x 8: see _sensitivity


def run(ncases=100, seed=0, verbose=True):
    """Returns (number of failed cases, worst [sse, grad, pred] errors)."""
    rs = np.random.RandomState(seed)
    worst = [0.0, 0.0, 0.0]; nfail = 0
    for case in range(ncases):
        nhid = int(rs.randint(1, 6))
        if rs.rand() < 0.6:
            hid = (int(rs.choice([3, 17, 32, 64, 64, 65, 100, 128, 200, 256])),) * nhid
        else:
            hid = tuple(int(v) for v in rs.randint(1, 90, size=nhid))
        d = int(rs.choice([1, 1, 2, 3, 4, 5, 8, 16])); o = int(rs.choice([1, 1, 1, 2, 3, 4, 6, 12]))
        act = str(rs.choice(sorted(_lib.ACT_CODES))); bias = bool(rs.rand() < 0.75)
        N = int(rs.choice([rs.randint(1, 40), rs.randint(40, 700)])); B = int(rs.choice([1, 2, rs.randint(3, 13)]))
        if max(hid) > 128: N = min(N, 200)
        dims = (d,) + hid + (o,)
        arch = MLPArch(dims, act, bias=bias)
        wscale = float(rs.choice([0.01, 0.3, 1.0, 3.0]))
        x = rs.rand(N, d) * 2 - 1; y = rs.randn(N, o)
        parts = []
        for a_, b_ in zip(dims[:-1], dims[1:]):
            parts.append(wscale * rs.randn(B, b_ * a_) / np.sqrt(a_))
            if bias: parts.append(wscale * rs.randn(B, b_))
        W = np.concatenate(parts, axis=1)
        idx = rs.randint(0, N, size=(B, int(rs.randint(1, N + 1)))) if rs.rand() < 0.3 else None
        dtype = "float32" if rs.rand() < 0.25 else "float64"
        op = BatchedMLP(arch, x, y, dtype=dtype)
        if B > 2 and rs.rand() < 0.25:              # a workspace cap that forces the call into chunks of weight vectors
            op.max_ws = op.workspace_bytes(max(1, B // 3), N if idx is None else idx.shape[1], True)
        s, g = op.sse_grad(W, row_idx=idx); s2, pr = op.sse_pred(W, row_idx=idx)
        s, g, s2, pr = (t.double().cpu().numpy() for t in (s, g, s2, pr))
        Nb = N if idx is None else idx.shape[1]
        pth = PATHS.get(op.path(B, Nb, True), "?") + "/" + PATHS.get(op.path(B, Nb, False), "?")
        mod = mlp_ref.build_module(mlp_ref.MLPSpec(dims, act, bias))
        e = [0.0, 0.0, 0.0]
        ts, tg = (1e-11, 1e-10) if dtype == "float64" else (2e-4, 2e-3)      # (float32: the bars of tests/test_gpu_rnet_parity.py)
        ok = True
        for b in range(B):
            xb, yb = (x, y) if idx is None else (x[idx[b]], y[idx[b]])
            sref = mlp_ref.sse(mod, W[b], xb, yb)
            pref = mlp_ref.forward_flat(mod, W[b], xb)
            gref = -2.0 * mlp_ref.logpostgrad(mod, W[b], xb, [v for v in yb], 1.0)          # dSSE/dw
            eb = [max(abs(s[b] / sref - 1), abs(s2[b] / sref - 1)), np.abs(g[b] - gref).max() / max(np.abs(gref).max(), 1e-300),
                  np.abs(pr[b].reshape(pref.shape) - pref).max() / max(np.abs(pref).max(), 1e-300)]
            tp = ts if dtype == "float64" else tg
            if not (eb[0] <= ts and eb[1] <= tg and eb[2] <= tp):            # beyond the fixed bars: this chain's own floor decides
                sens = _sensitivity(mod, W[b], xb, yb, rs)
                ok &= eb[0] <= max(ts, K_SLICED * sens[0]) and eb[1] <= max(tg, K_SLICED * sens[1]) and eb[2] <= max(tp, K_SLICED * sens[2])
            e = [max(u, v) for u, v in zip(e, eb)]
        nfail += not ok
        if dtype == "float64": worst = [max(u, v) for u, v in zip(worst, e)]
        if verbose or not ok:
            print(("ok  " if ok else "FAIL"), dims, act, "N", N, "B", B, "bias", bias, "rows", None if idx is None else idx.shape[1], "wscale", wscale, dtype, pth,
              "| sse %.1e grad %.1e pred %.1e" % tuple(e), flush=True)
        del op
    if verbose:
        print("worst: sse %.2e grad %.2e pred %.2e; %d of %d failed" % (*worst, nfail, ncases))
    return nfail, worst


def _random_rnet(rs):
    """A random residual network of the reference's family: (oracle spec, operator architecture)."""
    from oracle.rnet_ref import RNetSpec
    from quinn_amd.nns import rnet as R
    r = int(rs.choice([1, 2, 3, 3, 4, 5, 8, 9, 20, 33, 64, 70])); nl = int(rs.choice([0, 1, 2, 3, 5, 7, 15]))
    kind = str(rs.choice(["const", "lin", "quad", "cubic", "poly", "nonpar"]))
    arg = int(rs.randint(0, 4)) if kind == "poly" else (int(rs.choice([0, rs.randint(1, nl + 2)])) if kind == "nonpar" else 0)
    pre, post = bool(rs.rand() < 0.6), bool(rs.rand() < 0.6)
    spec = RNetSpec(r, nl, kind, arg, int(rs.randint(1, 6)) if pre else 0, int(rs.randint(1, 5)) if post else 0,
                    bias=bool(rs.rand() < 0.8), nonlin=bool(rs.rand() < 0.8), mlp=bool(rs.rand() < 0.25), layer_pre=pre, layer_post=post)
    wp = {"const": R.Const, "lin": R.Lin, "quad": R.Quad, "cubic": R.Cubic}.get(spec.wp_kind)
    wp = wp() if wp else R.Poly(spec.wp_arg) if spec.wp_kind == "poly" else (R.NonPar(spec.wp_arg) if spec.wp_arg else None)
    net = R.RNet(spec.rdim, spec.nlayers, wp_function=wp, indim=spec.indim or None, outdim=spec.outdim or None, biasorno=spec.bias,
                 nonlin=spec.nonlin, mlp=spec.mlp, layer_pre=spec.layer_pre, layer_post=spec.layer_post)
    arch = MLPArch.from_module(net)
    assert arch.nparams == spec.nparams
    return spec, arch, net


def run_rnet(ncases=60, seed=0, verbose=True):
    """The residual networks of the reference (quinn/nns/rnet.py) the same way: random width, depth, weight
    parameterisation, pre / post layers, bias, plain-layer mode."""
    rs = np.random.RandomState(seed)
    old_dt = torch.get_default_dtype()
    torch.set_default_dtype(torch.double)
    worst = [0.0, 0.0, 0.0]; nfail = 0
    try:
        for case in range(ncases):
            spec, arch, _ = _random_rnet(rs)
            N = int(rs.choice([rs.randint(1, 40), rs.randint(40, 600)])); B = int(rs.choice([1, 2, rs.randint(3, 40)]))
            x = rs.uniform(-2, 2, (N, spec.d)); y = rs.randn(N, spec.o)
            wscale = float(rs.choice([0.05, 0.4, 1.0]))
            W = wscale * rs.randn(B, spec.nparams)
            idx = rs.randint(0, N, size=(B, int(rs.randint(1, N + 1)))) if rs.rand() < 0.3 else None
            op = BatchedMLP(arch, x, y)
            s, g = op.sse_grad(W, row_idx=idx); s2, pr = op.sse_pred(W, row_idx=idx)
            s, g, s2, pr = (t.double().cpu().numpy() for t in (s, g, s2, pr))
            mod = mlp_ref.build_module(spec)
            e = [0.0, 0.0, 0.0]
            ok = True
            for b in range(B):
                xb, yb = (x, y) if idx is None else (x[idx[b]], y[idx[b]])
                sref = mlp_ref.sse(mod, W[b], xb, yb)
                pref = mlp_ref.forward_flat(mod, W[b], xb)
                gref = -2.0 * mlp_ref.logpostgrad(mod, W[b], xb, [v for v in yb], 1.0)
                eb = [max(abs(s[b] / sref - 1), abs(s2[b] / sref - 1)), np.abs(g[b] - gref).max() / max(np.abs(gref).max(), 1e-300),
                      np.abs(pr[b].reshape(pref.shape) - pref).max() / max(np.abs(pref).max(), 1e-300)]
                if not (eb[0] <= 1e-11 and eb[1] <= 1e-10 and eb[2] <= 1e-11):   # beyond the fixed bars: this chain's own floor decides
                    sens = _sensitivity(mod, W[b], xb, yb, rs)
                    ok &= eb[0] <= max(1e-11, 20 * sens[0]) and eb[1] <= max(1e-10, 20 * sens[1]) and eb[2] <= max(1e-11, 20 * sens[2])
                e = [max(u, v) for u, v in zip(e, eb)]
            nfail += not ok
            worst = [max(u, v) for u, v in zip(worst, e)]
            Nb = N if idx is None else idx.shape[1]
            if verbose or not ok:
                print(("ok  " if ok else "FAIL"), spec, "N", N, "B", B, "rows", None if idx is None else idx.shape[1],
                      PATHS.get(op.path(B, Nb, True), "?"), "| sse %.1e grad %.1e pred %.1e" % tuple(e), flush=True)
            del op
    finally:
        torch.set_default_dtype(old_dt)
    if verbose:
        print("rnet worst: sse %.2e grad %.2e pred %.2e; %d of %d failed" % (*worst, nfail, ncases))
    return nfail, worst


def run_mcmc(ncases=12, seed=0, verbose=True):
    """NN_MCMC (host engine: the reference's random-number stream) against the oracle's sequential chains
    (oracle/mcmc_ref.py = quinn/mcmc): random small networks, data sizes, noise levels, sampler settings, chain counts.
    AMCMC: chains and acceptance indices bit for bit; HMC / MALA: acceptance indices equal, states to 1e-8."""
    from oracle import mcmc_ref
    from quinn_amd.nns.mlp import MLP
    from quinn_amd.solvers.nn_mcmc import NN_MCMC
    rs = np.random.RandomState(seed)
    old_dt = torch.get_default_dtype()
    torch.set_default_dtype(torch.double)
    nfail = 0
    try:
        for case in range(ncases):
            d = int(rs.choice([1, 2, 3, 3, 6, 8])); o = int(rs.choice([1, 1, 2])); hid = tuple(int(v) for v in rs.choice([2, 3, 5, 8, 16, 64, 70], size=rs.randint(1, 4)))
            act = str(rs.choice(["tanh", "relu"])); N = int(rs.randint(5, 300)); sigma = float(rs.choice([0.05, 0.2, 1.0]))
            sampler = str(rs.choice(["amcmc", "amcmc", "hmc", "mala"])); C = int(rs.randint(1, 5)); nmcmc = int(rs.randint(40, 160))
            seeds = [int(v) for v in rs.randint(0, 10000, size=C)]
            x = rs.rand(N, d) * 4 - 2; y = np.sin(x.sum(axis=1, keepdims=True)) * np.ones((1, o)) + sigma * rs.randn(N, o)
            dims = (d,) + hid + (o,)
            spec = mlp_ref.MLPSpec(dims, act)
            if spec.nparams > (250 if sampler == "amcmc" else 3000):      # (the reference's proposal is an SVD of p x p per step: seconds per case at p = 250, minutes at 500)
                hid = hid[:1]; dims = (d,) + hid + (o,); spec = mlp_ref.MLPSpec(dims, act)
            net = None
            if rs.rand() < 0.3:                                            # a residual network (the model of examples/ex_ufit.py)
                for _ in range(20):
                    rspec, _, rnet_ = _random_rnet(rs)
                    if rspec.nparams <= 400: break
                if rspec.nparams <= 400:
                    spec, net, d, o, dims, act = rspec, rnet_, rspec.d, rspec.o, rspec, rspec.activ
                    x = rs.rand(N, d) * 4 - 2; y = np.sin(x.sum(axis=1, keepdims=True)) * np.ones((1, o)) + sigma * rs.randn(N, o)
            yd = [v for v in y]
            if sampler == "amcmc":
                sp = {'gamma': float(rs.choice([0.01, 0.1, 0.5])), 't0': int(rs.randint(3, 40)), 'tadapt': int(rs.randint(2, 30))}
                mk = lambda: mcmc_ref.AmcmcState(gamma=sp['gamma'], t0=sp['t0'], tadapt=sp['tadapt'])
            elif sampler == "hmc":
                sp = {'epsilon': float(rs.choice([1e-3, 5e-3, 2e-2])) * sigma, 'L': int(rs.randint(1, 6))}
                mk = lambda: mcmc_ref.HmcState(epsilon=sp['epsilon'], L=sp['L'])
            else:
                sp = {'epsilon': float(rs.choice([1e-3, 5e-3, 2e-2])) * sigma}
                mk = lambda: mcmc_ref.MalaState(epsilon=sp['epsilon'])
            mods = []
            def mkmod():
                mods.append(mlp_ref.build_module(spec)); return mods[-1]
            ref = mcmc_ref.run_multichain(lambda: (lambda w, m=mkmod(): mlp_ref.logpost(m, w, x, yd, sigma)), mk, nmcmc, spec.nparams, seeds,
                                          make_logpostgrad=None if sampler == "amcmc" else (lambda: (lambda w, m=mkmod(): mlp_ref.logpostgrad(m, w, x, yd, sigma))))
            solver = NN_MCMC(net if net is not None else MLP(d, o, hid, activ=act), verbose=False)
            solver.fit(x, y, zflag=False, datanoise=sigma, nmcmc=nmcmc, sampler=sampler, sampler_params=dict(sp), seeds=seeds)
            chain = np.asarray(solver.samples).reshape(C, nmcmc + 1, -1)
            acc = (chain[:, 1:] != chain[:, :-1]).any(axis=2)
            ok = np.array_equal(acc, ref["accepted"])
            if sampler == "amcmc":
                ok = ok and np.array_equal(chain, ref["chain"])
            else:
                ok = ok and np.allclose(chain, ref["chain"], rtol=1e-8, atol=1e-8)
            lp = np.asarray(solver.mcmc_results["logpost"]).reshape(C, -1)
            ok = ok and np.allclose(lp, ref["logpost"], rtol=1e-9)
            nfail += not ok
            if verbose or not ok:
                print(("ok  " if ok else "FAIL"), dims, act, "N", N, "sigma", sigma, sampler, sp, "chains", C, "steps", nmcmc,
                      "| accept %.2f  max |dchain| %.1e" % (acc.mean(), np.abs(chain - ref["chain"]).max()), flush=True)
    finally:
        torch.set_default_dtype(old_dt)
    if verbose:
        print("mcmc: %d of %d failed" % (nfail, ncases))
    return nfail


def _run_exceptional(ncases, seed, verbose, paths):
    """Not-finite / huge / denormal values at random places (a weight, a bias, an input, a target) of random networks:
    the kernels the dispatcher picks (QN_PATH_AUTO: fused, int8-slice, ...) and the layer-wise kernels (QN_PATH_GENERIC) must
    follow the IEEE semantics of the reference's torch ops (the oracle): same NaN / +Inf / -Inf pattern in SSE, predictions
    and gradient, finite values equal to 1e-11 / 1e-9."""
    rs = np.random.RandomState(seed)
    nfail = 0
    codes = {"auto": _lib.PATH_AUTO, "generic": _lib.PATH_GENERIC}
    cls = lambda v: np.where(np.isnan(v), 3, np.where(np.isposinf(v), 1, np.where(np.isneginf(v), 2, 0)))
    for case in range(ncases):
        N = int(rs.randint(1, 400)); B = int(rs.randint(1, 6))
        if rs.rand() < 0.25:
            spec, arch, _ = _random_rnet(rs)
            dims, act, bias, d, o, h = spec, spec.activ, spec.bias, spec.d, spec.o, spec.rdim
        else:
            h = int(rs.choice([8, 33, 64, 64, 128, 256])); nhid = int(rs.randint(1, 5)); d = int(rs.choice([1, 2, 4, 6, 12])); o = int(rs.choice([1, 1, 2, 7]))
            act = str(rs.choice(["tanh", "tanh", "relu", "identity"])); bias = bool(rs.rand() < 0.8)
            dims = (d,) + (h,) * nhid + (o,)
            arch = MLPArch(dims, act, bias=bias)
            spec = mlp_ref.MLPSpec(dims, act, bias)
        x = rs.rand(N, d) * 2 - 1; y = rs.randn(N, o)
        W = 0.5 * rs.randn(B, arch.nparams) / np.sqrt(h)
        val = float(rs.choice([np.nan, np.inf, -np.inf, 1e200, -1e160, 1e-310, 3e101, 1e30]))
        where = str(rs.choice(["w", "w", "x", "y"]))
        if where == "w": W[rs.randint(B), rs.randint(arch.nparams)] = val
        elif where == "x": x[rs.randint(N), rs.randint(d)] = val
        else: y[rs.randint(N), rs.randint(o)] = val
        mod = mlp_ref.build_module(spec)
        ref = [np.empty(B), np.empty((B, arch.nparams)), np.empty((B, N, o))]
        with np.errstate(all="ignore"):
            for b in range(B):
                ref[0][b] = mlp_ref.sse(mod, W[b], x, y)
                ref[1][b] = -2.0 * mlp_ref.logpostgrad(mod, W[b], x, [v for v in y], 1.0)
                ref[2][b] = mlp_ref.forward_flat(mod, W[b], x)
        op = BatchedMLP(arch, x, y)
        ok = True; why = ""
        for pname in paths:
            op.set_path(codes[pname])
            sg, g = op.sse_grad(W); s2, pr = op.sse_pred(W)
            # (predictions: 3e-11 -- a bias-free network with small weights just above the tiny-activation guard of the int8-slice
            # kernels reaches 1.1e-11 of max |pred|: 2^-47 / 0.04 per layer times the cancellation in the last dot product)
            got = {"sse": (sg.double().cpu().numpy(), ref[0], 1e-11), "sse2": (s2.double().cpu().numpy(), ref[0], 1e-11),
                   "pred": (pr.double().cpu().numpy().reshape(B, N, o), ref[2], 3e-11), "grad": (g.double().cpu().numpy(), ref[1], 1e-9)}
            with np.errstate(invalid="ignore", over="ignore"):
                for name, (u, v, tol) in got.items():
                    cu, cv = cls(u), cls(v)
                    if name == "grad":      # an entry that is +-Inf in the reference may be NaN here: the summation order of terms that
                        cu = np.where((cu == 3) & ((cv == 1) | (cv == 2)), cv, cu)      # overflow is not the reference's
                    if not np.array_equal(cu, cv):
                        ok = False; why += " %s:class(%s: %d entries)" % (pname, name, int((cu != cv).sum()))
                        continue
                    for b in range(B):                          # per vector: a huge vector must not hide the others
                        fb = np.isfinite(v[b])
                        if np.any(fb) and np.abs(u[b][fb] - v[b][fb]).max() > tol * max(np.abs(v[b][fb]).max(), 1e-300):
                            err = np.abs(u[b][fb] - v[b][fb]).max() / max(np.abs(v[b][fb]).max(), 1e-300)
                            # beyond the fixed bar: as in run(), an all-finite chain's own one-ulp sensitivity decides (a deep bias-free
                            # linear network whose last dot product cancels to 1e-4 of its terms moves by 4e-13 per ulp: round 4,
                            # seed 61).  Its own generator: the case stream does not depend on which chains needed it.
                            if not isinstance(spec, mlp_ref.MLPSpec) or not (np.all(np.isfinite(ref[0][b])) and np.all(np.isfinite(ref[1][b])) and np.all(np.isfinite(ref[2][b])) and np.all(np.isfinite(W[b]))):
                                floor = 0.0
                            else:
                                sens = _sensitivity(mod, W[b], x, y, np.random.RandomState(977 + b))
                                floor = K_SLICED * sens[{"sse": 0, "sse2": 0, "grad": 1, "pred": 2}[name]]
                            if err > max(tol, floor):
                                ok = False; why += " %s:val(%s[%d] %.1e, floor %.1e)" % (pname, name, b, err, floor)
        op.set_path(_lib.PATH_AUTO)
        nfail += not ok
        if verbose or not ok:
            print(("ok  " if ok else "FAIL"), dims, act, "N", N, "B", B, "bias", bias, where, val, PATHS.get(op.path(B, N, True), "?") + "/" + PATHS.get(op.path(B, N, False), "?"), why, flush=True)
        del op
    if verbose:
        print("exceptional: %d of %d failed" % (nfail, ncases))
    return nfail


def run_exceptional(ncases=60, seed=0, verbose=True, paths=("auto", "generic")):
    old_dt = torch.get_default_dtype()
    torch.set_default_dtype(torch.double)
    try:
        return _run_exceptional(ncases, seed, verbose, paths)
    finally:
        torch.set_default_dtype(old_dt)


def run_fit(ncases=30, seed=0, verbose=True):
    """The training loop of ensemble members (nnfit: quinn/nns/nnfit.py:125-166 -- epochs, minibatches from torch.randperm,
    Adam / SGD with weight decay, best-on-validation snapshot) against the oracle's loop (oracle/fit_ref.py) from the same
    initial weights and generator state: random networks, data sizes, batch sizes (ragged last batch), learning rates; in
    three forms: MSE loss, negative log-posterior with an anchored Gaussian prior (NN_RMS members, nnfit.py:64-66), MSE with
    ReduceLROnPlateau (nnfit.py:91-92, 170-172)."""
    from oracle import fit_ref
    from quinn_amd.nns.mlp import MLP
    from quinn_amd.nns.nnfit import load_flat_into, nnfit
    rs = np.random.RandomState(seed)
    old_dt = torch.get_default_dtype()
    torch.set_default_dtype(torch.double)
    nfail = 0
    flat = lambda m: np.concatenate([q.detach().flatten().cpu().numpy() for q in m.parameters()])
    try:
        for case in range(ncases):
            d = int(rs.choice([1, 2, 3, 3, 6, 8])); o = int(rs.choice([1, 1, 2])); hid = tuple(int(v) for v in rs.choice([3, 8, 11, 32, 64, 70, 128], size=rs.randint(1, 4)))
            act = str(rs.choice(["tanh", "tanh", "relu"])); N = int(rs.randint(2, 200)); Nv = int(rs.randint(1, 60))
            bs = None if rs.rand() < 0.3 else int(rs.randint(1, N + 5)); opt = str(rs.choice(["adam", "adam", "sgd"]))
            lr = float(rs.choice([1e-3, 1e-2, 5e-2])); wd = float(rs.choice([0.0, 1e-3, 0.1])); nep = int(rs.randint(1, 9))
            mode = str(rs.choice(["mse", "mse", "logpost", "plateau"]))
            dims = (d,) + hid + (o,)
            spec = mlp_ref.MLPSpec(dims, act)
            x = rs.rand(N, d) * 2 - 1; y = np.sin(3 * x.sum(axis=1, keepdims=True)) * np.ones((1, o)) + 0.1 * rs.randn(N, o)
            xv = rs.rand(Nv, d) * 2 - 1; yv = np.sin(3 * xv.sum(axis=1, keepdims=True)) * np.ones((1, o))
            w0 = rs.randn(spec.nparams) * 0.3
            gs = int(rs.randint(0, 10000))
            gen = torch.Generator(); gen.manual_seed(gs)
            net = MLP(d, o, hid, activ=act)
            load_flat_into(net, w0)
            if mode == "mse":
                ref = fit_ref.fit_member_mse(spec, w0, x, y, xv, yv, nep, bs, lr, gen, wd=wd, optimizer=opt)
                torch.manual_seed(gs)                       # (after the oracle built its module: that draws from the global generator)
                info = nnfit(net, x, y, val=[xv, yv], lrate=lr, batch_size=bs, nepochs=nep, wd=wd, optimizer=opt, freq_out=100000)
                hist, rh = np.array(info["history"]), ref["history"]
            elif mode == "logpost":
                sig, ps = float(rs.choice([0.1, 0.5])), float(rs.choice([0.5, 2.0]))
                anchor = rs.randn(spec.nparams) * 0.5 if rs.rand() < 0.7 else None
                ref = fit_ref.fit_member_logpost(spec, w0, x, y, xv, yv, nep, bs, lr, gen, sig, anchor=anchor, prior_sigma=ps if anchor is not None else None)
                pp = None if anchor is None else {'sigma': ps, 'anchor': torch.as_tensor(anchor)}
                torch.manual_seed(gs)                       # (after the oracle built its module: that draws from the global generator)
                info = nnfit(net, x, y, val=[xv, yv], loss_fn='logpost', datanoise=sig, priorparams=pp, lrate=lr, batch_size=bs, nepochs=nep, freq_out=100000)
                hist, rh = np.array(info["history"]), ref["history"]
            else:
                nep = int(rs.randint(15, 40)); cd, fac = int(rs.randint(0, 4)), float(rs.choice([0.5, 0.1]))
                ref = fit_ref.fit_member_plateau(spec, w0, x, y, xv, yv, nep, bs, lr, gen, cooldown=cd, factor=fac)
                torch.manual_seed(gs)                       # (after the oracle built its module: that draws from the global generator)
                info = nnfit(net, x, y, val=[xv, yv], lrate=lr, batch_size=bs, nepochs=nep, scheduler_lr="ReduceLROnPlateau", cooldown=cd, factor=fac, freq_out=100000)
                hist, rh = np.array(info["history"])[:, [1, 3]], ref["history"]
            fin = flat(net)
            at = 1e-8 if mode == "plateau" else 1e-10         # (hundreds of Adam steps: rounding differences grow)
            ok = hist.shape == rh.shape and np.allclose(hist, rh, rtol=1e-8, atol=at) and np.allclose(fin, ref["final"], rtol=1e-8, atol=at)
            if mode != "plateau":
                ok = ok and info["best_epoch"] == ref["best_epoch"] and np.allclose(flat(info["best_nnmodel"]), ref["best"], rtol=1e-8, atol=1e-10)
            nfail += not ok
            if verbose or not ok:
                print(("ok  " if ok else "FAIL"), mode, dims, act, "N", N, "Nval", Nv, "batch", bs, opt, "lr", lr, "wd", wd, "epochs", nep,
                      "| max |dw| %.1e" % np.abs(fin - ref["final"]).max(), flush=True)
    finally:
        torch.set_default_dtype(old_dt)
    if verbose:
        print("fit: %d of %d failed" % (nfail, ncases))
    return nfail


def run_ens(ncases=12, seed=0, verbose=True):
    """The batched ensemble trainers (NN_Ens.fit, NN_RMS.fit: every member's forward / backward in ONE call per optimiser
    step) against the oracle's member-after-member loops (oracle/fit_ref.py: fit_ensemble / fit_rms = nn_ens.py:36-69,
    nn_rms.py:41-57): random networks, member counts, data fractions (ragged member subsets), batch sizes, with and without a
    validation set.  Histories, best and final weights of every member."""
    from oracle import fit_ref
    from quinn_amd.nns.mlp import MLP
    from quinn_amd.nns.nnfit import load_flat_into
    from quinn_amd.solvers.nn_ens import NN_Ens
    from quinn_amd.solvers.nn_rms import NN_RMS
    rs = np.random.RandomState(seed)
    old_dt = torch.get_default_dtype()
    torch.set_default_dtype(torch.double)
    nfail = 0
    try:
        for case in range(ncases):
            d = int(rs.choice([1, 2, 3, 3, 6, 8])); o = int(rs.choice([1, 1, 2])); hid = tuple(int(v) for v in rs.choice([3, 8, 11, 32, 64, 70, 128], size=rs.randint(1, 4)))
            act = str(rs.choice(["tanh", "tanh", "relu"])); N = int(rs.randint(4, 150)); Nv = int(rs.randint(1, 40))
            M = int(rs.randint(1, 7)); dfrac = float(rs.choice([1.0, 0.8, 0.5])); kind = str(rs.choice(["ens", "ens", "rms"]))
            bs = None if rs.rand() < 0.3 else int(rs.randint(1, N + 5)); lr = float(rs.choice([1e-3, 1e-2, 5e-2])); nep = int(rs.randint(1, 7))
            noval = bool(rs.rand() < 0.3)
            dims = (d,) + hid + (o,)
            spec = mlp_ref.MLPSpec(dims, act)
            x = rs.rand(N, d) * 2 - 1; y = np.sin(3 * x.sum(axis=1, keepdims=True)) * np.ones((1, o)) + 0.1 * rs.randn(N, o)
            xv = rs.rand(Nv, d) * 2 - 1; yv = np.sin(3 * xv.sum(axis=1, keepdims=True)) * np.ones((1, o))
            w0 = rs.randn(spec.nparams) * 0.3
            ns, ts = int(rs.randint(0, 10000)), int(rs.randint(0, 10000))
            gen = torch.Generator(); gen.manual_seed(ts)
            vx, vy = (None, None) if noval else (xv, yv)
            net = MLP(d, o, hid, activ=act)
            load_flat_into(net, w0)
            if kind == "ens":
                ref = fit_ref.fit_ensemble(spec, w0, x, y, vx, vy, M, dfrac, nep, bs, lr, np.random.RandomState(ns), gen)
                uq = NN_Ens(net, nens=M, dfrac=dfrac, verbose=False)
            else:
                sig, ps = float(rs.choice([0.1, 0.5])), float(rs.choice([0.5, 2.0]))
                ref = fit_ref.fit_rms(spec, w0, x, y, vx, vy, M, dfrac, nep, bs, lr, np.random.RandomState(ns), gen, sig, ps)
                uq = NN_RMS(net, nens=M, dfrac=dfrac, verbose=False, datanoise=sig, priorsigma=ps)
            np.random.seed(ns)
            torch.manual_seed(ts)
            kw = {} if noval else {"val": [xv, yv]}
            uq.fit(x, y, lrate=lr, batch_size=bs, nepochs=nep, freq_out=100000, **kw)
            ok = True
            for j in range(M):
                h = np.array(uq.learners[j].history)
                ok = ok and h.shape == ref[j]["history"].shape and np.allclose(h, ref[j]["history"], rtol=1e-8, atol=1e-10) \
                    and np.allclose(uq.fit_results["best_w"][j], ref[j]["best"], rtol=1e-8, atol=1e-10) \
                    and np.allclose(uq.fit_results["final_w"][j], ref[j]["final"], rtol=1e-8, atol=1e-10)
            nfail += not ok
            if verbose or not ok:
                print(("ok  " if ok else "FAIL"), kind, dims, act, "N", N, "members", M, "dfrac", dfrac, "batch", bs, "lr", lr, "epochs", nep, "noval" if noval else "val",
                      "| max |dw| %.1e" % max(np.abs(uq.fit_results["final_w"][j] - ref[j]["final"]).max() for j in range(M)), flush=True)
    finally:
        torch.set_default_dtype(old_dt)
    if verbose:
        print("ensembles: %d of %d failed" % (nfail, ncases))
    return nfail


def run_vifit(ncases=10, seed=0, verbose=True):
    """NN_VI.fit (nnfit with the ELBO loss: every loss evaluation draws fresh standard normals from the generator) against
    the oracle's loop (oracle/fit_ref.py: fit_vi) from the same (mu, rho) and generator state."""
    from oracle import fit_ref
    from quinn_amd.nns.mlp import MLP
    from quinn_amd.solvers.nn_vi import NN_VI
    rs = np.random.RandomState(seed)
    old_dt = torch.get_default_dtype()
    torch.set_default_dtype(torch.double)
    nfail = 0
    try:
        for case in range(ncases):
            d = int(rs.randint(1, 3)); o = 1; hid = tuple(int(v) for v in rs.choice([3, 8, 16, 33, 64], size=rs.randint(1, 3)))
            act = str(rs.choice(["tanh", "relu"])); N = int(rs.randint(4, 120)); Nv = int(rs.randint(2, 40))
            bs = None if rs.rand() < 0.3 else int(rs.randint(2, N + 3)); lr = float(rs.choice([1e-3, 1e-2])); nep = int(rs.randint(1, 6))
            S = int(rs.choice([1, 3, 8])); sig = float(rs.choice([0.1, 0.5]))
            dims = (d,) + hid + (o,)
            spec = mlp_ref.MLPSpec(dims, act)
            p = spec.nparams
            x = rs.rand(N, d) * 2 - 1; y = np.sin(3 * x.sum(axis=1, keepdims=True)) + 0.1 * rs.randn(N, 1)
            xv = rs.rand(Nv, d) * 2 - 1; yv = np.sin(3 * xv.sum(axis=1, keepdims=True))
            mu0 = rs.uniform(-0.3, 0.3, p); rho0 = rs.uniform(-5.0, -3.0, p)
            gs = int(rs.randint(0, 10000))
            gen = torch.Generator(); gen.manual_seed(gs)
            ref = fit_ref.fit_vi(spec, mu0, rho0, x, y, xv, yv, nep, bs, lr, S, sig, gen)
            vi = NN_VI(MLP(d, o, hid, activ=act), verbose=False)
            with torch.no_grad():
                vi.bmodel.theta.copy_(torch.as_tensor(np.concatenate([mu0, rho0]), device=vi.bmodel.theta.device))
            torch.manual_seed(gs)
            vi.fit(x, y, val=[xv, yv], datanoise=sig, lrate=lr, batch_size=bs, nsam=S, nepochs=nep, freq_out=100000)
            hist = np.array(vi.fit_info["history"])
            th = vi.bmodel.theta.detach().cpu().numpy()
            ok = hist.shape == ref["history"].shape and np.allclose(hist, ref["history"], rtol=1e-8, atol=1e-9) and \
                np.allclose(th[:p], ref["final"][0], rtol=1e-8, atol=1e-10) and np.allclose(th[p:], ref["final"][1], rtol=1e-8, atol=1e-10) and \
                vi.fit_info["best_epoch"] == ref["best_epoch"]
            nfail += not ok
            if verbose or not ok:
                print(("ok  " if ok else "FAIL"), "vi fit", dims, act, "N", N, "batch", bs, "S", S, "lr", lr, "epochs", nep,
                      "| max |dmu| %.1e" % np.abs(th[:p] - ref["final"][0]).max(), flush=True)
    finally:
        torch.set_default_dtype(old_dt)
    if verbose:
        print("vi fit: %d of %d failed" % (nfail, ncases))
    return nfail


def run_device(ncases=20, seed=0, verbose=True):
    """The device-resident samplers (engine='device': adaptive Metropolis and HMC without host synchronisation) on random
    networks / data / settings.  Their random numbers are the kernels' own (Philox), so chains are checked structurally and
    against the oracle's log-posterior at stored states: start state kept, state and log-posterior move together, acceptance
    rate = fraction of moves, stored log-posteriors = oracle's at the stored states (1e-9), MAP bookkeeping, everything finite."""
    from quinn_amd.nns.mlp import MLP
    from quinn_amd.solvers.nn_mcmc import NN_MCMC
    rs = np.random.RandomState(seed)
    old_dt = torch.get_default_dtype()
    torch.set_default_dtype(torch.double)
    nfail = 0
    try:
        for case in range(ncases):
            d = int(rs.choice([1, 2, 3, 3, 6, 8])); o = int(rs.choice([1, 1, 2])); hid = tuple(int(v) for v in rs.choice([4, 8, 11, 16, 33, 64, 64, 100, 128], size=rs.randint(1, 4)))
            act = str(rs.choice(["tanh", "tanh", "relu"])); N = int(rs.randint(3, 500)); sigma = float(rs.choice([0.1, 0.3, 1.0]))
            sampler = str(rs.choice(["amcmc", "hmc"])); C = int(rs.choice([1, 3, 8, 20])); nmcmc = int(rs.randint(30, 120))
            x = rs.rand(N, d) * 4 - 2; y = np.sin(x.sum(axis=1, keepdims=True)) * np.ones((1, o)) + sigma * rs.randn(N, o)
            dims = (d,) + hid + (o,)
            spec = mlp_ref.MLPSpec(dims, act)
            sp = {'gamma': 0.1, 't0': int(rs.randint(5, 25)), 'tadapt': int(rs.randint(5, 25))} if sampler == "amcmc" else \
                 {'epsilon': float(rs.choice([5e-3, 3e-2, 0.1]) * sigma / np.sqrt(N)), 'L': int(rs.randint(1, 5))}
            solver = NN_MCMC(MLP(d, o, hid, activ=act), verbose=False)
            ini = 0.3 * rs.randn(C, spec.nparams)
            solver.fit(x, y, zflag=False, datanoise=sigma, nmcmc=nmcmc, param_ini=ini, sampler=sampler, sampler_params=dict(sp),
                       seeds=[int(v) for v in rs.randint(0, 10000, size=C)], engine='device')
            r = solver.mcmc_results
            chain, lps = np.asarray(r['chain']).reshape(C, nmcmc + 1, -1), np.asarray(r['logpost']).reshape(C, nmcmc + 1)
            moved = (chain[:, 1:] != chain[:, :-1]).any(axis=2)
            why = ""
            if not np.array_equal(chain[:, 0], ini): why += " start"
            if not (np.isfinite(chain).all() and np.isfinite(lps).all()): why += " finite"
            if not np.array_equal(moved, lps[:, 1:] != lps[:, :-1]): why += " moves"
            if not np.allclose(moved.mean(axis=1), np.asarray(r['accrate']).reshape(-1), atol=1e-12): why += " accrate"
            if not np.all(np.asarray(r['maxpost']).reshape(-1) >= lps.max(axis=1) - 1e-9 * np.abs(lps.max(axis=1))): why += " maxpost"
            mod = mlp_ref.build_module(spec)
            yd = [v for v in y]
            for c, i in [(0, 0), (C - 1, nmcmc // 2), (C // 2, nmcmc)]:
                ref = mlp_ref.logpost(mod, chain[c, i], x, yd, sigma)
                if abs(lps[c, i] - ref) > 1e-9 * abs(ref): why += " logpost(%d,%d: %.1e)" % (c, i, abs(lps[c, i] / ref - 1))
            nfail += bool(why)
            if verbose or why:
                print(("FAIL" if why else "ok  "), dims, act, "N", N, "sigma", sigma, sampler, sp, "chains", C, "steps", nmcmc, "| accept %.2f" % moved.mean(), why, flush=True)
    finally:
        torch.set_default_dtype(old_dt)
    if verbose:
        print("device engines: %d of %d failed" % (nfail, ncases))
    return nfail


def run_vi(ncases=40, seed=0, verbose=True):
    """The ELBO Monte-Carlo estimator (BNet.viloss: quinn/vi/bnet.py:178-232) and its gradient with respect to (mu, rho)
    against the oracle (oracle/vi_ref.py) on the same standard normals: random networks, MC sample counts, mixture
    priors, noise levels, batch counts."""
    from oracle import vi_ref
    from quinn_amd.nns.mlp import MLP
    from quinn_amd.vi.bnet import BNet
    rs = np.random.RandomState(seed)
    old_dt = torch.get_default_dtype()
    torch.set_default_dtype(torch.double)
    worst = [0.0, 0.0]; nfail = 0
    try:
        for case in range(ncases):
            d = int(rs.choice([1, 2, 3, 4, 5, 7])); o = int(rs.choice([1, 1, 2, 3]))
            hid = tuple(int(v) for v in rs.choice([2, 5, 11, 16, 64, 64, 70, 128], size=rs.randint(1, 4)))
            act = str(rs.choice(["tanh", "tanh", "relu"])); N = int(rs.choice([rs.randint(1, 30), rs.randint(30, 500)])); S = int(rs.choice([1, 2, 5, 16, 33]))
            prior = dict(pi=float(rs.choice([0.5, 0.25, 1.0])), sigma1=float(rs.choice([1.0, 0.5, 2.0])), sigma2=float(rs.choice([1.0, 0.1, 0.0025])))
            sig = float(rs.choice([0.05, 0.3, 1.0])); nb = int(rs.randint(1, 9))
            dims = (d,) + hid + (o,)
            spec = mlp_ref.MLPSpec(dims, act)
            p = spec.nparams
            mu = rs.uniform(-0.5, 0.5, p); rho = rs.uniform(-6.0, -1.0, p); eps = rs.randn(S, p)
            x = rs.rand(N, d) * 4 - 2; y = rs.randn(N, o)
            ref = vi_ref.viloss(spec, mu, rho, eps, x, y, sig, nb, **prior)
            bm = BNet(MLP(d, o, hid, activ=act), **prior)
            with torch.no_grad():
                bm.theta.copy_(torch.as_tensor(np.concatenate([mu, rho]), device=bm.theta.device))
            bm._draw_eps = lambda n: torch.as_tensor(eps, device=bm.device)
            bm.loss_params = [sig, S, nb]
            loss = bm.viloss(x, y)
            loss.backward()
            gr = bm.theta.grad.cpu().numpy()
            sc = max(np.abs(ref["dmu"]).max(), np.abs(ref["drho"]).max())
            e = [abs(loss.item() / ref["loss"] - 1), max(np.abs(gr[:p] - ref["dmu"]).max(), np.abs(gr[p:] - ref["drho"]).max()) / sc]
            ok = e[0] <= 1e-11 and e[1] <= 1e-10
            nfail += not ok
            worst = [max(u, v) for u, v in zip(worst, e)]
            if verbose or not ok:
                print(("ok  " if ok else "FAIL"), dims, act, "N", N, "S", S, prior, "sigma", sig, "batches", nb, "| loss %.1e grad %.1e" % tuple(e), flush=True)
    finally:
        torch.set_default_dtype(old_dt)
    if verbose:
        print("vi worst: loss %.2e grad %.2e; %d of %d failed" % (*worst, nfail, ncases))
    return nfail, worst


FAMILIES = (("operator", "run", 1.0, 1), ("residual networks", "run_rnet", 0.5, 10), ("ELBO", "run_vi", 0.25, 10),
            ("not-finite values", "run_exceptional", 1.0, 1), ("training loops", "run_fit", 0.2, 10), ("ensembles", "run_ens", 0.1, 6),
            ("VI fits", "run_vifit", 0.1, 6), ("device samplers", "run_device", 0.2, 10), ("host samplers", "run_mcmc", 0.05, 6))


if __name__ == "__main__":
    # usage: tests/fuzz_all.py [ncases] [seed] [family,family,...]   (families by function name: run, run_rnet, run_vi, run_exceptional,
    # run_fit, run_ens, run_vifit, run_device, run_mcmc; default: all).  EVERY case prints its own line as it finishes (a GPU box takes
    # seven silent minutes for a hang: round 3 lost a sweep to that while the host-sampler family -- CPU-side reference loops with an
    # SVD of p x p per step -- ran without output); run that family on its own (`... run_mcmc`) so that the GPU is not held for it
    # longer than needed.
    import time
    nc, sd = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0
    only = set(sys.argv[3].split(",")) if len(sys.argv) > 3 else None
    first = lambda r: r[0] if isinstance(r, tuple) else r
    total = 0
    summary = []
    for name, fname, frac, least in FAMILIES:
        if only is not None and fname not in only:
            continue
        n = max(least, int(nc * frac))
        t0 = time.time()
        print("---- %s: %d cases" % (name, n), flush=True)
        nf = first(globals()[fname](n, sd, verbose=True))
        total += nf
        summary.append("%-20s %4d cases, %d failed, %.0f s" % (name, n, nf, time.time() - t0))
        print(summary[-1], flush=True)
    print("==== summary")
    print("\n".join(summary), flush=True)
    sys.exit(1 if total else 0)
