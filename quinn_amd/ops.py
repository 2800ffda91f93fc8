"""The batched operator behind every solver: for B flat weight vectors at once, the MLP's
sum of squared errors over a (shared or per-member) set of data rows, its gradient, and
the predictions -- evaluated by the HIP kernels through the C ABI.

torch is used for device memory and streams only.
"""
import ctypes
import os
from dataclasses import dataclass
from typing import Tuple

import numpy as np
import torch

from . import _lib
from ._lib import QuinnAmdError

_TORCH_DT = {"float64": torch.float64, "float32": torch.float32}
_QN_DT = {"float64": _lib.QN_F64, "float32": _lib.QN_F32}


@dataclass(frozen=True)
class MLPArch:
    """dims = (d, h_1, ..., h_L, o); flat layout [W_0, b_0, W_1, b_1, ...] with W row-major
    (out x in), i.e. module.parameters() order (reference quinn/nns/nnwrap.py:70-77)."""
    dims: Tuple[int, ...]
    activ: str = "tanh"
    bias: bool = True

    @property
    def nparams(self):
        return sum(a * b + (b if self.bias else 0) for a, b in zip(self.dims[:-1], self.dims[1:]))

    @property
    def nweights(self):
        return sum(a * b for a, b in zip(self.dims[:-1], self.dims[1:]))

    def param_shapes(self):
        """Shapes of the module's parameter tensors in parameters() order."""
        shapes = []
        for a, b in zip(self.dims[:-1], self.dims[1:]):
            shapes.append((b, a))
            if self.bias:
                shapes.append((b,))
        return shapes

    def create_desc(self, L):
        dims = (ctypes.c_int * len(self.dims))(*self.dims)
        h = ctypes.c_void_p()
        _lib.check(L.qn_mlp_desc_create(dims, len(self.dims), _lib.ACT_CODES[self.activ], int(self.bias),
                                        ctypes.byref(h)), "qn_mlp_desc_create")
        return h

    def flops_fwd(self, n):
        """2*N*Wn."""
        return 2 * n * self.nweights

    def flops_fwdbwd(self, n):
        """6*N*Wn - 2*N*d*h_1 (no input gradient for the first layer)."""
        return 6 * n * self.nweights - 2 * n * self.dims[0] * self.dims[1]

    @staticmethod
    def from_module(nnmodel):
        """Pattern-match Sequential(Linear, act, Linear, ..., Linear) -- directly, or as the
        `.nnmodel` attribute of a quinn-style MLP (reference quinn/nns/mlp.py:86).  A quinn-style
        residual network (attributes of quinn/nns/rnet.py:16-127) yields an `RNetArch`."""
        if RNetArch.matches(nnmodel):
            return RNetArch.from_module(nnmodel)
        seq = nnmodel
        if isinstance(seq, torch.nn.Linear):                   # a bare linear model (reference examples/ex_lreg_mcmc.py:54)
            seq = torch.nn.Sequential(seq)
        if not isinstance(seq, torch.nn.Sequential):
            seq = getattr(nnmodel, "nnmodel", None)
        if not isinstance(seq, torch.nn.Sequential):
            raise NotImplementedError(
                f"{type(nnmodel).__name__}: only MLPs built as Sequential(Linear, act, ..., Linear) "
                "are handled by the MI355X path")
        mods = list(seq)
        dims, acts, bias = [], set(), None
        expect_linear = True
        for m in mods:
            if expect_linear:
                if not isinstance(m, torch.nn.Linear):
                    raise NotImplementedError(f"unexpected layer {type(m).__name__} (wanted Linear)")
                if not dims:
                    dims.append(m.in_features)
                elif dims[-1] != m.in_features:
                    raise ValueError("layer widths do not chain")
                dims.append(m.out_features)
                b = m.bias is not None
                if bias is None:
                    bias = b
                elif bias != b:
                    raise NotImplementedError("mixed bias / no-bias layers")
                expect_linear = False
            else:
                if isinstance(m, torch.nn.Tanh):
                    acts.add("tanh")
                elif isinstance(m, torch.nn.ReLU):
                    acts.add("relu")
                elif isinstance(m, torch.nn.Identity):
                    acts.add("identity")
                else:
                    raise NotImplementedError(f"activation {type(m).__name__} is not handled")
                expect_linear = True
        if expect_linear or len(dims) < 2:
            raise NotImplementedError("module must end with a Linear layer")
        if len(acts) > 1:
            raise NotImplementedError("mixed activations")
        return MLPArch(tuple(dims), acts.pop() if acts else "identity", bool(bias))


@dataclass(frozen=True)
class RNetArch:
    """Residual network of the reference (quinn/nns/rnet.py:16-165): optional pre layer (with
    activation), `nsteps = nlayers + 1` residual steps out += h * act(W_i out + b_i) (or plain
    layers if `mlp`), optional linear post layer.  `coef[i][k]` expresses the weight
    parameterisation W_i = sum_k coef[i][k] ww_k (rnet.py:217-380, all linear in the ww_k).
    Flat layout = parameters() order: weight_pre, bias_pre, weight_post, bias_post, ww_*, bb_*."""
    indim: int
    rdim: int
    outdim: int
    nsteps: int
    coef: Tuple[Tuple[float, ...], ...]
    activ: str = "tanh"
    bias: bool = True
    layer_pre: bool = False
    layer_post: bool = False
    mlp: bool = False
    uses: Tuple[Tuple[bool, ...], ...] = ()     # uses[i][k]: tensor k enters step i at all (default: every one does)

    @property
    def dims(self):
        return (self.indim, self.rdim, self.outdim)

    @property
    def npar(self):
        return len(self.coef[0])

    def param_shapes(self):
        r = self.rdim
        shapes = []
        if self.layer_pre:
            shapes += [(r, self.indim), (r,)]
        if self.layer_post:
            shapes += [(self.outdim, r), (self.outdim,)]
        shapes += [(r, r)] * self.npar
        if self.bias:
            shapes += [(r,)] * self.npar
        return shapes

    @property
    def nparams(self):
        return sum(int(np.prod(s)) for s in self.param_shapes())

    @property
    def nweights(self):
        """Multiply-accumulates per data row."""
        r = self.rdim
        return (r * self.indim if self.layer_pre else 0) + self.nsteps * r * r + \
            (self.outdim * r if self.layer_post else 0)

    def flops_fwd(self, n):
        return 2 * n * self.nweights

    def flops_fwdbwd(self, n):
        return 6 * n * self.nweights - (2 * n * self.rdim * self.indim if self.layer_pre else 0)

    def create_desc(self, L):
        if self.activ not in ("tanh", "identity"):
            raise NotImplementedError("RNet activations: tanh (nonlin=True) or identity")
        flat = [float(c) for row in self.coef for c in row]
        arr = (ctypes.c_double * len(flat))(*flat)
        h = ctypes.c_void_p()
        _lib.check(L.qn_rnet_desc_create(self.indim, self.rdim, self.outdim, self.nsteps, self.npar, arr,
                                         _lib.ACT_CODES[self.activ], int(self.bias), int(self.layer_pre),
                                         int(self.layer_post), int(self.mlp), ctypes.byref(h)),
                   "qn_rnet_desc_create")
        if self.uses:
            u = [1 if v else 0 for row in self.uses for v in row]
            _lib.check(L.qn_rnet_desc_set_uses(h, (ctypes.c_ubyte * len(u))(*u), len(u)), "qn_rnet_desc_set_uses")
        return h

    @staticmethod
    def matches(nnmodel):
        return all(hasattr(nnmodel, a) for a in ("rdim", "nlayers", "wp_function", "step_size", "layer_pre",
                                                 "layer_post", "biasorno", "mlp"))

    @staticmethod
    def from_module(nnmodel):
        """From this package's `RNet` or the reference's (same attributes).  The weight
        parameterisation is probed with unit 'parameters' to get its coefficients, and checked to
        be linear."""
        if getattr(nnmodel, "final_layer", None) is not None:
            raise NotImplementedError("RNet final_layer is outside the MI355X hot path")
        wp = nnmodel.wp_function
        npar, nsteps = int(wp.npar), int(nnmodel.nlayers) + 1
        coef, uses = [], []
        for i in range(nsteps):
            t = nnmodel.step_size * i                                  # rnet.py:146
            row = []
            # does tensor k enter the step at all?  A NaN there shows in the result if it does -- also through a zero
            # coefficient (the polynomials multiply t^k out; NonPar picks one tensor and never touches the others)
            uses.append(tuple(bool(np.isnan(float(wp([float('nan') if q == k else 0.0 for q in range(npar)], t)))) for k in range(npar)))
            for k in range(npar):
                e = [1.0 if q == k else 0.0 for q in range(npar)]
                c = float(wp(e, t))
                if float(wp([2.0 * v for v in e], t)) != 2.0 * c or float(wp([0.0] * npar, t)) != 0.0:
                    raise NotImplementedError("weight parameterisation must be linear in its parameters")
                row.append(c)
            coef.append(tuple(row))
        act = nnmodel.activ
        activ = "tanh" if isinstance(act, torch.nn.Tanh) else "identity" if isinstance(act, torch.nn.Identity) \
            else None
        if activ is None:
            raise NotImplementedError(f"RNet activation {type(act).__name__}")
        return RNetArch(int(nnmodel.indim), int(nnmodel.rdim), int(nnmodel.outdim), nsteps, tuple(coef), activ,
                        bool(nnmodel.biasorno), bool(nnmodel.layer_pre), bool(nnmodel.layer_post),
                        bool(nnmodel.mlp), tuple(uses))


def flatten_module(nnmodel):
    """Flat float64 numpy vector of module.parameters() (reference nnwrap.py:70-77)."""
    return np.concatenate([p.detach().cpu().double().flatten().numpy() for p in nnmodel.parameters()])


def default_device(device=None):
    if device is not None:
        dev = torch.device(device)
    else:
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    if dev is None or dev.type != "cuda" or not torch.cuda.is_available():
        raise QuinnAmdError("quinn_amd needs an AMD GPU (HIP device) -- there is no CPU fallback")
    return dev


class BatchedMLP:
    """Device-resident dataset + architecture descriptor + workspace; calls the C ABI."""

    def __init__(self, arch: MLPArch, x, y, device=None, dtype="float64", max_workspace_bytes=48 << 30):
        self.arch = arch
        self.device = default_device(device)
        if dtype not in _TORCH_DT:
            raise ValueError(f"dtype {dtype!r}")
        self.dtype = dtype
        self.tdt = _TORCH_DT[dtype]
        self.qdt = _QN_DT[dtype]
        self.max_ws = int(max_workspace_bytes)
        self._L = _lib.lib()
        self._desc = h = arch.create_desc(self._L)
        self.p = int(self._L.qn_mlp_num_params(h))
        assert self.p == arch.nparams
        self._ws = None
        self.set_data(x, y)

    def __del__(self):
        try:
            if getattr(self, "_desc", None):
                self._L.qn_mlp_desc_destroy(self._desc)
                self._desc = None
        except Exception:
            pass

    # ------------------------------------------------------------------ data / buffers
    def _dev(self, a, dt=None):
        dt = dt or self.tdt
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dt).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a), device=self.device).to(dt).contiguous()

    def set_data(self, x, y):
        d, o = self.arch.dims[0], self.arch.dims[-1]
        self.X = self._dev(x).reshape(-1, d)
        self.N = self.X.shape[0]
        if y is None:
            self.Y = torch.zeros(self.N, o, device=self.device, dtype=self.tdt)
        else:
            self.Y = self._dev(y).reshape(-1, o)
        if self.Y.shape[0] != self.N:
            raise ValueError("x and y row counts differ")

    def workspace_bytes(self, B, Nb, want_grad):
        return int(self._L.qn_workspace_bytes(self._desc, B, Nb, int(want_grad), self.qdt))

    def path(self, B, Nb=None, want_grad=False):
        return int(self._L.qn_mlp_path(self._desc, B, Nb or self.N, int(want_grad), self.qdt))

    def arith(self, B, Nb=None, want_grad=False):
        """`_lib.ARITH_PLAIN` / `ARITH_I8_FUSED` / `ARITH_I8_WIDE` / `ARITH_I8_LAYERS`: the arithmetic the next call with these
        sizes forms the hidden-layer products in (qn_mlp_arith)."""
        return int(self._L.qn_mlp_arith(self._desc, B, Nb or self.N, int(want_grad), self.qdt))

    def set_path(self, path):
        """Force a kernel family for THIS operator (`_lib.PATH_AUTO` / `PATH_GENERIC` / `PATH_FUSED`; tests and
        profiling); returns the previous setting.  Per descriptor: other operators are unaffected."""
        return int(self._L.qn_mlp_desc_set_path(self._desc, int(path)))

    def set_plan_batch(self, batch):
        """Split every chain's rows as a launch of max(B, batch) chains would (qn_mlp_desc_set_plan_batch): a batch evaluated
        as several smaller launches then gives the one-launch results bit for bit.  Returns the previous setting."""
        return int(self._L.qn_mlp_desc_set_plan_batch(self._desc, int(batch)))

    def use_exact_float64(self):
        """Plain float64 arithmetic for THIS operator: under `PATH_AUTO` the float64 operator of 64 / 128 / 256-wide tanh
        networks runs the sliced int8-product kernels (operands rounded to 2^-47 of their row / activation scale: a
        norm-wise 47-bit bound, ~1e-14 .. 1e-13 on SSE / gradients of ordinary networks).  This selects the float64-MFMA
        fused kernels where the shape allows (`PATH_FUSED_DP`), the layer-wise float64 kernels otherwise (`PATH_GENERIC`).
        Returns the path taken."""
        from . import _lib as L
        self.set_path(L.PATH_FUSED_DP)
        try:
            ok = self.path(1, self.N, True) == L.PATH_FUSED and self.path(1, self.N, False) == L.PATH_FUSED
        except Exception:
            ok = False
        if not ok:
            self.set_path(L.PATH_GENERIC)
        return L.PATH_FUSED_DP if ok else L.PATH_GENERIC

    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws

    def _chunk(self, B, Nb, want_grad):
        bc = min(B, 65535)                       # one chain per blockIdx.y / .z: the grid limit of the C ABI
        while bc > 1 and self.workspace_bytes(bc, Nb, want_grad) > self.max_ws:
            bc = (bc + 1) // 2
        return bc

    def weights(self, W):
        """[B, p] device tensor in the compute dtype (numpy float64 input is uploaded)."""
        Wt = self._dev(W)
        if Wt.dim() == 1:
            Wt = Wt.unsqueeze(0)
        if Wt.shape[1] != self.p:
            raise ValueError(f"weight vectors have {Wt.shape[1]} entries, the network has {self.p}")
        return Wt

    # ------------------------------------------------------------------ the operator
    def _call(self, W, row_idx, want_pred, want_grad, X=None, Y=None, out=None):
        X = self.X if X is None else X
        Y = self.Y if Y is None else Y
        N = X.shape[0]
        Wt = self.weights(W)
        B = Wt.shape[0]
        if row_idx is not None:
            ridx = torch.as_tensor(row_idx, device=self.device).to(torch.int32).contiguous().reshape(B, -1)
            Nb = ridx.shape[1]
        else:
            ridx, Nb = None, N
        o = self.arch.dims[-1]
        if out is not None:                          # caller-owned result buffers (engines that run inside a HIP graph)
            sse, grad = out
            if sse.shape != (B,) or sse.dtype != torch.float64 or not sse.is_contiguous() or \
                    (want_grad and (grad.shape != (B, self.p) or grad.dtype != self.tdt or not grad.is_contiguous())):
                raise ValueError("out=(sse [B] float64, grad [B, p] compute dtype) does not match the call")
        else:
            sse = torch.empty(B, dtype=torch.float64, device=self.device)
            grad = torch.empty(B, self.p, dtype=self.tdt, device=self.device) if want_grad else None
        pred = torch.empty(B, Nb, o, dtype=self.tdt, device=self.device) if want_pred else None
        if B == 0:                                   # nothing to evaluate (e.g. an empty shard of chains)
            return sse, pred, grad
        bc = self._chunk(B, Nb, want_grad)
        ws = self._workspace(self.workspace_bytes(bc, Nb, want_grad))
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        with torch.cuda.device(self.device):
            for b0 in range(0, B, bc):
                b1 = min(B, b0 + bc)
                nb = b1 - b0
                args = [self._desc, self.qdt, Wt[b0:b1].data_ptr(), X.data_ptr(), Y.data_ptr(),
                        ridx[b0:b1].data_ptr() if ridx is not None else None, nb, N, Nb,
                        sse[b0:b1].data_ptr(), pred[b0:b1].data_ptr() if pred is not None else None]
                if want_grad:
                    rc = self._L.qn_mlp_sse_fwdbwd(*args, grad[b0:b1].data_ptr(), ws.data_ptr(), ws.numel(), stream)
                    _lib.check(rc, "qn_mlp_sse_fwdbwd")
                else:
                    rc = self._L.qn_mlp_sse_fwd(*args, ws.data_ptr(), ws.numel(), stream)
                    _lib.check(rc, "qn_mlp_sse_fwd")
        return sse, pred, grad

    def sse(self, W, row_idx=None):
        """sum_{n,o} (y - f_W(x))^2 for every weight vector: float64 device tensor [B]."""
        return self._call(W, row_idx, False, False)[0]

    def sse_parts(self, W):
        """[B, parts] float64 partial sums whose left-to-right sum is `sse(W)` bit for bit (qn_mlp_sse_fwd_parts): the
        forward without its final summation launch, for `qn_mcmc_accept`, which adds the handful of numbers itself."""
        Wt = self.weights(W)
        B, N = Wt.shape[0], self.N
        if B == 0 or self._chunk(B, N, False) < B or os.environ.get("QUINN_AMD_NO_PARTS"):     # (env: A/B measurements)
            return self.sse(Wt).reshape(B, 1)
        parts = int(self._L.qn_mlp_sse_parts(self._desc, B, N, self.qdt))
        if parts < 1:
            raise _lib.QuinnAmdError("qn_mlp_sse_parts failed")
        out = torch.empty(B, parts, dtype=torch.float64, device=self.device)
        ws = self._workspace(self.workspace_bytes(B, N, False))
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        with torch.cuda.device(self.device):
            _lib.check(self._L.qn_mlp_sse_fwd_parts(self._desc, self.qdt, Wt.data_ptr(), self.X.data_ptr(), self.Y.data_ptr(),
                                                    None, B, N, N, out.data_ptr(), ws.data_ptr(), ws.numel(), stream),
                       "qn_mlp_sse_fwd_parts")
        return out

    def sse_grad(self, W, row_idx=None, out=None):
        """(sse [B] float64, d sse / d W [B, p] compute dtype), device tensors; `out=(sse, grad)` writes into the
        caller's buffers instead of allocating."""
        s, _, g = self._call(W, row_idx, False, True, out=out)
        return s, g

    def sse_pred(self, W, row_idx=None):
        s, pr, _ = self._call(W, row_idx, True, False)
        return s, pr

    def predict(self, W, x=None):
        """f_W(x) for every weight vector: [B, N, o] device tensor (x defaults to the stored X)."""
        if x is None:
            return self._call(W, None, True, False)[1]
        X = self._dev(x).reshape(-1, self.arch.dims[0])
        Y = torch.zeros(X.shape[0], self.arch.dims[-1], device=self.device, dtype=self.tdt)
        return self._call(W, None, True, False, X=X, Y=Y)[1]


def neg_log_post_from_sse(sse, n, sigma):
    """0.5*SSE/sigma^2 + (n/2)*log(2*pi) + n*log(sigma) in float64 with the operation order
    of the reference's NegLogPost.forward (quinn/nns/losses.py:198-200); sse: float64 array."""
    sse = np.asarray(sse, dtype=np.float64)
    sig = np.float64(sigma)
    val = 0.5 * sse / sig ** 2
    val = val + (n / 2) * np.log(2 * np.float64(np.pi))
    val = val + n * np.log(sig)
    return val
