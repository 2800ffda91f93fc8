"""Oracle: MLP log-posterior and its parameter gradient, one weight vector at a time.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  float64 on the CPU, through
the same torch primitives the reference calls, in the same order.

Restates (reference file:line, relative to /root/reference):
  * flat parameter layout / unflatten ........ quinn/nns/nnwrap.py:64-79, 81-106
  * MLP = Linear, act, ..., Linear ............ quinn/nns/mlp.py:46-84
  * negative log-posterior (no prior) ......... quinn/nns/losses.py:197-200
  * NN_MCMC.logpost / logpostgrad ............. quinn/solvers/nn_mcmc.py:45-71, 73-98
  * NNWrap.calc_loss / calc_lossgrad .......... quinn/nns/nnwrap.py:109-126, 128-150
  * nn_p (forward with a flat vector) ......... quinn/nns/nnwrap.py:330-347
  * numpy -> tensor conversion (tch) .......... quinn/nns/tchutils.py:11-28
"""
from dataclasses import dataclass
from typing import Tuple

import numpy as np
import torch

F64 = torch.float64


@dataclass(frozen=True)
class MLPSpec:
    """Architecture of a quinn-style MLP: dims = (d, h_1, ..., h_L, o)."""
    dims: Tuple[int, ...]
    activ: str = "tanh"          # 'tanh' | 'relu' | anything else = identity (mlp.py:46-53)
    bias: bool = True

    @property
    def nparams(self):
        n = 0
        for a, b in zip(self.dims[:-1], self.dims[1:]):
            n += a * b + (b if self.bias else 0)
        return n

    @property
    def nweights(self):
        return sum(a * b for a, b in zip(self.dims[:-1], self.dims[1:]))


def _activation(name):
    if name == "tanh":
        return torch.nn.Tanh()
    if name == "relu":
        return torch.nn.ReLU()
    return torch.nn.Identity()


def build_module(spec) -> torch.nn.Module:
    """Sequential(Linear, act, Linear, ..., act, Linear) in float64 (mlp.py:55-84); a residual
    network for an rnet_ref.RNetSpec."""
    if hasattr(spec, "build_module"):
        return spec.build_module()
    layers = []
    nl = len(spec.dims) - 1
    for i in range(nl):
        layers.append(torch.nn.Linear(spec.dims[i], spec.dims[i + 1], bias=spec.bias, dtype=F64))
        if i < nl - 1:
            layers.append(_activation(spec.activ))
    return torch.nn.Sequential(*layers)


def to_tensor(arr):
    """numpy / list -> float64 CPU tensor by copy (tchutils.py:23-27)."""
    if isinstance(arr, (list, float, int)):
        arr = np.array(arr)      # python floats become float64 here (the reference runs with
                                 # torch's default dtype set to double, tchutils.py:9)
    t = torch.tensor(arr, requires_grad=False, device="cpu")
    if t.is_floating_point():
        t = t.to(F64)
    return t


def index_table(module):
    """[start, end) of every parameter in the flat vector, in parameters() order
    (nnwrap.py:70-77).  The torch.cat is part of what the reference does per call."""
    flat = [torch.flatten(p) for p in module.parameters()]
    table, s = [], 0
    for p in flat:
        table.append((s, s + p.shape[0]))
        s += p.shape[0]
    _ = torch.cat(flat).view(-1, 1)
    return table


def load_flat(module, w, table=None):
    """Overwrite the module's parameters from flat vector w (nnwrap.py:99-104)."""
    if table is None:
        table = index_table(module)
    pieces = [to_tensor(w[s:e]) for (s, e) in table]
    for piece, p in zip(pieces, module.parameters()):
        p.data = piece.view(*p.shape) if p.dim() > 0 else piece


def neg_log_post(module, x_t, y_t, sigma):
    """0.5*sum((y-f(x))^2)/sigma^2 + (n/2)*log(2*pi) + n*log(sigma), n = len(pred)
    (losses.py:197-200).  sigma and pi are 0-d float64 tensors as in losses.py:181-183."""
    sig = to_tensor(float(sigma))
    pi = to_tensor(np.pi)
    pred = module(x_t)
    val = 0.5 * torch.sum(torch.pow(y_t - pred, 2)) / sig ** 2
    val = val + (len(pred) / 2) * torch.log(2 * pi)
    val = val + len(pred) * torch.log(sig)
    return val


def logpost(module, w, xd, yd, sigma):
    """One log-posterior evaluation exactly as the reference's sequential path does it:
    index table, unflatten, loss object, dataset list->array->tensor copies, unflatten
    again, forward, .item() (nn_mcmc.py:55-66 -> nnwrap.py:121-126)."""
    table = index_table(module)
    load_flat(module, w, table)
    x_t = to_tensor(xd)
    y_t = to_tensor(yd)
    load_flat(module, w, table)
    val = neg_log_post(module, x_t, y_t, sigma)   # graph is built, as in the reference
    return -val.item()


def logpostgrad(module, w, xd, yd, sigma):
    """Gradient of the log-posterior w.r.t. the flat vector (nn_mcmc.py:83-93 ->
    nnwrap.py:141-150): autograd through the loss, gather p.grad, flip the sign."""
    table = index_table(module)
    load_flat(module, w, table)
    x_t = to_tensor(xd)
    y_t = to_tensor(yd)
    load_flat(module, w, table)
    for p in module.parameters():
        p.requires_grad_(True)
    val = neg_log_post(module, x_t, y_t, sigma)
    val.backward()
    grads = []
    for p in module.parameters():
        grads.append(p.grad.cpu().data.numpy().flatten())
        p.grad = None
    return -np.concatenate(grads, axis=0)


def forward_flat(module, w, x):
    """f_w(x) as numpy (N, o) (nnwrap.py:344-347)."""
    load_flat(module, w)
    with torch.no_grad():
        return module(to_tensor(x)).cpu().data.numpy()


def sse(module, w, x, y):
    """sum((y - f_w(x))^2): the quantity the HIP kernel returns per weight vector."""
    load_flat(module, w)
    with torch.no_grad():
        return torch.sum(torch.pow(to_tensor(y) - module(to_tensor(x)), 2)).item()


def logpost_from_sse(sse_val, n, sigma):
    """Scalar tail of losses.py:198-200 applied to a given SSE, in float64 with the
    reference's operation order (host-side formula used with the kernel's sse[B])."""
    sig = to_tensor(float(sigma))
    pi = to_tensor(np.pi)
    val = 0.5 * to_tensor(float(sse_val)) / sig ** 2
    val = val + (n / 2) * torch.log(2 * pi)
    val = val + n * torch.log(sig)
    return -val.item()


def synthetic_data(N, d, datanoise=0.02, seed=0):
    """Benchmark data (shape of examples/ex_ufit.py:56-62): x uniform on
    [-pi, pi]^d (maps.py:22), y = sum_j sin(x_j) + datanoise * randn (funcs.py:42-43)."""
    rs = np.random.RandomState(seed)
    dom = np.tile(np.array([-np.pi, np.pi]), (d, 1))
    x = rs.rand(N, d) * np.abs(dom[:, 1] - dom[:, 0]) + np.min(dom, axis=1)
    y = datanoise * rs.randn(N, 1)
    y += np.sum(np.sin(x), axis=1).reshape(-1, 1)
    return x, y
