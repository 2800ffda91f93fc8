"""numpy <-> module glue on top of the batched device operator.

Mirror of the hot-path members of the reference's `NNWrap` and `nn_p` (quinn/nns/nnwrap.py:9-150,
330-347): flat-vector <-> parameters (`p_flatten` / `p_unflatten`, parameters() order), prediction
with a given flat vector, loss and loss-gradient of a `NegLogPost` as numpy values.  The Hessian
helpers of the reference (nnwrap.py:153-229) belong to the Laplace solver and are out of scope.
"""
import numpy as np
import torch

from ..ops import MLPArch, BatchedMLP
from .losses import NegLogPost


class NNWrap():
    def __init__(self, nnmodel, device=None, dtype="float64"):
        self.nnmodel = nnmodel
        self.indices = None
        self._arch = MLPArch.from_module(nnmodel)
        self._opargs = dict(device=device, dtype=dtype)
        self._op = None
        _ = self.p_flatten()

    def p_flatten(self):
        """`(p,1)` tensor of all parameters; also (re)builds `indices` = [start, end) per parameter."""
        flat = [torch.flatten(p) for p in self.nnmodel.parameters()]
        self.indices, s = [], 0
        for p in flat:
            self.indices.append((s, s + p.shape[0]))
            s += p.shape[0]
        return torch.cat(flat).view(-1, 1)

    def p_unflatten(self, flat_parameter):
        """Fill the module's parameters from a flat numpy vector; returns the list of tensors."""
        flat_parameter = np.asarray(flat_parameter, dtype=np.float64).reshape(-1)
        out = []
        for (s, e), p in zip(self.indices, self.nnmodel.parameters()):
            t = torch.tensor(flat_parameter[s:e], dtype=torch.float64)
            t = t.view(*p.shape) if p.dim() > 0 else t
            p.data = t.to(p.device)
            out.append(t)
        return out

    def _predict(self, weights, x):
        x = np.asarray(x, dtype=np.float64)
        if self._op is None:
            self._op = BatchedMLP(self._arch, x, None, **self._opargs)
        return self._op.predict(np.asarray(weights, dtype=np.float64).reshape(1, -1), x)[0].double().cpu().numpy()

    def __call__(self, x):
        return self._predict(self.p_flatten().detach().cpu().numpy().reshape(-1), x)

    def predict(self, x_in, weights):
        self.p_unflatten(weights)
        return self._predict(weights, x_in)

    def calc_loss(self, weights, loss_fn, inputs, targets):
        """float: `loss_fn(inputs, targets)` with the module carrying the given flat weights (nnwrap.py:109-126).
        A `NegLogPost` is evaluated on the device operator; any other callable is simply called on the tensors,
        as the reference does (`loss_fn(inputs, targets).item()`, e.g. `torch.nn.MSELoss()` in its tests)."""
        self.p_unflatten(weights)
        if isinstance(loss_fn, NegLogPost):
            return loss_fn.value_and_grad(weights, inputs, np.asarray(targets))[0]
        return float(loss_fn(torch.as_tensor(np.asarray(inputs), dtype=torch.float64),
                             torch.as_tensor(np.asarray(targets), dtype=torch.float64)).item())

    def calc_lossgrad(self, weights, loss_fn, inputs, targets):
        """np.ndarray `(p,)`: gradient of `loss_fn` w.r.t. the flat weights (nnwrap.py:128-150)."""
        if not isinstance(loss_fn, NegLogPost):
            raise NotImplementedError("calc_lossgrad on the accelerated path takes a quinn_amd NegLogPost")
        self.p_unflatten(weights)
        return loss_fn.value_and_grad(weights, inputs, np.asarray(targets), want_grad=True)[1]


def nnwrapper(x, nnmodel):
    """numpy `(N,d)` -> numpy `(N,o)` through the module's current weights (nnwrap.py:306-327)."""
    return NNWrap(nnmodel)(x)


def nn_p(p, x, *otherpars):
    """f_p(x): evaluate the module `otherpars[0]` with flat weights `p` at `x` `(N,d)` -> `(N,o)`."""
    nnw = NNWrap(otherpars[0])
    nnw.p_unflatten(p)
    return nnw._predict(p, x)
