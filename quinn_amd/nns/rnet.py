"""Residual network and its layer-weight parameterisations, in the reference's layout.

Mirror of the reference's `RNet` and `LayerFcn` family (quinn/nns/rnet.py:16-165 and :217-380):
same constructor arguments, attribute names, parameter names and registration order
(`weight_pre, bias_pre, weight_post, bias_post, ww_k, bb_k`), and the same sequence of
`torch.rand` draws at construction, so a seeded model equals the reference's.  The solvers
evaluate it through the device operator (`quinn_amd.ops.RNetArch` -> `qn_rnet_desc_create`);
`forward` below is the plain torch definition of the module.  `final_layer` variants are outside
the accelerated path and raise.
"""
import math

import torch

from .mlp import MLPBase


class LayerFcn():
    """A layer's weight as a function of 'time' t = i / (L+1) and `npar` parameter tensors."""

    def __init__(self):
        self.npar = None

    def __call__(self, pars, t):
        raise NotImplementedError


class Poly(LayerFcn):
    """sum_i pars[i] * t**i  (rnet.py:324-353)."""

    def __init__(self, order):
        super().__init__()
        self.npar = order + 1

    def __call__(self, pars, t):
        assert len(pars) == self.npar
        val = 0.0
        for i in range(self.npar):
            val += pars[i] * t**i
        return val


class Const(LayerFcn):
    """The same weight in every layer (rnet.py:217-241)."""

    def __init__(self):
        super().__init__()
        self.npar = 1

    def __call__(self, pars, t):
        assert len(pars) == self.npar
        return pars[0]


class Lin(LayerFcn):
    def __init__(self):
        super().__init__()
        self.npar = 2

    def __call__(self, pars, t):
        assert len(pars) == self.npar
        return pars[0] + pars[1] * t


class Quad(LayerFcn):
    def __init__(self):
        super().__init__()
        self.npar = 3

    def __call__(self, pars, t):
        assert len(pars) == self.npar
        return pars[0] + pars[1] * t + pars[2] * t**2


class Cubic(LayerFcn):
    def __init__(self):
        super().__init__()
        self.npar = 4

    def __call__(self, pars, t):
        assert len(pars) == self.npar
        return pars[0] + pars[1] * t + pars[2] * t**2 + pars[3] * t**3


class NonPar(LayerFcn):
    """One parameter tensor per layer: pars[int(t * npar)]  (rnet.py:355-380)."""

    def __init__(self, npar):
        super().__init__()
        self.npar = npar

    def __call__(self, pars, t):
        assert len(pars) == self.npar
        return pars[int(t * self.npar)]


class RNet(MLPBase):
    def __init__(self, rdim, nlayers, wp_function=None, indim=None, outdim=None, biasorno=True, nonlin=True,
                 mlp=False, layer_pre=False, layer_post=False, final_layer=None, device='cpu', init_factor=1.0,
                 sum_dim=1):
        super().__init__(indim, outdim, device=device)
        if final_layer is not None:
            raise NotImplementedError("final_layer is outside the MI355X hot path")
        if self.indim is None:
            self.indim = rdim
        if self.outdim is None:
            self.outdim = rdim
        self.rdim = rdim
        self.nlayers = nlayers
        self.biasorno = biasorno
        if wp_function is None:
            wp_function = NonPar(nlayers + 1)
        assert isinstance(wp_function, LayerFcn)
        self.wp_function = wp_function
        self.step_size = 1.0 / (nlayers + 1.0)
        self.mlp = mlp
        self.layer_pre = layer_pre
        self.layer_post = layer_post
        self.final_layer = final_layer
        self.init_factor = init_factor
        self.sum_dim = sum_dim
        if self.indim != rdim:
            assert layer_pre
        if self.outdim != rdim:
            assert layer_post

        def uniform(*shape, fan):
            # U(-1,1) / sqrt(fan) * init_factor, one torch.rand call per tensor (rnet.py:90-118)
            return torch.nn.Parameter(init_factor * (2. * torch.rand(*shape) - 1.) / math.sqrt(fan))
        if layer_pre:
            self.weight_pre = uniform(rdim, self.indim, fan=self.indim)
            self.bias_pre = uniform(rdim, fan=self.indim)
        if layer_post:
            self.weight_post = uniform(self.outdim, rdim, fan=rdim)
            self.bias_post = uniform(self.outdim, fan=rdim)
        for ip in range(wp_function.npar):
            self.register_parameter(name='ww_' + str(ip), param=uniform(rdim, rdim, fan=rdim))
        if biasorno:
            for ip in range(wp_function.npar):
                self.register_parameter(name='bb_' + str(ip), param=uniform(rdim, fan=rdim))
        self.activ = torch.nn.Tanh() if nonlin else torch.nn.Identity()
        self.to(device)

    def forward(self, x):
        F = torch.nn.functional
        out = x + 0.0
        if self.layer_pre:
            out = self.activ(F.linear(out, self.weight_pre, self.bias_pre))
        npar = self.wp_function.npar
        ws = [getattr(self, 'ww_' + str(ip)) for ip in range(npar)]
        bs = [getattr(self, 'bb_' + str(ip)) for ip in range(npar)] if self.biasorno else None
        for i in range(self.nlayers + 1):
            t = self.step_size * i
            w = self.wp_function(ws, t)
            b = self.wp_function(bs, t) if self.biasorno else None
            z = self.activ(F.linear(out, w, b))
            out = z if self.mlp else out + self.step_size * z
        if self.layer_post:
            out = F.linear(out, self.weight_post, self.bias_post)
        return out
