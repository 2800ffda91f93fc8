"""Multilayer perceptron in the reference's layout.

Mirror of the reference's `MLP` / `MLPBase` (quinn/nns/mlp.py:7-101, quinn/nns/nnbase.py:19-115)
restricted to what the hot path's configurations use: `Sequential(Linear, act, ..., Linear)`
held in `.nnmodel`, float64 parameters, activ in {'tanh','relu', else identity}, default
'relu'.  The dropout / batch-norm / final_transform variants are outside the accelerated
path and raise.
"""
import numpy as np
import torch


class MLPBase(torch.nn.Module):
    def __init__(self, indim, outdim, device='cpu'):
        super().__init__()
        self.indim = indim
        self.outdim = outdim
        self.best_model = None
        self.trained = False
        self.history = None
        self.device = device

    def fit(self, xtrn, ytrn, **kwargs):
        """Train with `nnfit`; keeps the best model and the loss history (nnbase.py:95-115)."""
        from .nnfit import nnfit
        fit_info = nnfit(self, xtrn, ytrn, **kwargs)
        object.__setattr__(self, 'best_model', fit_info['best_nnmodel'])
        self.history = fit_info['history']
        self.trained = True
        return self.best_model

    def numpar(self):
        return sum(p.numel() for p in self.parameters())

    def predict(self, x):
        """numpy `(N,d)` -> numpy `(N,o)` with the best trained weights if any (nnbase.py:61-85), evaluated
        by the device operator."""
        from ..ops import MLPArch, BatchedMLP, flatten_module
        model = self.best_model if self.trained else self
        x = np.asarray(x, dtype=np.float64)
        op = BatchedMLP(MLPArch.from_module(model), x, None)
        return op.predict(flatten_module(model)[None, :], x)[0].double().cpu().numpy()


class MLP(MLPBase):
    def __init__(self, indim, outdim, hls, biasorno=True, activ='relu', bnorm=False, bnlearn=True,
                 dropout=0.0, final_transform=None, device='cpu'):
        super().__init__(indim, outdim, device=device)
        if bnorm or dropout > 0.0 or final_transform is not None:
            raise NotImplementedError("batch-norm / dropout / final_transform are outside the MI355X hot path")
        self.nlayers = len(hls)
        assert self.nlayers > 0
        self.hls = hls
        self.biasorno = biasorno
        self.activ = activ

        def act():
            if activ == 'tanh':
                return torch.nn.Tanh()
            if activ == 'relu':
                return torch.nn.ReLU()
            return torch.nn.Identity()
        widths = (indim,) + tuple(hls) + (outdim,)
        mods = []
        for i in range(len(widths) - 1):
            mods.append(torch.nn.Linear(widths[i], widths[i + 1], bias=biasorno, dtype=torch.float64))
            if i < len(widths) - 2:
                mods.append(act())
        self.nnmodel = torch.nn.Sequential(*mods)

    def forward(self, x):
        return self.nnmodel(x)
