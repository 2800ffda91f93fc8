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


    def printParams(self):
        """Names and values of the trainable parameters (nnbase.py:118-122)."""
        for name, param in self.named_parameters():
            if param.requires_grad:
                print(name, param.data)

    def printParamNames(self):
        """Names and shapes of the trainable parameters (nnbase.py:125-129)."""
        for name, param in self.named_parameters():
            if param.requires_grad:
                print(name, param.data.shape)

    def predict_plot(self, xx_list, yy_list, labels=None, colors=None, iouts=None):
        """Predicted-vs-data ('diagonal') figure per output, saved as `fitdiag_o<iout>.png` (nnbase.py:132-173;
        the probabilistic counterpart is `QUiNNBase.predict_plot`)."""
        import matplotlib
        matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt
        assert len(xx_list) == len(yy_list)
        preds = [self.predict(xx) for xx in xx_list]
        nset, nout = len(xx_list), preds[0].shape[1]
        labels = labels or [f'Set {i + 1}' for i in range(nset)]
        colors = colors or (['b', 'g', 'r', 'c', 'm', 'y'] * nset)[:nset]
        assert len(labels) == nset and len(colors) == nset
        for iout in (range(nout) if iouts is None else iouts):
            plt.figure(figsize=(10, 10))
            lo = min(float(yy[:, iout].min()) for yy in yy_list)
            hi = max(float(yy[:, iout].max()) for yy in yy_list)
            plt.plot([lo, hi], [lo, hi], 'k--', linewidth=1)
            for pr, yy, lab, col in zip(preds, yy_list, labels, colors):
                plt.plot(yy[:, iout], pr[:, iout], col + 'o', markersize=13, markeredgecolor='w', label=lab)
            plt.xlabel(f'Model output # {iout + 1}')
            plt.ylabel(f'Fit output # {iout + 1}')
            plt.legend()
            plt.savefig(f'fitdiag_o{iout}.png')
            plt.close()

    def plot_1d_fits(self, xx_list, yy_list, domain=None, ngr=111, true_model=None, labels=None, colors=None):
        """One-dimensional slices of the fit through the middle of the domain, one figure per (input, output), saved
        as `fit_d<idim>_o<iout>.png` (nnbase.py:176-237)."""
        import matplotlib
        matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt
        assert len(xx_list) == len(yy_list)
        nset = len(xx_list)
        labels = labels or [f'Set {i + 1}' for i in range(nset)]
        colors = colors or (['b', 'g', 'r', 'c', 'm', 'y'] * nset)[:nset]
        assert len(labels) == nset and len(colors) == nset
        if domain is None:
            xall = np.vstack(xx_list)
            domain = np.stack([xall.min(axis=0), xall.max(axis=0)], axis=1)
        domain = np.asarray(domain, dtype=np.float64)
        ndim, nout = xx_list[0].shape[1], yy_list[0].shape[1]
        for idim in range(ndim):
            unit = np.full((ngr, ndim), 0.5)
            unit[:, idim] = np.linspace(0.0, 1.0, ngr)
            xgrid = domain[:, 0] + unit * (domain[:, 1] - domain[:, 0])
            ygrid = self.predict(xgrid)
            truth = true_model(xgrid, 0.0) if true_model is not None else None
            for iout in range(nout):
                for xx, yy, lab, col in zip(xx_list, yy_list, labels, colors):
                    plt.plot(xx[:, idim], yy[:, iout], col + 'o', markersize=13, markeredgecolor='w', label=lab)
                if truth is not None:
                    plt.plot(xgrid[:, idim], truth[:, iout], 'k-', label='Truth', alpha=0.5)
                plt.plot(xgrid[:, idim], ygrid[:, iout], 'm-', linewidth=5, label='Mean Pred.')
                plt.legend()
                plt.xlabel(f'Input # {idim + 1}')
                plt.ylabel(f'Output # {iout + 1}')
                plt.savefig(f'fit_d{idim}_o{iout}.png')
                plt.clf()


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
