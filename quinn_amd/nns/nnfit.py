"""The training loop -- for M independent ensemble members at once, or for one module with a
custom loss (variational inference).

Mirror of the reference's `nnfit` (quinn/nns/nnfit.py:15-218): same signature, same epoch /
minibatch structure (one `torch.randperm` per epoch; per update the batch loss, the validation
loss and -- on the first minibatch of an epoch -- the full-training loss; best model = lowest
validation loss; Adam / SGD step), same result dict.  Progress table as in the reference; the
loss-curve PNG side effects (nnfit.py:195-216) are not reproduced.

Two execution modes
  * MLP + 'mse' / 'logpost' loss: every member's forward / backward / Adam step is one batched
    kernel launch (`fit_members`); `nnfit` on a single module is the M = 1 case.
  * custom `loss_xy` (NN_VI passes `BNet.viloss`): the reference's loop, driven on the host, with
    the loss evaluated by the HIP kernels inside the callable and `qn_adam_batched` as optimiser.
"""
import copy
import ctypes
import sys

import numpy as np
import torch

from .. import _lib
from ..ops import MLPArch, BatchedMLP, flatten_module


def load_flat_into(module, w):
    """Write flat float64 vector `w` into module.parameters() (reference nnwrap.py:99-104)."""
    s = 0
    with torch.no_grad():
        for p in module.parameters():
            n = p.numel()
            p.copy_(torch.as_tensor(np.asarray(w[s:s + n])).view(p.shape).to(p.dtype))
            s += n


def adam_step(W, G, m, v, lr, step, gscale=1.0, wd=0.0, beta1=0.9, beta2=0.999, eps=1e-8):
    """One Adam step for the rows of W [B,p] (float64 device tensors; G may be float32)."""
    L = _lib.lib()
    B, p = W.shape
    qdt = _lib.QN_F64 if G.dtype == torch.float64 else _lib.QN_F32
    st = ctypes.c_void_p(torch.cuda.current_stream(W.device).cuda_stream)
    with torch.cuda.device(W.device):
        _lib.check(L.qn_adam_batched(W.data_ptr(), G.data_ptr(), m.data_ptr(), v.data_ptr(), lr.data_ptr(), B, p,
                                     qdt, gscale, wd, beta1, beta2, eps, step, st), "qn_adam_batched")


def draw_perms(nmembers, nepochs, ntrn):
    """The `torch.randperm(ntrn)` draws of `nmembers` sequential reference fits of `nepochs`
    epochs each, consumed from torch's GLOBAL CPU generator in the reference's order (member-major:
    member j trains all its epochs before member j+1 starts, nn_ens.py:59-69 -> nnfit.py:126).
    int64 array [nmembers, nepochs, ntrn]."""
    if nmembers * nepochs * ntrn > 4e8:
        raise MemoryError("reference-order permutations would need > 3 GB; pass perm_mode='device'")
    out = np.empty((nmembers, nepochs, ntrn), dtype=np.int64)
    for j in range(nmembers):
        for t in range(nepochs):
            out[j, t] = torch.randperm(ntrn).numpy()
    return out


def fit_members(arch, W0, xtrn, ytrn, rows, xval, yval, nepochs, batch_size, lrate=0.1, wd=0.0,
                optimizer='adam', loss_fn='mse', datanoise=None, lmbd=None, perm_mode='reference',
                device=None, dtype='float64', freq_out=100, verbose=True, perms=None, anchors=None,
                prior_sigma=None, scheduler_lr=None, cooldown=100, factor=0.95):
    """Train M members in lock-step.

    Args:
        arch (MLPArch); W0 [M,p] initial flat weights; xtrn (N,d), ytrn (N,o): the FULL dataset;
        rows [M, ntrn] int: the dataset rows member j trains on; xval, yval: validation set
        shared by all members (None: each member validates on its own rows); perms: optional precomputed [M, nepochs, ntrn] permutations (a shard of
        `draw_perms` when members are split over ranks); anchors [M,p] + prior_sigma: per-member Gaussian
        prior N(anchor, prior_sigma^2) added to the 'logpost' loss as the reference's NegLogPost does
        (losses.py:202-204, weight len(batch)/ntrn; used by NN_RMS); the rest as in `nnfit`.
    Returns:
        dict with per-member arrays: 'best_w' [M,p], 'final_w' [M,p], 'best_loss' [M],
        'best_epoch' [M], 'best_fepoch' [M], 'history' [M, nupdates, 4].
    """
    W0 = np.atleast_2d(np.asarray(W0, dtype=np.float64))
    M, p = W0.shape
    ntrn = np.asarray(rows).shape[-1]
    if batch_size is None or batch_size > ntrn:
        batch_size = ntrn
    if M == 0:      # this rank owns no member (more ranks than members): empty shard for the all_gather
        nupd = nepochs * len(range(0, ntrn, batch_size))
        return {'best_w': np.zeros((0, p)), 'final_w': np.zeros((0, p)), 'best_loss': np.zeros(0),
                'best_epoch': np.zeros(0, dtype=np.int64), 'best_fepoch': np.zeros(0),
                'history': np.zeros((0, nupd, 4))}
    rows = np.asarray(rows).reshape(M, -1)
    o = arch.dims[-1]
    op = BatchedMLP(arch, xtrn, ytrn, device=device, dtype=dtype)
    # xval None: every member validates on its OWN training rows (the reference's nnfit copies the member's
    # subset as the validation set when val is None, nnfit.py:106-109)
    opv = BatchedMLP(arch, xval, yval, device=device, dtype=dtype) if xval is not None else None
    dev = op.device
    nval = opv.N if opv is not None else ntrn
    if loss_fn == 'mse':
        tail = lambda sse, n: sse / (n * o)                       # MSELoss(mean), nnfit.py:59-63
        gscale = lambda n: 1.0 / (n * o)
    elif loss_fn == 'logpost':                                    # NegLogPost without prior, losses.py:197-200
        sig = float(datanoise)
        tail = lambda sse, n: 0.5 * sse / sig ** 2 + (n / 2) * np.log(2 * np.pi) + n * np.log(sig)
        gscale = lambda n: 0.5 / sig ** 2
    else:
        print(f"Loss function {loss_fn} is unknown. Exiting.")
        sys.exit()
    prior = None
    if anchors is not None:
        if loss_fn != 'logpost':
            raise ValueError("a prior needs loss_fn='logpost'")
        A = torch.as_tensor(np.asarray(anchors, dtype=np.float64), device=dev).reshape(M, p)
        sp = float(prior_sigma)
        prior = (A, sp, (p / 2) * np.log(2 * np.pi * sp ** 2))
    if optimizer not in ('adam', 'sgd'):
        print(f"Optimizer {optimizer} is unknown. Exiting.")
        sys.exit()
    if lmbd is None:
        def lmbd(epoch): return 1.0

    W = torch.as_tensor(W0, device=dev).clone()
    m = torch.zeros_like(W)
    v = torch.zeros_like(W)
    rows_d = torch.as_tensor(rows, device=dev, dtype=torch.int64)
    if perm_mode == 'reference':
        perms = torch.as_tensor(draw_perms(M, nepochs, ntrn) if perms is None else perms, device=dev)
    nsub = len(range(0, ntrn, batch_size))
    nupd = nepochs * nsub
    hist = torch.zeros(M, nupd, 4, dtype=torch.float64, device=dev)
    best_loss = torch.full((M,), 1.e+100, dtype=torch.float64, device=dev)
    best_w = W.clone()
    best_epoch = torch.zeros(M, dtype=torch.int64, device=dev)
    best_fepoch = torch.zeros(M, dtype=torch.float64, device=dev)
    rows32 = rows_d.to(torch.int32)
    fepoch, upd, step = 0.0, 0, 0
    loss_full = None
    plateau = None
    if scheduler_lr == "ReduceLROnPlateau":
        plateau = PlateauLR(M, lrate, factor, cooldown, dev)
    elif scheduler_lr is not None:
        raise NotImplementedError(f"scheduler {scheduler_lr!r}")
    for t in range(nepochs):
        lr = plateau.lr if plateau is not None else torch.full((M,), lrate * lmbd(t), dtype=torch.float64, device=dev)
        perm = perms[:, t] if perm_mode == 'reference' else torch.rand(M, ntrn, device=dev).argsort(dim=1)
        for i in range(0, ntrn, batch_size):
            idx = torch.gather(rows_d, 1, perm[:, i:i + batch_size]).to(torch.int32)
            nb = idx.shape[1]
            Wc = W if op.tdt == torch.float64 else W.to(op.tdt)
            sse, g = op.sse_grad(Wc, row_idx=idx)
            loss_trn = tail(sse, nb)
            # Evaluations the reference makes per update (nnfit.py:133-140): minibatch loss (with gradient), validation loss and --
            # on an epoch's first minibatch -- the loss over the member's whole training subset.  Two of them coincide in common
            # settings and are then taken from ONE forward pass: a full-batch step's minibatch IS the training subset (a
            # permutation of the same rows: equal up to summation order, 1e-16 relative), and without a validation set every
            # member validates on its training subset (nnfit.py:106-109).  cfg4 (512 members, full batch): 1 of 3 forwards fewer.
            sse_sub = None                                         # SSE over the member's whole training subset at the current weights
            if nb == ntrn:
                sse_sub = sse
            if opv is not None:
                loss_val = tail(opv.sse(Wc), nval)
            else:
                if sse_sub is None:
                    sse_sub = op.sse(Wc, row_idx=rows32)
                loss_val = tail(sse_sub, nval)
            if i == 0:
                if sse_sub is None:
                    sse_sub = op.sse(Wc, row_idx=rows32)
                loss_full = tail(sse_sub, ntrn)
            gextra = None
            if prior is not None:                                  # NegLogPrior, losses.py:238-256
                A, sp, cst = prior
                dev2 = W - A
                nlp = (dev2 * dev2).sum(dim=1) / 2 / sp ** 2 + cst
                loss_trn = loss_trn + nb * nlp / ntrn
                loss_val = loss_val + nval * nlp / ntrn
                if i == 0:
                    loss_full = loss_full + ntrn * nlp / ntrn
                gextra = (nb / ntrn) * dev2 / sp ** 2
            fepoch += 1. / nsub
            hist[:, upd, 0] = fepoch
            hist[:, upd, 1] = loss_trn
            hist[:, upd, 2] = loss_full
            hist[:, upd, 3] = loss_val
            better = loss_val < best_loss
            best_loss = torch.where(better, loss_val, best_loss)
            best_w = torch.where(better[:, None], W, best_w)
            best_epoch = torch.where(better, torch.full_like(best_epoch, t), best_epoch)
            best_fepoch = torch.where(better, torch.full_like(best_fepoch, fepoch), best_fepoch)
            step += 1
            if gextra is not None:
                g = g.double() * gscale(nb) + gextra
                gs = 1.0
            else:
                gs = gscale(nb)
            if optimizer == 'adam':
                adam_step(W, g, m, v, lr, step, gscale=gs, wd=wd)
            else:
                W.sub_(lr[:, None] * (g.double() * gs + wd * W))
            upd += 1
        if plateau is not None:
            plateau.step(hist[:, upd - 1, 3])                   # scheduler.step(curr_state[3]), nnfit.py:170-172
        if verbose and (t == 0 or (t + 1) % freq_out == 0 or t == nepochs - 1):
            if t == 0:
                print('{:>10} {:>10} {:>12} {:>12} {:>12} {:>18} {:>10}'.format(
                    "NEpochs", "NUpdates", "BatchLoss", "TrnLoss", "ValLoss", "BestLoss (Epoch)", "LrnRate"), flush=True)
            h = hist[:, upd - 1].mean(dim=0).cpu().numpy()
            print(f"{t + 1:>10}{upd:>10}{h[1]:>14.6f}{h[2]:>13.6f}{h[3]:>13.6f}"
                  f"{best_loss.mean().item():>14.6f} ({int(best_epoch.max().item())}){lrate * lmbd(t):>10}", flush=True)
    return {'best_w': best_w.cpu().numpy(), 'final_w': W.cpu().numpy(), 'best_loss': best_loss.cpu().numpy(),
            'best_epoch': best_epoch.cpu().numpy(), 'best_fepoch': best_fepoch.cpu().numpy(),
            'history': hist.cpu().numpy()}


class PlateauLR:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode='min', factor, cooldown) with torch's other defaults
    (patience 10, relative threshold 1e-4, min_lr 0, eps 1e-8), vectorised over M members on the device
    (the reference builds one per member, nnfit.py:91-92)."""

    def __init__(self, M, lr0, factor, cooldown, device, patience=10, threshold=1e-4, eps=1e-8):
        f64 = torch.float64
        self.lr = torch.full((M,), float(lr0), dtype=f64, device=device)
        self.best = torch.full((M,), float('inf'), dtype=f64, device=device)
        self.bad = torch.zeros(M, dtype=torch.int64, device=device)
        self.cool = torch.zeros(M, dtype=torch.int64, device=device)
        self.factor, self.cooldown, self.patience, self.threshold, self.eps = factor, cooldown, patience, threshold, eps

    def step(self, metric):
        better = metric < self.best * (1.0 - self.threshold)
        self.best = torch.where(better, metric, self.best)
        self.bad = torch.where(better, torch.zeros_like(self.bad), self.bad + 1)
        in_cool = self.cool > 0
        self.cool = torch.where(in_cool, self.cool - 1, self.cool)
        self.bad = torch.where(in_cool, torch.zeros_like(self.bad), self.bad)
        reduce = self.bad > self.patience
        new_lr = self.lr * self.factor
        apply = reduce & ((self.lr - new_lr) > self.eps)
        self.lr = torch.where(apply, new_lr, self.lr)
        self.cool = torch.where(reduce, torch.full_like(self.cool, self.cooldown), self.cool)
        self.bad = torch.where(reduce, torch.zeros_like(self.bad), self.bad)


class _FlatAdam:
    """torch.optim.Adam semantics for a list of CUDA float64 parameters, stepped by qn_adam_batched."""

    def __init__(self, params, lr, weight_decay=0.0):
        self.params = [q for q in params]
        self.lr, self.wd, self.t = lr, weight_decay, 0
        self.state = [(torch.zeros_like(q.data).view(1, -1), torch.zeros_like(q.data).view(1, -1)) for q in self.params]
        self.param_groups = [{'lr': lr}]

    def zero_grad(self):
        for q in self.params:
            q.grad = None

    def step(self):
        self.t += 1
        for q, (m, v) in zip(self.params, self.state):
            if q.grad is None:
                continue
            lr = torch.full((1,), self.param_groups[0]['lr'], dtype=torch.float64, device=q.device)
            adam_step(q.data.view(1, -1), q.grad.contiguous().view(1, -1), m, v, lr, self.t, wd=self.wd)


def nnfit(nnmodel, xtrn, ytrn, val=None, loss_fn='mse', loss_xy=None, datanoise=None, wd=0.0, priorparams=None,
          lossparams=None, optimizer='adam', lrate=0.1, lmbd=None, scheduler_lr=None, nepochs=5000, batch_size=None,
          gradcheck=False, cooldown=100, factor=0.95, freq_out=100, freq_plot=1000, lhist_suffix='',
          *, device=None, dtype='float64', perm_mode='reference'):
    """Train `nnmodel` (reference signature and result keys: 'best_fepoch', 'best_epoch',
    'best_loss', 'best_nnmodel', 'history').  `nnmodel` is trained in place, as in the reference."""
    if scheduler_lr == "ReduceLROnPlateau" and lmbd is not None:
        print("Trying to use two schedulers. Exiting.")
        sys.exit()
    anchors, prior_sigma = None, None
    if priorparams is not None:                      # Gaussian prior of NegLogPost (nnfit.py:64-66, losses.py:202-204)
        if loss_fn != 'logpost' or loss_xy is not None:
            raise ValueError("priorparams go with loss_fn='logpost'")
        a = priorparams['anchor']
        anchors = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, dtype=np.float64).reshape(1, -1)
        prior_sigma = float(priorparams['sigma'])
    ntrn = xtrn.shape[0]
    if val is None:
        xval, yval = xtrn.copy(), ytrn.copy()
    else:
        xval, yval = val

    if loss_xy is None:
        arch = MLPArch.from_module(nnmodel)
        res = fit_members(arch, flatten_module(nnmodel)[None, :], xtrn, ytrn, np.arange(ntrn)[None, :], xval, yval,
                          nepochs, batch_size, lrate=lrate, wd=wd, optimizer=optimizer, loss_fn=loss_fn,
                          datanoise=datanoise, lmbd=lmbd, perm_mode=perm_mode, device=device, dtype=dtype,
                          freq_out=freq_out, scheduler_lr=scheduler_lr, cooldown=cooldown, factor=factor,
                          anchors=anchors, prior_sigma=prior_sigma)
        load_flat_into(nnmodel, res['final_w'][0])
        best = copy.deepcopy(nnmodel)
        load_flat_into(best, res['best_w'][0])
        return {'best_fepoch': float(res['best_fepoch'][0]), 'best_epoch': int(res['best_epoch'][0]),
                'best_loss': float(res['best_loss'][0]), 'best_nnmodel': best,
                'history': [list(r) for r in res['history'][0]]}

    # ---- custom loss (VI): the reference's loop with the loss evaluated on the device
    if batch_size is None or batch_size > ntrn:
        batch_size = ntrn
    dev = next(nnmodel.parameters()).device
    if optimizer != 'adam':
        raise NotImplementedError("custom-loss training uses Adam")
    opt = _FlatAdam(nnmodel.parameters(), lr=lrate, weight_decay=wd)
    if lmbd is None:
        def lmbd(epoch): return 1.0
    xt = torch.as_tensor(np.asarray(xtrn), dtype=torch.float64, device=dev)
    yt = torch.as_tensor(np.asarray(ytrn), dtype=torch.float64, device=dev)
    xv = torch.as_tensor(np.asarray(xval), dtype=torch.float64, device=dev)
    yv = torch.as_tensor(np.asarray(yval), dtype=torch.float64, device=dev)
    fit_info = {'best_fepoch': 0, 'best_epoch': 0, 'best_loss': 1.e+100, 'best_nnmodel': nnmodel, 'history': []}
    fepoch = 0
    plateau = PlateauLR(1, lrate, factor, cooldown, dev) if scheduler_lr == "ReduceLROnPlateau" else None
    for t in range(nepochs):
        opt.param_groups[0]['lr'] = float(plateau.lr[0]) if plateau is not None else lrate * lmbd(t)
        permutation = torch.randperm(ntrn)
        nsub = len(range(0, ntrn, batch_size))
        for i in range(0, ntrn, batch_size):
            indices = permutation[i:i + batch_size].to(dev)
            loss_trn = loss_xy(xt[indices, :], yt[indices, :])
            with torch.no_grad():
                loss_val = loss_xy(xv, yv)
            if i == 0:
                with torch.no_grad():
                    loss_trn_full = loss_xy(xt, yt)
            fepoch += 1. / nsub
            crit = loss_val.item()
            fit_info['history'].append([fepoch + 0.0, loss_trn.item(), loss_trn_full.item(), crit])
            if crit < fit_info['best_loss']:
                fit_info['best_loss'] = crit
                fit_info['best_nnmodel'] = copy.deepcopy(nnmodel)
                fit_info['best_fepoch'] = fepoch
                fit_info['best_epoch'] = t
            opt.zero_grad()
            loss_trn.backward()
            opt.step()
        if plateau is not None:
            plateau.step(torch.tensor([fit_info['history'][-1][3]], dtype=torch.float64, device=dev))
        if t == 0:
            print('{:>10} {:>10} {:>12} {:>12} {:>12} {:>18} {:>10}'.format(
                "NEpochs", "NUpdates", "BatchLoss", "TrnLoss", "ValLoss", "BestLoss (Epoch)", "LrnRate"), flush=True)
        if (t + 1) % freq_out == 0 or t == 0 or t == nepochs - 1:
            h = fit_info['history'][-1]
            print(f"{t + 1:>10}{len(fit_info['history']):>10}{h[1]:>14.6f}{h[2]:>13.6f}{h[3]:>13.6f}"
                  f"{fit_info['best_loss']:>14.6f} ({fit_info['best_epoch']}){opt.param_groups[0]['lr']:>10}", flush=True)
    return fit_info
