"""numpy <-> torch helpers with the reference's contract (quinn/nns/tchutils.py:11-41):
`tch` copies a numpy array / list into a float64 tensor, `npy` brings a tensor back to numpy.
Unlike the reference this module does NOT change torch's global default dtype on import;
float64 is requested explicitly wherever a tensor is created."""
import numpy as np
import torch


def tch(arr, device='cpu', rgrad=False):
    if isinstance(arr, list):
        arr = np.array(arr)
    if isinstance(arr, (float, int)):
        arr = np.array(arr, dtype=np.float64)
    t = torch.tensor(arr, device=device)
    if t.is_floating_point():
        t = t.to(torch.float64)
    return t.requires_grad_(rgrad) if rgrad else t


def npy(arr):
    return arr.detach().cpu().numpy()


def print_nnparams(nnmodel, names_only=False):
    for name, param in nnmodel.named_parameters():
        if names_only:
            print(f"{name}, shape {tuple(param.data.shape)}")
        else:
            print(name, param.data)
