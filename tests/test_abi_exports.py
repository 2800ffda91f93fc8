"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol the header
declares; descriptor / size queries work without a GPU; the product path refuses to run
without one (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from quinn_amd import _lib
from quinn_amd.ops import MLPArch, BatchedMLP

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    _lib.build()
    return _lib.lib()


def test_header_symbols_all_exported(L):
    hdr = open(os.path.join(ROOT, "include", "quinn_amd.h")).read()
    declared = set(re.findall(r"\b(qn_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s
    assert b"gfx950" in L.qn_version()


def test_descriptor_and_sizes(L):
    dims = (ctypes.c_int * 5)(1, 64, 64, 64, 1)
    h = ctypes.c_void_p()
    assert L.qn_mlp_desc_create(dims, 5, 1, 1, ctypes.byref(h)) == 0
    assert L.qn_mlp_num_params(h) == 8513 == MLPArch((1, 64, 64, 64, 1)).nparams
    assert L.qn_workspace_bytes(h, 64, 4096, 1, 0) > 0
    assert L.qn_mlp_path(h, 64, 4096, 0, 0) in (_lib.PATH_GENERIC, _lib.PATH_FUSED)
    assert L.qn_mlp_desc_destroy(h) == 0
    bad = (ctypes.c_int * 2)(3, 0)
    assert L.qn_mlp_desc_create(bad, 2, 1, 1, ctypes.byref(h)) == -1
    assert b"dims" in L.qn_last_error()
    assert L.qn_mlp_desc_create(dims, 5, 7, 1, ctypes.byref(h)) == -1


def test_arch_extraction():
    seq = torch.nn.Sequential(torch.nn.Linear(2, 5), torch.nn.Tanh(), torch.nn.Linear(5, 1))
    a = MLPArch.from_module(seq)
    assert a.dims == (2, 5, 1) and a.activ == "tanh" and a.bias and a.nparams == 21   # numpar()==21
    from quinn_amd.nns.mlp import MLP
    m = MLP(1, 1, (16, 16), activ='tanh')
    assert MLPArch.from_module(m).dims == (1, 16, 16, 1) and m.numpar() == 321
    with pytest.raises(NotImplementedError):
        MLPArch.from_module(torch.nn.Sequential(torch.nn.Linear(2, 2), torch.nn.Sigmoid(), torch.nn.Linear(2, 1)))
    with pytest.raises(NotImplementedError):
        MLPArch.from_module(torch.nn.Conv1d(1, 1, 1))


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback():
    with pytest.raises(_lib.QuinnAmdError):
        BatchedMLP(MLPArch((1, 4, 1)), np.zeros((3, 1)), np.zeros((3, 1)))
