#!/usr/bin/env python3
"""A/B variant of the library with SEVERAL translation units recompiled with extra flags:
tools/ab_build3.py <name> a.hip,b.hip [-DFLAG ...] -> quinn_amd/lib/libquinn_amd_<name>.so"""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quinn_amd import _lib
name, units, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
_lib.build()
objdir = os.path.join(_lib.LIBDIR, "obj")
def one(unit):
    obj = os.path.join(objdir, f"{unit[:-4]}_{name}.o")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form", "-fPIC"] + flags +
                   ["-c", os.path.join(_lib.CSRC, unit), "-o", obj], check=True)
    return unit, obj
with ThreadPoolExecutor(max_workers=4) as ex:
    built = dict(ex.map(one, units))
objs = [built.get(s, os.path.join(objdir, s[:-4] + ".o")) for s in _lib.SOURCES]
out = os.path.join(_lib.LIBDIR, f"libquinn_amd_{name}.so")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out] + objs, check=True)
print(out)
