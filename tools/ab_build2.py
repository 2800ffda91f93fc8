#!/usr/bin/env python3
"""A/B variant of the library: recompile ONE translation unit with extra flags, link it with the shipped objects of the
others (quinn_amd/lib/obj, made by quinn_amd._lib.build).  usage: tools/ab_build2.py <name> <file.hip> [-DFLAG ...]
-> quinn_amd/lib/libquinn_amd_<name>.so (select it with QUINN_AMD_LIB=<path>)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quinn_amd import _lib
name, unit, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
_lib.build()
objdir = os.path.join(_lib.LIBDIR, "obj")
obj = os.path.join(objdir, f"{unit[:-4]}_{name}.o")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form", "-fPIC"] + flags +
               ["-c", os.path.join(_lib.CSRC, unit), "-o", obj], check=True)
objs = [obj if s == unit else os.path.join(objdir, s[:-4] + ".o") for s in _lib.SOURCES]
out = os.path.join(_lib.LIBDIR, f"libquinn_amd_{name}.so")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out] + objs, check=True)
print(out)
