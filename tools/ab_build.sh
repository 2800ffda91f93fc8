#!/bin/bash
# Build an A/B variant of the library with extra -D flags: tools/ab_build.sh <name> [-DFLAG ...]   (every source of quinn_amd/_lib.py:SOURCES
# in one hipcc call: slow; tools/ab_build3.py recompiles chosen translation units only)
# -> quinn_amd/lib/libquinn_amd_<name>.so ; select it with QUINN_AMD_LIB=<path> (quinn_amd/_lib.py)
set -e
name=$1; shift
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -fPIC -shared "$@" \
  -o quinn_amd/lib/libquinn_amd_$name.so $(python3 -c 'from quinn_amd import _lib; print(" ".join("quinn_amd/csrc/" + s for s in _lib.SOURCES))')
echo quinn_amd/lib/libquinn_amd_$name.so
