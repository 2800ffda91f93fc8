#!/bin/bash
# Build an A/B variant of the library with extra -D flags: tools/ab_build.sh <name> [-DFLAG ...]
# -> quinn_amd/lib/libquinn_amd_<name>.so ; select it with QUINN_AMD_LIB=<path> (quinn_amd/_lib.py)
set -e
name=$1; shift
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -fPIC -shared "$@" \
  -o quinn_amd/lib/libquinn_amd_$name.so quinn_amd/csrc/qn_api.hip quinn_amd/csrc/qn_generic.hip quinn_amd/csrc/qn_fused.hip quinn_amd/csrc/qn_fused_i8.hip quinn_amd/csrc/qn_fused_bwd_i8.hip quinn_amd/csrc/qn_wide_i8.hip quinn_amd/csrc/qn_dw_i8.hip quinn_amd/csrc/qn_mcmc.hip quinn_amd/csrc/qn_rnet.hip
echo quinn_amd/lib/libquinn_amd_$name.so
