#!/bin/bash
# Host-side AddressSanitizer run: the library with its HOST code instrumented (-Xarch_host -fsanitize=address;
# device code as usual -- GPU sanitizers are not available on this pool), then the CPU tests that drive the C ABI's
# descriptor / size / validation paths under it.  Needs no GPU.  Output: profiles/r03_asan_host.txt
set -e
cd "$(dirname "$0")/.."
RT=$(find /opt/rocm/lib/llvm -name "libclang_rt.asan-x86_64.so" | head -1)
/opt/rocm/bin/hipcc -O1 -g --offload-arch=gfx950 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -fPIC -shared \
  -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer -shared-libsan \
  -o quinn_amd/lib/libquinn_amd_asan.so quinn_amd/csrc/qn_api.hip quinn_amd/csrc/qn_generic.hip quinn_amd/csrc/qn_fused.hip \
  quinn_amd/csrc/qn_fused_i8.hip quinn_amd/csrc/qn_fused_bwd_i8.hip quinn_amd/csrc/qn_wide_i8.hip quinn_amd/csrc/qn_dw_i8.hip quinn_amd/csrc/qn_mcmc.hip quinn_amd/csrc/qn_rnet.hip
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$RT QUINN_AMD_LIB=$PWD/quinn_amd/lib/libquinn_amd_asan.so \
  python3 -m pytest tests/test_abi_exports.py tests/test_host_api_errors.py -q 2>&1 | tee profiles/r03_asan_host.txt
rm -f quinn_amd/lib/libquinn_amd_asan.so
