// Micro-benchmark: issue cost (one wave per SIMD) of the building blocks of k_fused_bwd_i8's transposition / exponent /
// recombination phases.  build: hipcc -O3 --offload-arch=gfx950 tools/ubench_xpose.hip -o tools/ubench_xpose
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 2048, NW = 24;

template <int OP>
__global__ void k_op(int* out, long long* cyc, int selA, int selB) {
    extern __shared__ int lds[];
    int g[NW];
    double f[NW / 2];
    const int xi = threadIdx.x * 2654435761u;
    for (int i = 0; i < NW; ++i) g[i] = xi + i * 77;
    for (int i = 0; i < NW / 2; ++i) f[i] = xi * 1e-9 + i;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
        if (OP == 0) {          // NW x v_mov_b32_dpp quad_perm (+ NW v_xor to keep them apart)
#pragma unroll
            for (int i = 0; i < NW; ++i) g[i] = __builtin_amdgcn_mov_dpp(g[i], 0xB1, 0xF, 0xF, true);
        } else if (OP == 1) {   // NW x v_perm_b32 with a VGPR selector
#pragma unroll
            for (int i = 0; i < NW; ++i) g[i] = __builtin_amdgcn_perm(g[i], g[(i + 1) % NW], selA);
        } else if (OP == 2) {   // the 4 x 4 byte transposition, step-major over NW words: 2 DPP + 2 perm per word
            int y[NW], z[NW], w[NW];
#pragma unroll
            for (int i = 0; i < NW; ++i) y[i] = __builtin_amdgcn_mov_dpp(g[i], 0xB1, 0xF, 0xF, true);
#pragma unroll
            for (int i = 0; i < NW; ++i) z[i] = __builtin_amdgcn_perm(y[i], g[i], selA);
#pragma unroll
            for (int i = 0; i < NW; ++i) w[i] = __builtin_amdgcn_mov_dpp(z[i], 0x4E, 0xF, 0xF, true);
#pragma unroll
            for (int i = 0; i < NW; ++i) g[i] = __builtin_amdgcn_perm(w[i], z[i], selB);
        } else if (OP == 3) {   // the same with ds_bpermute in place of the DPP moves
            int y[NW], z[NW], w[NW];
            const int a1 = ((threadIdx.x & 63) ^ 1) * 4, a2 = ((threadIdx.x & 63) ^ 2) * 4;
#pragma unroll
            for (int i = 0; i < NW; ++i) y[i] = __builtin_amdgcn_ds_bpermute(a1, g[i]);
#pragma unroll
            for (int i = 0; i < NW; ++i) z[i] = __builtin_amdgcn_perm(y[i], g[i], selA);
#pragma unroll
            for (int i = 0; i < NW; ++i) w[i] = __builtin_amdgcn_ds_bpermute(a2, z[i]);
#pragma unroll
            for (int i = 0; i < NW; ++i) g[i] = __builtin_amdgcn_perm(w[i], z[i], selB);
        } else if (OP == 4) {   // NW x ds_write_b32 (stride-256 pairs: ds_write2st64), conflict-free addresses
#pragma unroll
            for (int i = 0; i < NW; ++i) lds[(threadIdx.x & 255) + 256 * i] = g[i];
#pragma unroll
            for (int i = 0; i < NW; ++i) g[i] += it;
        } else if (OP == 5) {   // NW x (v_accvgpr_write + v_accvgpr_read)
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                int a;
                asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(g[i]));
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(g[i]) : "a"(a));
            }
        } else if (OP == 6) {   // NW/2 x (v_cvt_f64_i32 + v_fma_f64)
#pragma unroll
            for (int i = 0; i < NW / 2; ++i) f[i] = fma(f[i], 65536.0, (double)g[i]);
        } else if (OP == 7) {   // NW x v_lshl_add_u32
#pragma unroll
            for (int i = 0; i < NW; ++i) g[i] = (g[(i + 1) % NW] << 8) + g[i];
        } else if (OP == 8) {   // NW/2 x (2 v_bfe_u32 + v_max3_u32)
            int m = 0;
#pragma unroll
            for (int i = 0; i < NW; i += 2) m = max(max(m, (g[i] >> 20) & 0x7ff), (g[i + 1] >> 20) & 0x7ff);
            g[0] += m;
        } else if (OP == 9) {   // NW x v_mov_b32 (constants re-materialised)
#pragma unroll
            for (int i = 0; i < NW; ++i) asm volatile("v_mov_b32 %0, 0x43380000" : "=v"(g[i]));
        } else if (OP == 10) {  // NW/2 x v_fma_f64 (reference)
#pragma unroll
            for (int i = 0; i < NW / 2; ++i) f[i] = fma(f[i], 0.999, 0.001);
        } else if (OP == 11) {  // v_fma_f64 with a 32-bit literal operand
#pragma unroll
            for (int i = 0; i < NW / 2; ++i) asm volatile("v_fmac_f64_e32 %0, 0x40f00000, %1" : "+v"(f[i]) : "v"(f[(i + 1) % (NW / 2)]));
        } else if (OP == 12) {  // v_fma_f64 with the constant in an SGPR pair
#pragma unroll
            for (int i = 0; i < NW / 2; ++i) asm volatile("v_fma_f64 %0, %0, %2, %1" : "+v"(f[i]) : "v"(f[(i + 1) % (NW / 2)]), "s"(65536.0));
        } else if (OP == 13) {  // v_fmac_f64 (VOP2) with a literal
#pragma unroll
            for (int i = 0; i < NW / 2; ++i) asm volatile("v_fmac_f64_e32 %0, 0x40f00000, %1" : "+v"(f[i]) : "v"(f[(i + 1) % (NW / 2)]));
        } else if (OP == 14) {  // v_fma_f64, all VGPR
#pragma unroll
            for (int i = 0; i < NW / 2; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[i]) : "v"(f[(i + 1) % (NW / 2)]), "v"(f[(i + 2) % (NW / 2)]));
        } else if (OP == 15) {  // v_fma_f64 with an inline constant
#pragma unroll
            for (int i = 0; i < NW / 2; ++i) asm volatile("v_fma_f64 %0, %0, 2.0, %1" : "+v"(f[i]) : "v"(f[(i + 1) % (NW / 2)]));
        } else if (OP == 16) {  // v_mul_f64 VGPR x VGPR
#pragma unroll
            for (int i = 0; i < NW / 2; ++i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) % (NW / 2)]));
        } else if (OP == 17) {  // v_cvt_f64_i32
#pragma unroll
            for (int i = 0; i < NW / 2; ++i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(f[i]) : "v"(g[i]));
        } else if (OP == 18) {  // v_lshl_add_u32 with an inline shift
#pragma unroll
            for (int i = 0; i < NW; ++i) asm volatile("v_lshl_add_u32 %0, %1, 8, %0" : "+v"(g[i]) : "v"(g[(i + 1) % NW]));
        } else if (OP == 19) {  // v_xor_b32 with a literal
#pragma unroll
            for (int i = 0; i < NW; ++i) asm volatile("v_xor_b32 %0, 0x80808080, %0" : "+v"(g[i]));
        } else if (OP == 20) {  // v_xor_b32 with an SGPR
#pragma unroll
            for (int i = 0; i < NW; ++i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(g[i]) : "s"(0x80808080));
        } else if (OP == 21) {  // v_perm_b32 with an SGPR selector
#pragma unroll
            for (int i = 0; i < NW; ++i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(g[i]) : "v"(g[(i + 1) % NW]), "s"(0x05010400));
        } else if (OP == 22) {  // v_accvgpr_write alone
#pragma unroll
            for (int i = 0; i < NW; ++i) { int a; asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(g[i])); asm volatile("" :: "a"(a)); }
        } else if (OP == 23) {  // v_mov_b32 VGPR -> VGPR
#pragma unroll
            for (int i = 0; i < NW; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(g[i]) : "v"(g[(i + 1) % NW]));
        } else if (OP == 24) {  // v_rcp_f64
#pragma unroll
            for (int i = 0; i < NW / 2; ++i) asm volatile("v_rcp_f64 %0, %0" : "+v"(f[i]));
        } else if (OP == 25) {  // v_min_f64 |x|, sgpr
#pragma unroll
            for (int i = 0; i < NW / 2; ++i) asm volatile("v_min_f64 %0, |%0|, %1" : "+v"(f[i]) : "s"(20.0));
        } else if (OP == 26) {  // v_add_f64 vgpr vgpr
#pragma unroll
            for (int i = 0; i < NW / 2; ++i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) % (NW / 2)]));
        } else if (OP == 27) {  // v_bfi_b32
#pragma unroll
            for (int i = 0; i < NW; ++i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(g[i]) : "s"(0x7fffffff), "v"(g[(i + 1) % NW]));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int i = 0; i < NW; ++i) s += g[i];
    for (int i = 0; i < NW / 2; ++i) s += (int)f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    int* out; long long* cyc;
    const int NB = 256;
    CK(hipMalloc(&out, NB * 256 * sizeof(int)));
    CK(hipMalloc(&cyc, NB * 4 * sizeof(long long)));
    std::vector<long long> h(NB * 4);
    const char* names[] = {"v_mov_b32_dpp quad_perm            [per instr]", "v_perm_b32 (VGPR selector)          [per instr]",
                           "quad transposition, DPP            [per word = 4 instr]", "quad transposition, ds_bpermute    [per word]",
                           "ds_write_b32 (+ v_add)             [per pair]", "v_accvgpr_write + v_accvgpr_read   [per pair]",
                           "v_cvt_f64_i32 + v_fma_f64          [per pair]", "v_lshl_add_u32                     [per instr]",
                           "2 v_bfe_u32 + v_max3_u32           [per triple]", "v_mov_b32 literal                  [per instr]",
                           "v_fma_f64                          [per instr]", "v_mul_f64_e32 v, LITERAL, v", "v_fma_f64 v, v, SGPR, v", "v_fmac_f64_e32 v, LITERAL, v",
                           "v_fma_f64 all VGPR", "v_fma_f64 inline constant", "v_mul_f64 v, v", "v_cvt_f64_i32", "v_lshl_add_u32 inline shift", "v_xor_b32 LITERAL", "v_xor_b32 SGPR",
                           "v_perm_b32 SGPR selector", "v_accvgpr_write", "v_mov_b32 v, v", "v_rcp_f64", "v_min_f64 |v|, s", "v_add_f64 v, v", "v_bfi_b32 s, v, v"};
    const double per[] = {NW, NW, NW, NW, NW, NW, NW / 2, NW, NW / 2, NW, NW / 2, NW / 2, NW / 2, NW / 2, NW / 2, NW / 2, NW / 2, NW / 2, NW, NW, NW, NW, NW, NW, NW / 2, NW / 2, NW / 2, NW};
#define RUN(O) hipLaunchKernelGGL((k_op<O>), dim3(NB), dim3(256), 256 * NW * 4, 0, out, cyc, 0x06020400, 0x05040100); \
    CK(hipDeviceSynchronize()); CK(hipMemcpy(h.data(), cyc, NB * 4 * sizeof(long long), hipMemcpyDeviceToHost)); \
    { std::vector<long long> v(h); std::sort(v.begin(), v.end()); printf("%-58s %8.2f cycles\n", names[O], v[v.size() / 2] / (double)ITER / per[O]); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17) RUN(18) RUN(19) RUN(20) RUN(21) RUN(22) RUN(23) RUN(24) RUN(25) RUN(26) RUN(27)
    return 0;
}
