// Micro-benchmark: does the int8 matrix pipe of gfx950 run BESIDE float64 vector work?
// tools/ubench.hip showed that the f64 / f32 MFMAs share the vector ALUs (kernel time = MFMA + VALU).  The question
// here decides whether a sliced exact-product ("Ozaki") float64 layer on v_mfma_i32_16x16x64_i8 can hide the tanh:
//  1. cycles per i8 MFMA, one and two waves per SIMD;
//  2. wave A = i8 MFMAs, wave B (same SIMD) = v_fma_f64 / integer VALU: does B keep its solo rate?
//  3. ONE wave: each i8 MFMA followed by K independent f64 FMAs (issue cost of an MFMA in a VALU stream);
//  4. issue cost of the slicing / recombination building blocks.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_i8.hip -o tools/ubench_i8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <algorithm>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v4d __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 512;

// roles: 0 = i8 MFMA 16x16x64, 1 = f64 FMA, 2 = int VALU (perm / bfe / lshl_add mix), 3 = idle, 4 = f64 MFMA,
//        5 = f64 tanh-like mix (fma chain + rcp)
template <int ROLE_A, int ROLE_B>
__global__ void k_roles(double* out, long long* cyc, int waves_a) {
    const int wave = threadIdx.x >> 6;
    const int role = wave < waves_a ? ROLE_A : ROLE_B;
    double x = threadIdx.x * 1e-3;
    const int xi = threadIdx.x * 2654435761u;
    v4i ia = {xi, xi ^ 0x55aa55aa, xi + 77, xi * 3}, ib = {xi * 5, xi ^ 0x0f0f0f0f, xi - 9, xi * 7};
    v4i acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (v4i){i, i, i, i};
    double f[16];
    int g[16];
    for (int i = 0; i < 16; ++i) { f[i] = x + i; g[i] = xi + i; }
    v4d dacc[8];
    for (int i = 0; i < 8; ++i) dacc[i] = (v4d){x, x, x, x};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 0) {
        for (int it = 0; it < ITER * 4; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, acc[i], 0, 0, 0);
        }
    } else if (role == 1) {
        for (int it = 0; it < ITER * 8; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) f[i] = fma(f[i], 0.999, 0.001);
        }
    } else if (role == 2) {
        for (int it = 0; it < ITER * 8; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) g[i] = __builtin_amdgcn_perm(g[i], g[(i + 1) & 15], 0x05010400) + (g[i] << 7);
        }
    } else if (role == 4) {
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) dacc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, dacc[i], 0, 0, 0);
        }
    } else if (role == 5) {
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                double t = f[i] * 0.25, t2 = t * t;
                double p = fma(t2, 0.1333, -0.3333);
                p = fma(p * t2, t, t);
                const double d = fma(p, 0.5, 1.0);
                double r = __builtin_amdgcn_rcp(d);
                r = fma(fma(-d, r, 1.0), r, r);
                f[i] = (p + 0.5) * r;
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + dacc[i][0] + dacc[i][3];
    for (int i = 0; i < 16; ++i) s += f[i] + g[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

// ONE wave: each i8 MFMA followed by K independent f64 FMAs (K compile time); INT: integer ops instead
template <int K, int INT>
__global__ void k_interleave(double* out, long long* cyc) {
    double x = threadIdx.x * 1e-3;
    const int xi = threadIdx.x * 2654435761u;
    v4i ia = {xi, xi ^ 0x55aa55aa, xi + 77, xi * 3}, ib = {xi * 5, xi ^ 0x0f0f0f0f, xi - 9, xi * 7};
    v4i acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (v4i){i, i, i, i};
    double f[16];
    int g[16];
    for (int i = 0; i < 16; ++i) { f[i] = x + i; g[i] = xi + i; }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, acc[i], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int j = (i * K + k) & 15;
                if (INT) g[j] = __builtin_amdgcn_perm(g[j], g[(j + 1) & 15], 0x05010400) + (g[j] << 7);
                else f[j] = fma(f[j], 0.999, 0.001);
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += f[i] + g[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

// op cost: 16 independent chains of one op kind
template <int OP>
__global__ void k_op(double* out, long long* cyc) {
    double f[16];
    int e[16], h[16];
    for (int i = 0; i < 16; ++i) { f[i] = 0.25 + (threadIdx.x + i) * 1e-3; e[i] = (threadIdx.x + i) * 2654435761u; h[i] = e[i] ^ 0x1234567; }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) f[i] = (double)e[i] + f[i];                                    // v_cvt_f64_i32 + add
            if (OP == 1) e[i] = __builtin_amdgcn_perm(e[i], h[i], 0x05010400) ^ h[(i + 1) & 15];   // v_perm_b32 + xor
            if (OP == 2) e[i] = (e[i] << 8) + h[i];                                     // v_lshl_add_u32
            if (OP == 3) { e[i] = __builtin_amdgcn_sbfe(e[i], 3, 8) + h[i]; }           // v_bfe_i32 + add
            if (OP == 4) {                                                              // magic-number fixed point: fma + 2 movs
                const double t = fma(f[i], 0x1p46, 0x1.8p52);
                const long long b = __builtin_bit_cast(long long, t);
                e[i] ^= (int)b; h[i] ^= (int)(b >> 32);
            }
            if (OP == 5) f[i] = fma((double)e[i], 0x1p-16, f[i]);                       // cvt + fma
            if (OP == 6) e[i] = __builtin_amdgcn_alignbit(e[i], h[i], 28) & 0x7f7f7f7f;   // v_alignbit + and
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 16; ++i) s += f[i] + e[i] + h[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

static double med(std::vector<long long> v) { std::sort(v.begin(), v.end()); return (double)v[v.size() / 2]; }

int main() {
    double* out; long long* cyc;
    const int NB = 256;
    CK(hipMalloc(&out, NB * 512 * sizeof(double)));
    CK(hipMalloc(&cyc, NB * 8 * sizeof(long long)));
    std::vector<long long> h(NB * 8);
    auto report = [&](const char* name, int waves, int wa, double per) {
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), cyc, NB * waves * sizeof(long long), hipMemcpyDeviceToHost));
        std::vector<long long> a, b;
        for (int blk = 0; blk < NB; ++blk) for (int w = 0; w < waves; ++w) (w < wa ? a : b).push_back(h[blk * waves + w]);
        printf("%-52s A: %8.0f cyc (%.2f/unit)", name, med(a), med(a) / per);
        if (!b.empty()) printf("   B: %8.0f cyc", med(b));
        printf("\n");
        return 0;
    };
    const double NM = ITER * 4.0 * 8;   // i8 MFMAs per wave in role 0
#define RUN(A, B, T, NAME, WAVES, PER) hipLaunchKernelGGL((k_roles<A, B>), dim3(NB), dim3(T), 0, 0, out, cyc, 4); report(NAME, WAVES, 4, PER);
    RUN(0, 3, 256, "MFMA i8 16x16x64 alone, 1 wave/SIMD   [per MFMA]", 4, NM)
    RUN(0, 0, 512, "MFMA i8 x2 waves/SIMD                 [per MFMA]", 8, NM)
    RUN(1, 3, 256, "FMAf64 alone                          [per FMA]", 4, ITER * 8.0 * 16)
    RUN(2, 3, 256, "int perm+lshl_add alone               [per pair]", 4, ITER * 8.0 * 16)
    RUN(5, 3, 256, "f64 tanh-like mix alone               [per element]", 4, ITER * 16.0)
    RUN(0, 1, 512, "A=MFMA i8  B=FMAf64 (same SIMD)", 8, NM)
    RUN(0, 2, 512, "A=MFMA i8  B=int VALU (same SIMD)", 8, NM)
    RUN(0, 5, 512, "A=MFMA i8  B=f64 tanh mix (same SIMD)", 8, NM)
    RUN(4, 3, 256, "MFMA f64 alone                        [per MFMA]", 4, ITER * 8.0)
    RUN(4, 0, 512, "A=MFMA f64  B=MFMA i8 (same SIMD)", 8, ITER * 8.0)
    const double NI = ITER * 8.0;
#define IL(K, F) hipLaunchKernelGGL((k_interleave<K, F>), dim3(NB), dim3(256), 0, 0, out, cyc); report(F ? "1 wave: i8 MFMA + " #K " int pairs [per MFMA]" : "1 wave: i8 MFMA + " #K " f64 FMA [per MFMA]", 4, 4, NI);
    IL(0, 0) IL(1, 0) IL(2, 0) IL(3, 0) IL(4, 0) IL(6, 0) IL(8, 0) IL(2, 1) IL(4, 1) IL(8, 1)
    const char* opn[] = {"v_cvt_f64_i32 + v_add_f64", "v_perm_b32 + v_xor", "v_lshl_add_u32", "v_bfe_i32 + v_add", "fma magic + 2 xor (fixed point)",
                         "v_cvt_f64_i32 + v_fma_f64", "v_alignbit + v_and"};
#define OPX(O) hipLaunchKernelGGL((k_op<O>), dim3(NB), dim3(256), 0, 0, out, cyc); report(opn[O], 4, 4, ITER * 16.0);
    OPX(0) OPX(1) OPX(2) OPX(3) OPX(4) OPX(5) OPX(6)
    return 0;
}
