// Micro-benchmarks on gfx950 that decide the kernel design (run: tools/ubench on the GPU box).
//  1. does f64 MFMA (wave A) overlap with f64 / f32 VALU (wave B) on the same SIMD?
//  2. can ONE wave issue VALU work under its own in-flight f64 MFMA?
//  3. issue cost of the f64 building blocks of tanh.
//  4. accuracy of v_rcp_f64.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <algorithm>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 512;

// role: 0 = MFMA f64, 1 = VALU f64 fma, 2 = VALU f32 fma, 3 = idle, 4 = MFMA f32 16x16x4, 5 = MFMA f32 32x32x2,
//       6 = f32 transcendental mix (exp2 + rcp + fma)
template <int ROLE_A, int ROLE_B>
__global__ void k_roles(double* out, long long* cyc, int waves_a) {
    const int wave = threadIdx.x >> 6;
    const int role = wave < waves_a ? ROLE_A : ROLE_B;
    double x = threadIdx.x * 1e-3;
    v4d acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (v4d){x, x, x, x};
    double f[16];
    float g[16];
    for (int i = 0; i < 16; ++i) { f[i] = x + i; g[i] = (float)(x + i); }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 0) {
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc[i], 0, 0, 0);
        }
    } else if (role == 1) {
        for (int it = 0; it < ITER * 8; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) f[i] = fma(f[i], 0.999, 0.001);
        }
    } else if (role == 2) {
        for (int it = 0; it < ITER * 16; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) g[i] = fmaf(g[i], 0.999f, 0.001f);
        }
    } else if (role == 4) {
        v4f fa[8];
        for (int i = 0; i < 8; ++i) fa[i] = (v4f){(float)x, 0.f, 0.f, 0.f};
        for (int it = 0; it < ITER * 2; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) fa[i] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)x, (float)x, fa[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) g[i] += fa[i][0] + fa[i][3];
    } else if (role == 5) {
        v16f fb[4];
        for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) fb[i][k] = (float)x;
        for (int it = 0; it < ITER * 2; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) fb[i] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)x, (float)x, fb[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) g[i] += fb[i][0] + fb[i][15];
    } else if (role == 6) {
        for (int it = 0; it < ITER * 4; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float e = __builtin_amdgcn_exp2f(g[i] * 0.01f);
                g[i] = fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f) + g[i] * 0.5f;
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += f[i] + g[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

// one wave: each MFMA followed by K independent f64 FMAs (K compile time)
template <int K, int F32>
__global__ void k_interleave(double* out, long long* cyc) {
    double x = threadIdx.x * 1e-3;
    v4d acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (v4d){x, x, x, x};
    double f[16];
    float g[16];
    for (int i = 0; i < 16; ++i) { f[i] = x + i; g[i] = (float)(x + i); }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc[i], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (F32) g[(i * K + k) & 15] = fmaf(g[(i * K + k) & 15], 0.999f, 0.001f);
                else f[(i * K + k) & 15] = fma(f[(i * K + k) & 15], 0.999, 0.001);
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += f[i] + g[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

// op cost: 16 independent chains of one op kind
template <int OP>
__global__ void k_op(double* out, long long* cyc) {
    double f[16];
    int e[16];
    for (int i = 0; i < 16; ++i) { f[i] = 1.0 + (threadIdx.x + i) * 1e-3; e[i] = i - 8; }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) f[i] = fma(f[i], 0.999, 0.001);
            if (OP == 1) f[i] = f[i] * 0.999;
            if (OP == 2) f[i] = f[i] + 0.001;
            if (OP == 3) f[i] = __builtin_amdgcn_rcp(f[i]);
            if (OP == 4) f[i] = __builtin_rint(f[i] * 1.5);
            if (OP == 5) f[i] = __builtin_amdgcn_ldexp(f[i], e[i] & 1);
            if (OP == 6) { e[i] = (int)f[i]; f[i] += 1.0; }
            if (OP == 7) f[i] = f[i] > 1.5 ? 0.5 : f[i] + 0.25;
            if (OP == 8) f[i] = fmin(fabs(f[i]) + 1.0, 20.0);
            if (OP == 9) f[i] = __builtin_copysign(f[i] + 1.0, -1.0);
            if (OP == 10) f[i] = sqrt(f[i]);
            if (OP == 11) { float t = __builtin_amdgcn_exp2f((float)f[i] * 1e-3f); f[i] = f[i] * 0.5 + t; }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 16; ++i) s += f[i] + e[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

__global__ void k_rcp_acc(const double* x, double* y, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = __builtin_amdgcn_rcp(x[i]);
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
        printf("%-44s A: %8.0f cyc (%.2f/unit)", name, med(a), med(a) / per);
        if (!b.empty()) printf("   B: %8.0f cyc", med(b));
        printf("\n");
        return 0;
    };
    const double NM = ITER * 8.0;   // MFMAs per wave in role 0
    hipLaunchKernelGGL((k_roles<0, 3>), dim3(NB), dim3(256), 0, 0, out, cyc, 4);  report("MFMAf64 alone, 1 wave/SIMD  [per MFMA]", 4, 4, NM);
    hipLaunchKernelGGL((k_roles<0, 0>), dim3(NB), dim3(512), 0, 0, out, cyc, 4);  report("MFMAf64 x2 waves/SIMD       [per MFMA]", 8, 4, NM);
    hipLaunchKernelGGL((k_roles<1, 3>), dim3(NB), dim3(256), 0, 0, out, cyc, 4);  report("FMAf64 alone, 1 wave/SIMD   [per FMA]", 4, 4, ITER * 8.0 * 16);
    hipLaunchKernelGGL((k_roles<1, 1>), dim3(NB), dim3(512), 0, 0, out, cyc, 4);  report("FMAf64 x2 waves/SIMD        [per FMA]", 8, 4, ITER * 8.0 * 16);
    hipLaunchKernelGGL((k_roles<2, 3>), dim3(NB), dim3(256), 0, 0, out, cyc, 4);  report("FMAf32 alone, 1 wave/SIMD   [per FMA]", 4, 4, ITER * 16.0 * 16);
    hipLaunchKernelGGL((k_roles<0, 1>), dim3(NB), dim3(512), 0, 0, out, cyc, 4);  report("A=MFMAf64  B=FMAf64 (same SIMD)", 8, 4, NM);
    hipLaunchKernelGGL((k_roles<0, 2>), dim3(NB), dim3(512), 0, 0, out, cyc, 4);  report("A=MFMAf64  B=FMAf32 (same SIMD)", 8, 4, NM);
    hipLaunchKernelGGL((k_roles<1, 2>), dim3(NB), dim3(512), 0, 0, out, cyc, 4);  report("A=FMAf64   B=FMAf32 (same SIMD)", 8, 4, ITER * 8.0 * 16);
    hipLaunchKernelGGL((k_roles<4, 3>), dim3(NB), dim3(256), 0, 0, out, cyc, 4);  report("MFMAf32 16x16x4 alone       [per MFMA]", 4, 4, ITER * 16.0);
    hipLaunchKernelGGL((k_roles<5, 3>), dim3(NB), dim3(256), 0, 0, out, cyc, 4);  report("MFMAf32 32x32x2 alone       [per MFMA]", 4, 4, ITER * 8.0);
    hipLaunchKernelGGL((k_roles<4, 2>), dim3(NB), dim3(512), 0, 0, out, cyc, 4);  report("A=MFMAf32 16x16x4 B=FMAf32 (B alone 385804)", 8, 4, ITER * 16.0);
    hipLaunchKernelGGL((k_roles<5, 2>), dim3(NB), dim3(512), 0, 0, out, cyc, 4);  report("A=MFMAf32 32x32x2 B=FMAf32 (B alone 385804)", 8, 4, ITER * 8.0);
    hipLaunchKernelGGL((k_roles<6, 3>), dim3(NB), dim3(256), 0, 0, out, cyc, 4);  report("f32 tanh-like mix alone      [per element]", 4, 4, ITER * 4.0 * 16);
    hipLaunchKernelGGL((k_roles<4, 6>), dim3(NB), dim3(512), 0, 0, out, cyc, 4);  report("A=MFMAf32 16x16x4 B=f32 tanh mix", 8, 4, ITER * 16.0);
    const double NI = ITER * 4.0;
#define IL(K, F) hipLaunchKernelGGL((k_interleave<K, F>), dim3(NB), dim3(256), 0, 0, out, cyc); report(F ? "1 wave: MFMA + " #K " f32 FMA [per MFMA]" : "1 wave: MFMA + " #K " f64 FMA [per MFMA]", 4, 4, NI);
    IL(0, 0) IL(2, 0) IL(4, 0) IL(8, 0) IL(12, 0) IL(16, 0) IL(4, 1) IL(8, 1) IL(16, 1)
    const char* opn[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_mul+v_rndne_f64", "v_ldexp_f64(+and)", "v_cvt_i32_f64+add",
                         "cmp+cndmask+add", "v_min(|x|+1)", "v_bfi copysign+add", "sqrt f64", "cvt+exp2f32+cvt+fma"};
#define OPX(O) hipLaunchKernelGGL((k_op<O>), dim3(NB), dim3(256), 0, 0, out, cyc); report(opn[O], 4, 4, ITER * 16.0);
    OPX(0) OPX(1) OPX(2) OPX(3) OPX(4) OPX(5) OPX(6) OPX(7) OPX(8) OPX(9) OPX(10) OPX(11)
    // rcp accuracy
    const int n = 1 << 16;
    std::vector<double> hx(n), hy(n);
    for (int i = 0; i < n; ++i) hx[i] = 1.0 + (double)i / n;
    double *dx, *dy;
    CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&dy, n * 8));
    CK(hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_rcp_acc, dim3(n / 256), dim3(256), 0, 0, dx, dy, n);
    CK(hipMemcpy(hy.data(), dy, n * 8, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int i = 0; i < n; ++i) worst = std::max(worst, std::fabs(hy[i] * hx[i] - 1.0));
    printf("v_rcp_f64 max relative error on [1,2): %.3e (= 2^%.1f)\n", worst, std::log2(worst));
    return 0;
}
