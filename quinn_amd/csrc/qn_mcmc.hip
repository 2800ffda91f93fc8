// Device-side Metropolis-Hastings step kernels (throughput engine of the AMCMC sampler):
//   qn_mcmc_propose : proposal = current + sqrt(diag) * z + c1 * z0      (Philox4x32-10 normals in-kernel)
//                     or writes the standard normals z for a dense factor product
//   qn_mcmc_accept  : log-posterior from the SSE, accept test, state / best / history / statistics update
// Both read the step counter from DEVICE memory (the accept kernel increments it), so one step is a
// static HIP graph: propose -> batched log-posterior -> accept.  HBM-bound elementwise work:
// propose 3 x 8 B per element, accept <= 6 x 8 B per element.
#include "qn_common.h"
#include <cmath>

namespace {

constexpr int BLK = 256;

struct Philox {
    uint32_t c[4], k[2];
    __device__ __forceinline__ void round() {
        const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
        const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
        const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k[0], n1 = lo1, n2 = hi0 ^ c[3] ^ k[1], n3 = lo0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
    }
    // 4 x 32 random bits for (seed, stream, counter)
    __device__ __forceinline__ void gen(uint64_t seed, uint64_t stream, uint64_t ctr) {
        c[0] = (uint32_t)ctr; c[1] = (uint32_t)(ctr >> 32); c[2] = (uint32_t)stream; c[3] = (uint32_t)(stream >> 32);
        k[0] = (uint32_t)seed; k[1] = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) round();
    }
};

// Philox counter of one draw: [global chain id : 24][purpose : 4][index : 36].  The chain id is GLOBAL
// (chain0 + local index), so a chain's random numbers do not depend on how chains are split over launches /
// ranks.  purposes: 0 elementwise normals (index = column pair), 1 per-chain scalar normal, 2 accept uniform,
// 3 history coefficients (index = row pair)
__device__ __forceinline__ uint64_t ctr_of(int chain, int purpose, uint64_t index) {
    return ((uint64_t)chain << 40) | ((uint64_t)purpose << 36) | (index & 0xFFFFFFFFFull);
}

// uniform in (0, 1) with 53 random bits
__device__ __forceinline__ double u01(uint32_t hi, uint32_t lo) {
    const uint64_t bits = ((uint64_t)hi << 21) ^ (uint64_t)(lo >> 11);        // 53 bits
    return ((double)bits + 0.5) * (1.0 / 9007199254740992.0);
}

// two independent standard normals (Box-Muller) from one Philox block.  Proposal noise does not need float64
// transcendental functions (they made the propose / apply kernels compute-bound, ~9 us per step at cfg2): the
// radius and the angle are formed with the hardware float32 log2 / sin / cos (angle in revolutions), ~1e-6
// relative accuracy, tails to 8 sigma; the accept test keeps its 53-bit uniform.
__device__ __forceinline__ void normal2(const Philox& ph, double& a, double& b) {
    const uint32_t hi = ph.c[0] >> 8;                                                     // 24 bits
    // (0, 1); the lowest of the 2^24 bins is subdivided by 24 more bits so that the tails reach 8 sigma
    const float u1t = hi ? ((float)hi + 0.5f) * (1.0f / 16777216.0f)
                         : ((float)(ph.c[1] >> 8) + 0.5f) * (1.0f / 16777216.0f) * (1.0f / 16777216.0f);
    const float u2 = (float)(ph.c[2] >> 8) * (1.0f / 16777216.0f);                        // [0, 1) revolutions
    const float r = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1t));   // sqrt(-2 ln u1), ln = ln2 * log2
    a = (double)(r * __builtin_amdgcn_cosf(u2));
    b = (double)(r * __builtin_amdgcn_sinf(u2));
}

// stream ids: 2*step -> elementwise normals, 2*step+1 -> per-chain scalars (z0, uniform)
__global__ __launch_bounds__(BLK) void k_propose(const double* __restrict__ cur, const double* __restrict__ sd,
                                                 double c1, int chain0, int64_t p, uint64_t seed,
                                                 const int64_t* __restrict__ step_ptr, double* __restrict__ out) {
    const int b = blockIdx.y;
    const uint64_t step = (uint64_t)*step_ptr;
    double z0 = 0.0;
    if (c1 != 0.0) {
        Philox ph;
        ph.gen(seed, 2 * step + 1, ctr_of(chain0 + b, 1, 0));
        double dummy;
        normal2(ph, z0, dummy);
    }
    const int64_t npair = (p + 1) / 2;
    for (int64_t j = (int64_t)blockIdx.x * BLK + threadIdx.x; j < npair; j += (int64_t)gridDim.x * BLK) {
        Philox ph;
        ph.gen(seed, 2 * step, ctr_of(chain0 + b, 0, (uint64_t)j));
        double za, zb;
        normal2(ph, za, zb);
        const int64_t e0 = (int64_t)b * p + 2 * j;
        if (cur) {
            out[e0] = fma(c1, z0, fma(sd[e0], za, cur[e0]));          // explicit: the fused accept kernel must match
            if (2 * j + 1 < p) out[e0 + 1] = fma(c1, z0, fma(sd[e0 + 1], zb, cur[e0 + 1]));
        } else {
            out[e0] = za;
            if (2 * j + 1 < p) out[e0 + 1] = zb;
        }
    }
}

// Adapted proposal drawn in SAMPLE SPACE.  The reference's proposal covariance after an adaptation at step i
// is c (cov_i + 1e-8 I), cov_i = unbiased sample covariance of x_0..x_i (admcmc.py:52-67).  With the distinct
// states x_k of that history, their multiplicities w_k, the mean m and n = i + 1,
//     delta = sqrt(c / (n-1)) * sum_k sqrt(w_k) u_k (x_k - m)  +  sqrt(c 1e-8) v,    u_k, v_j iid N(0, 1)
// has exactly that covariance -- no p x p matrix, no factorisation: a K x p GEMV over the stored states
// (K = accepted moves so far; float32, shifted by x_0).  One chain per blockIdx.y, 2 columns per thread;
// the coefficients sqrt(w_k) u_k are generated per block into LDS, KC at a time.
// History rows are HALF precision since round 4: row = (x_k - ref) * S as float16, ref [C, p] float64 a per-chain reference
// point (the start; after a compression of the history the weighted mean of the compressed states) and S [C] a per-chain power
// of two that keeps the rows' magnitudes around 2^10 (clamped to +-65504).  Half the bytes of the float32 rows for the kernel
// that streams the whole history (k_hist_block16), and its products run on the float16 matrix cores at 16 x the float32 MFMA
// rate -- the float32 version of that kernel was bound by the matrix pipe, not by HBM (139 GFLOP per block of 64 steps at cfg2
// against 157 TFLOP/s).  Relative rounding 2^-11 of |x - ref|: the proposal stays symmetric (its covariance is that of the
// ROUNDED states), and once a chain is stationary around its reference that is 5e-4 of the posterior spread itself.
typedef _Float16 half_t;
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
__device__ __forceinline__ half_t to_half_sat(double v) {
    const float f = (float)v;
    return (half_t)fminf(fmaxf(f, -65504.f), 65504.f);                 // (NaN passes through)
}
constexpr int KC = 2048;
struct HistArgs {
    int chain0, kcap;
    int64_t p, pstride;
    double s_lr, s_iso;
    uint64_t seed;
    const double* hscale;        // [C] S per chain, or NULL (1)
};
__global__ __launch_bounds__(BLK) void k_propose_hist(HistArgs a, const double* __restrict__ cur,
                                                      const half_t* __restrict__ hist, const float* __restrict__ wsnap,
                                                      const int32_t* __restrict__ ksnap,
                                                      const double* __restrict__ msnap,
                                                      const int64_t* __restrict__ step_ptr, double* __restrict__ out) {
    __shared__ float coef[KC];
    const int b = blockIdx.y;
    const uint64_t step = (uint64_t)*step_ptr;
    const int K = ksnap[b] < a.kcap ? ksnap[b] : a.kcap;
    const int64_t j = 2 * ((int64_t)blockIdx.x * BLK + threadIdx.x);          // first of this thread's 2 columns
    const bool live = j < a.p;
    const half_t* hcol = hist + (int64_t)b * a.kcap * a.pstride + (live ? j : 0);
    const float* wrow = wsnap + (int64_t)b * a.kcap;
    const double inv_s = a.hscale ? 1.0 / a.hscale[b] : 1.0;
    double acc0 = 0.0, acc1 = 0.0, sA = 0.0;
    for (int k0 = 0; k0 < K; k0 += KC) {
        const int kn = K - k0 < KC ? K - k0 : KC;
        for (int kk = 2 * threadIdx.x; kk < kn; kk += 2 * BLK) {
            Philox ph;
            ph.gen(a.seed, 2 * step + 1, ctr_of(a.chain0 + b, 3, (uint64_t)((k0 + kk) >> 1)));
            double za, zb;
            normal2(ph, za, zb);
            // (rounded to float16: the very coefficients the block kernel multiplies with)
            coef[kk] = (float)(half_t)(wrow[k0 + kk] * (float)za);
            if (kk + 1 < kn) coef[kk + 1] = (float)(half_t)(wrow[k0 + kk + 1] * (float)zb);
        }
        __syncthreads();
        const half_t* h = hcol + (int64_t)k0 * a.pstride;
        int kk = 0;
        for (; kk + 4 <= kn; kk += 4) {                             // 4 independent loads in flight
            const v2h h0 = *reinterpret_cast<const v2h*>(h + (int64_t)(kk + 0) * a.pstride);
            const v2h h1 = *reinterpret_cast<const v2h*>(h + (int64_t)(kk + 1) * a.pstride);
            const v2h h2 = *reinterpret_cast<const v2h*>(h + (int64_t)(kk + 2) * a.pstride);
            const v2h h3 = *reinterpret_cast<const v2h*>(h + (int64_t)(kk + 3) * a.pstride);
            const double c0 = coef[kk], c1 = coef[kk + 1], c2 = coef[kk + 2], c3 = coef[kk + 3];
            acc0 = fma(c0, (double)h0.x, acc0); acc1 = fma(c0, (double)h0.y, acc1);
            acc0 = fma(c1, (double)h1.x, acc0); acc1 = fma(c1, (double)h1.y, acc1);
            acc0 = fma(c2, (double)h2.x, acc0); acc1 = fma(c2, (double)h2.y, acc1);
            acc0 = fma(c3, (double)h3.x, acc0); acc1 = fma(c3, (double)h3.y, acc1);
            sA += (c0 + c1) + (c2 + c3);
        }
        for (; kk < kn; ++kk) {
            const v2h h0 = *reinterpret_cast<const v2h*>(h + (int64_t)kk * a.pstride);
            const double c0 = coef[kk];
            acc0 = fma(c0, (double)h0.x, acc0); acc1 = fma(c0, (double)h0.y, acc1);
            sA += c0;
        }
        __syncthreads();
    }
    if (!live) return;
    Philox ph;
    ph.gen(a.seed, 2 * step, ctr_of(a.chain0 + b, 0, (uint64_t)(j >> 1)));
    double za, zb;
    normal2(ph, za, zb);
    const int64_t e0 = (int64_t)b * a.p + j;
    out[e0] = cur[e0] + a.s_lr * (acc0 * inv_s - sA * msnap[e0]) + a.s_iso * za;
    if (j + 1 < a.p) out[e0 + 1] = cur[e0 + 1] + a.s_lr * (acc1 * inv_s - sA * msnap[e0 + 1]) + a.s_iso * zb;
}

// ---- the same draw for TB consecutive steps at once.  delta_t does not depend on the chain's state, only on
// the frozen snapshot and on the step's random numbers, so the increments of the next TB steps can be formed
// in ONE pass over the stored states: HBM traffic per step drops by TB and the kernel becomes a small
// (TB x K) . (K x p) product per chain.  Random numbers are keyed by the absolute step exactly as in
// k_propose_hist, so the increments do not depend on how steps are grouped.
#ifndef QN_TB
#define QN_TB 64
#endif
constexpr int TB = QN_TB;        // steps per block
static_assert(TB % 32 == 0, "k_hist_block16 works in tiles of 32 steps");
constexpr int KB2 = 128;         // history rows per LDS chunk of coefficients
constexpr int CSP = KB2 + 8;     // halves per step row of the chunk in LDS (stride = 4 banks mod 64: conflict-free 16-byte reads)
struct HistBlockArgs {
    int chain0, kcap, kstride;   // coef [C][TB][kstride] float16 (row index fastest), kstride % 8 == 0
    int64_t p, pstride, step0;
    const int64_t* step_ptr;     // non-NULL: the block starts at the CURRENT device step counter (static launch)
    double s_lr, s_iso;
    uint64_t seed;
    const double* hscale;        // [C] S per chain, or NULL (1)
};
// coef[c][t][k] = float16(sqrt(w_k) * u_k^(step0 + t)); entries k in [K, kstride) are zeroed (the product kernel reads whole chunks)
__global__ __launch_bounds__(BLK) void k_hist_coef(HistBlockArgs a, const float* __restrict__ wsnap,
                                                   const int32_t* __restrict__ ksnap, half_t* __restrict__ coef) {
    const int b = blockIdx.z, t = blockIdx.y;
    const int K = ksnap[b] < a.kcap ? ksnap[b] : a.kcap;
    const int kk = 2 * (blockIdx.x * BLK + threadIdx.x);
    if (kk >= a.kstride) return;
    half_t* cb = coef + ((int64_t)b * TB + t) * a.kstride;
    v2h v = {(half_t)0.f, (half_t)0.f};
    if (kk < K) {
        Philox ph;
        const int64_t step0 = a.step_ptr ? *a.step_ptr : a.step0;
        ph.gen(a.seed, 2 * (uint64_t)(step0 + t) + 1, ctr_of(a.chain0 + b, 3, (uint64_t)(kk >> 1)));
        double za, zb;
        normal2(ph, za, zb);
        v.x = (half_t)(wsnap[(int64_t)b * a.kcap + kk] * (float)za);
        if (kk + 1 < K) v.y = (half_t)(wsnap[(int64_t)b * a.kcap + kk + 1] * (float)zb);
    }
    *reinterpret_cast<v2h*>(cb + kk) = v;
}
// delta[c][t][:] = s_lr * (sum_k coef[c][t][k] hist[c][k][:] / S_c - (sum_k coef[c][t][k]) mean[c][:])   (the isotropic part is
// added per step by k_apply_delta): per chain a (TB x K) . (K x p) GEMM on the float16 matrix cores
// (v_mfma_f32_32x32x16_f16, float32 accumulation).  A wave owns 128 columns (4 tiles of 32; tile i holds columns 4 n + i, so a
// lane's history operand for one row is ONE 8-byte load, and the 32 lanes of a half wave read 256 contiguous bytes) and all TB
// steps; one MFMA k-step is 16 history rows, lanes 0-31 rows 0-7 and lanes 32-63 rows 8-15: a lane loads its 8 rows x 4
// columns and transposes them in registers (16 v_perm) into the four tiles' B operands (8 consecutive rows of one column).
// Coefficients: the chunk [TB][KB2] sits in LDS row-index-fastest, a lane's A operand (step n, 8 consecutive rows) is one
// 16-byte read.  The coefficient sums (mean term) are accumulated by the first TB threads while the chunks pass through LDS
// -- every workgroup of a chain sees all of them, in the same order.
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(BLK, 2) void k_hist_block16(HistBlockArgs a, const half_t* __restrict__ hist,
                                                         const half_t* __restrict__ coef,
                                                         const int32_t* __restrict__ ksnap,
                                                         const double* __restrict__ msnap,
                                                         const int32_t* __restrict__ order,
                                                         double* __restrict__ delta) {
    constexpr int MT = TB / 32;          // tiles of 32 steps
    constexpr int NTL = 4;               // tiles of 32 columns per wave
    __shared__ __attribute__((aligned(16))) half_t cs[TB * CSP];
    __shared__ float csum_s[TB];
    // chains are dispatched in the caller's order (longest history first): a workgroup's time is proportional to
    // its chain's K, which differs up to 3x between chains, and the hardware hands out workgroups in grid order
    const int b = order ? order[blockIdx.y] : blockIdx.y;
    const int K = ksnap[b] < a.kcap ? ksnap[b] : a.kcap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l32 = lane & 31, hi = lane >> 5;
    const int64_t j = (((int64_t)blockIdx.x * (BLK / 64) + wave) * 32 + l32) * NTL;    // this lane's 4 columns
    const bool live = j < a.pstride;                                                   // pstride % 4 == 0
    const half_t* hcol = hist + (int64_t)b * a.kcap * a.pstride + (live ? j : 0) + (int64_t)(8 * hi) * a.pstride;
    const half_t* cb = coef + (int64_t)b * TB * a.kstride;
    v16f acc[MT][NTL];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < NTL; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][i][r] = 0.f;
    float csacc = 0.f;
    const uint2 zero2 = make_uint2(0u, 0u);
    auto load8 = [&](const half_t* h, int kbase, int kn, uint2 (&hv)[8]) {              // rows kbase + 8 hi + i (i < 8) of this lane's 4 columns
#pragma unroll
        for (int i = 0; i < 8; ++i)
            hv[i] = (live && kbase + 8 * hi + i < kn) ? *reinterpret_cast<const uint2*>(h + (int64_t)(kbase + i) * a.pstride) : zero2;
    };
    for (int k0 = 0; k0 < K; k0 += KB2) {
        const int kn = K - k0 < KB2 ? K - k0 : KB2;
        const int knp = (kn + 15) & ~15;                                    // rows padded to a whole MFMA step: zero coefficients
        // stage the chunk: TB rows of KB2 coefficients, 16 bytes per thread and pass (coef is zero beyond K up to kstride; beyond
        // kstride the chunk is zero-filled here)
        for (int e = threadIdx.x; e < TB * (KB2 / 8); e += BLK) {
            const int t = e / (KB2 / 8), k8 = 8 * (e % (KB2 / 8));
            const uint4 v = k0 + k8 < a.kstride ? *reinterpret_cast<const uint4*>(cb + (int64_t)t * a.kstride + k0 + k8) : make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(cs + t * CSP + k8) = v;
        }
        __syncthreads();
        if (threadIdx.x < TB) {                                              // mean term: sum of the step's coefficients, rows ascending
            const half_t* c = cs + threadIdx.x * CSP;
            for (int k8 = 0; k8 < knp; k8 += 8) {
                const v8h v = *reinterpret_cast<const v8h*>(c + k8);
#pragma unroll
                for (int i = 0; i < 8; ++i) csacc += (float)v[i];
            }
        }
        const half_t* h = hcol + (int64_t)k0 * a.pstride;
        uint2 hn[8];
        load8(h, 0, kn, hn);
        for (int kk = 0; kk < knp; kk += 16) {
            uint2 hv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) hv[i] = hn[i];
            if (kk + 16 < knp) load8(h, kk + 16, kn, hn);
            // transpose 8 rows x 4 columns -> 4 tiles x 8 rows: column c of row i is half (c & 1) of dword (c >> 1) of hv[i]
            v8h bt[NTL];
#pragma unroll
            for (int c = 0; c < NTL; ++c) {
                unsigned d[4];
#pragma unroll
                for (int r2 = 0; r2 < 4; ++r2) {
                    const unsigned lo = (c >> 1) ? hv[2 * r2].y : hv[2 * r2].x, up = (c >> 1) ? hv[2 * r2 + 1].y : hv[2 * r2 + 1].x;
                    d[r2] = (c & 1) ? __builtin_amdgcn_perm(up, lo, 0x07060302) : __builtin_amdgcn_perm(up, lo, 0x05040100);
                }
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                const v4u dv = {d[0], d[1], d[2], d[3]};
                bt[c] = __builtin_bit_cast(v8h, dv);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const v8h av = *reinterpret_cast<const v8h*>(cs + (32 * m + l32) * CSP + kk + 8 * hi);
#pragma unroll
                for (int c = 0; c < NTL; ++c) acc[m][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bt[c], acc[m][c], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < TB) csum_s[threadIdx.x] = csacc;
    __syncthreads();
    if (j >= a.p) return;
    // C/D layout: lane holds column (lane & 31), rows 8 (r >> 2) + 4 (lane >> 5) + (r & 3)
    const double inv_s = a.hscale ? 1.0 / a.hscale[b] : 1.0;
    const double* mrow = msnap + (int64_t)b * a.p;
    const double mj0 = mrow[j], mj1 = mrow[j + 1 < a.p ? j + 1 : j], mj2 = mrow[j + 2 < a.p ? j + 2 : j],
                 mj3 = mrow[j + 3 < a.p ? j + 3 : j];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = 32 * m + 8 * (r >> 2) + 4 * hi + (r & 3);
            const double sA = (double)csum_s[t];
            double* d = delta + ((int64_t)b * TB + t) * a.p + j;
            d[0] = a.s_lr * ((double)acc[m][0][r] * inv_s - sA * mj0);
            if (j + 1 < a.p) d[1] = a.s_lr * ((double)acc[m][1][r] * inv_s - sA * mj1);
            if (j + 2 < a.p) d[2] = a.s_lr * ((double)acc[m][2][r] * inv_s - sA * mj2);
            if (j + 3 < a.p) d[3] = a.s_lr * ((double)acc[m][3][r] * inv_s - sA * mj3);
        }
}
// out[c][:] = cur[c][:] + delta[c][t][:] + s_iso * v,  v ~ N(0, I) on the stream of the current step
__global__ __launch_bounds__(BLK) void k_apply_delta(const double* __restrict__ cur, const double* __restrict__ delta,
                                                     int t, int chain0, int64_t p, double s_iso, uint64_t seed,
                                                     const int64_t* __restrict__ step_ptr, double* __restrict__ out) {
    const int b = blockIdx.y;
    const uint64_t step = (uint64_t)*step_ptr;
    const int64_t jp = (int64_t)blockIdx.x * BLK + threadIdx.x;       // column pair
    const int64_t j = 2 * jp;
    if (j >= p) return;
    Philox ph;
    ph.gen(seed, 2 * step, ctr_of(chain0 + b, 0, (uint64_t)jp));
    double za, zb;
    normal2(ph, za, zb);
    const double* d = delta + ((int64_t)b * TB + t) * p + j;
    const int64_t e0 = (int64_t)b * p + j;
    out[e0] = fma(s_iso, za, cur[e0] + d[0]);
    if (j + 1 < p) out[e0 + 1] = fma(s_iso, zb, cur[e0 + 1] + d[1]);
}

struct AcceptArgs {
    double half_inv_sig2, lp_const;       // log-posterior = -(half_inv_sig2 * sse + lp_const)
    int chain0, nmcmc, kcap;
    int nparts;                           // sse_prop is [C, nparts]: partial sums, added left to right
    int par, C;                           // parity of this step: scalar state is read from slot par, written to slot 1 - par
    int64_t p, pstride;
    uint64_t seed;
    const double* hscale;                 // [C] scale S of the chain's history rows (row = float16((x - x0) * S)), or NULL (1)
    // HMC (qn_hmc_accept): kinetic energies K = (sum of nkin partial sums, left to right) / 2 enter the MH ratio,
    // and an accepted proposal also carries its gradient row (gprop -> gcur).  nkin == 0: plain Metropolis.
    int nkin;
    const double* kin_cur;                // [C, nkin]
    const double* kin_prop;               // [C, nkin]
    const double* gprop;                  // [C, p]
    double* gcur;                         // [C, p]
};

// Optional fusion of the NEXT step's proposal into the accept kernel (it already has the new state in registers):
// the same formulas and the same random numbers (streams of step + 1) as k_propose / k_apply_delta, so a fused run
// is bit-identical to an unfused one.  mode 0: none; 1: initial proposal cur + sd z + c1 z0; 2: cur + delta[t] + s_iso z.
struct NextArgs {
    int mode, t;
    double c1, s_iso;
    const double* sd;
    const double* delta;
    double* out;
};

// A chain's elements are spread over gridDim.x workgroups of ABLK threads; a thread takes APT PAIRS of consecutive elements
// per pass, pair = one Philox block and ONE 16-byte access (consecutive lanes: consecutive pairs, so every access of the
// wave is a contiguous 1 KB; a chain's row starts at an odd multiple of 8 bytes for odd b * p: the accesses are declared
// 8-byte aligned, which the hardware's unaligned mode serves).  The kernel is a latency chain more than a bandwidth
// problem: what counts is the number of DEPENDENT memory round trips between launch and last store, then the bytes, then the
// vector instructions (Philox) per element.  Round 3: scalar loads -> exp -> element loads -> stores, 52 B of traffic per
// element and step, one Philox block per ELEMENT (13.2 us per step at cfg2, 64 x 8513 elements).  Now:
//   * every load a REJECTING chain needs -- the chain's scalars and, per element, current state, proposal (speculatively) and
//     the next proposal's ingredient -- is issued before the first result is used: ONE round trip, then the decision, then
//     the stores (chain row, next proposal);
//   * an ACCEPTING chain pays a second round trip for what only it needs (x0, the sum, the multiplicity of the state it
//     leaves) and stores the new state, the history row and the sum;
//   * the sum of (x - x0) over the samples is LAZY: a state's stay is added when the chain leaves it, mult x (x - x0) at once,
//     instead of one read-modify-write of the whole vector per step; the caller adds the current state's stay when it needs
//     the total (an adaptation: once per `tadapt` steps).  A rejecting chain moves 40 B per element.
// Every workgroup of a chain takes the accept decision itself, from the chain's scalar state (current / best log-posterior,
// number of stored states, step counter), which is DOUBLE-BUFFERED by the parity of the step: all workgroups read slot `par`,
// workgroup 0 of the chain writes slot 1 - par, so no workgroup can see a half-updated state and no fence, atomic or arrival
// counter is needed.
constexpr int ABLK = 256;
constexpr int APT = 2;                    // pairs per thread and pass
constexpr int APARTS_MAX = 64;
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));      // two doubles at an 8-byte aligned address
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ d2u ld2(const double* p, bool both) {
    if (both) return *reinterpret_cast<const d2u*>(p);
    d2u v; v.x = p[0]; v.y = 0.0; return v;                          // (the last element of an odd-length row)
}
__device__ __forceinline__ void st2(double* p, double x, double y, bool both) {
    if (both) { d2u v; v.x = x; v.y = y; *reinterpret_cast<d2u*>(p) = v; } else p[0] = x;
}
__global__ __launch_bounds__(ABLK) void k_accept(AcceptArgs a, const double* __restrict__ prop,
                                                const double* __restrict__ sse_prop, double* __restrict__ cur,
                                                double* __restrict__ cur_lp, double* __restrict__ best,
                                                double* __restrict__ best_lp, double* __restrict__ chain,
                                                double* __restrict__ lps, double* __restrict__ alphas,
                                                int64_t* __restrict__ nacc, const double* __restrict__ x0,
                                                half_t* __restrict__ hist, int32_t* __restrict__ mult,
                                                int32_t* __restrict__ kcur, double* __restrict__ sumx,
                                                int64_t* __restrict__ step_ptr, NextArgs nx) {
    const int b = blockIdx.y, part = blockIdx.x, lane = threadIdx.x & 63;
    const int so = a.par * a.C + b, sn = (1 - a.par) * a.C + b;     // old / new slot of the per-chain scalars
    const int64_t base = (int64_t)b * a.p;
    const int64_t npair = (a.p + 1) / 2;
    const int64_t dbase = nx.mode == 2 ? ((int64_t)b * TB + nx.t) * a.p : 0;
    d2u pv[APT], cv[APT], iv[APT], gv[APT];
    auto load = [&](int64_t pair0) {
#pragma unroll
        for (int h = 0; h < APT; ++h) {
            const int64_t pair = pair0 + (int64_t)h * ABLK, e = 2 * pair;
            const bool in = pair < npair, both = e + 1 < a.p;
            const d2u zero = {0.0, 0.0};
            pv[h] = in ? ld2(prop + base + e, both) : zero;
            cv[h] = in ? ld2(cur + base + e, both) : zero;
            iv[h] = !in ? zero : nx.mode == 1 ? ld2(nx.sd + base + e, both) : nx.mode == 2 ? ld2(nx.delta + dbase + e, both) : zero;
            gv[h] = (in && a.gcur) ? ld2(a.gprop + base + e, both) : zero;
        }
    };
    // ---- issue: the thread's first pairs, then the chain's scalars (partial sums one per lane: ONE round trip for up to 64)
    int64_t pair0 = (int64_t)part * (APT * ABLK) + threadIdx.x;
    load(pair0);
    const int64_t step = step_ptr[a.par];
    const double sp = lane < a.nparts ? sse_prop[(int64_t)b * a.nparts + lane] : 0.0;
    const double kc_l = lane < a.nkin ? a.kin_cur[(int64_t)b * a.nkin + lane] : 0.0;
    const double kp_l = lane < a.nkin ? a.kin_prop[(int64_t)b * a.nkin + lane] : 0.0;
    const double clp = cur_lp[so];
    const double blp = best_lp[so];
    const int kc = hist ? kcur[so] : 0;
    // ---- the decision (every wave of every workgroup of the chain computes the same numbers in the same order)
    double sse = 0.0;
    for (int i = 0; i < a.nparts && i < 64; ++i) sse += __shfl(sp, i, 64);            // left to right: order of k_sum_partials
    for (int i = 64; i < a.nparts; ++i) sse += sse_prop[(int64_t)b * a.nparts + i];   // (more than 64 parts: not at the sizes in use)
    double kin_c = 0.0, kin_p = 0.0;
    for (int i = 0; i < a.nkin; ++i) { kin_c += __shfl(kc_l, i, 64); kin_p += __shfl(kp_l, i, 64); }      // (nkin <= 64)
    const double plp = -(a.half_inv_sig2 * sse + a.lp_const);
    // exp(current_H - proposed_H), H = U + K, U = -log-posterior (mcmc.py:69-72); with K = 0 this is exp(plp - clp) exactly
    const double mh = exp((-clp + 0.5 * kin_c) - (-plp + 0.5 * kin_p));
    Philox ph;
    ph.gen(a.seed, 2 * (uint64_t)step + 1, ctr_of(a.chain0 + b, 2, 0));
    const double u = u01(ph.c[0], ph.c[1]);
    const bool take = u < mh;                                       // NaN -> reject, inf -> accept, as `u < mh_prob`
    const double nlp = take ? plp : clp;
    const bool better = take && nlp >= blp;
    double z0n = 0.0;
    if (nx.mode == 1 && nx.c1 != 0.0) {
        Philox pz;
        pz.gen(a.seed, 2 * (uint64_t)(step + 1) + 1, ctr_of(a.chain0 + b, 1, 0));
        double dummy;
        normal2(pz, z0n, dummy);
    }
    double* crow = chain ? chain + ((int64_t)b * (a.nmcmc + 1) + step + 1) * a.p : nullptr;
    // history of DISTINCT states (shifted by x0, float32) with multiplicities, and the sum of the stays the chain has left:
    // what the adapted proposal is drawn from (k_propose_hist)
    const int knew = take ? kc + 1 : kc;
    const bool leave = hist && take;                                // the chain leaves state kc: its stay enters the sum
    half_t* hrow = (leave && knew < a.kcap) ? hist + ((int64_t)b * a.kcap + knew) * a.pstride : nullptr;
    const double wold = (leave && kc < a.kcap) ? (double)mult[(int64_t)b * a.kcap + kc] : 0.0;
    const double hs = (leave && a.hscale) ? a.hscale[b] : 1.0;
    const int64_t stride = (int64_t)gridDim.x * (APT * ABLK);
    while (pair0 < npair) {
        d2u xv[APT], sv[APT];
#pragma unroll
        for (int h = 0; h < APT; ++h) {                              // (second round trip, accepting chains only)
            const int64_t pair = pair0 + (int64_t)h * ABLK, e = 2 * pair;
            const bool in = leave && pair < npair, both = e + 1 < a.p;
            const d2u zero = {0.0, 0.0};
            xv[h] = in ? ld2(x0 + base + e, both) : zero;
            sv[h] = in ? ld2(sumx + base + e, both) : zero;
        }
#pragma unroll
        for (int h = 0; h < APT; ++h) {
            const int64_t pair = pair0 + (int64_t)h * ABLK, e = 2 * pair;
            if (pair >= npair) break;
            const bool both = e + 1 < a.p;
            const d2u v = take ? pv[h] : cv[h];
            if (take) st2(cur + base + e, v.x, v.y, both);
            if (take && a.gcur) st2(a.gcur + base + e, gv[h].x, gv[h].y, both);
            if (better) st2(best + base + e, v.x, v.y, both);
            if (crow) st2(crow + e, v.x, v.y, both);
            if (leave) {
                if (hrow) {
                    const half_t r0 = to_half_sat((v.x - xv[h].x) * hs), r1 = to_half_sat((v.y - xv[h].y) * hs);
                    if (both) { const v2h rr = {r0, r1}; *reinterpret_cast<v2h*>(hrow + e) = rr; } else hrow[e] = r0;
                }
                // + (stay of the state the chain leaves) x (that state - x0)
                st2(sumx + base + e, fma(wold, cv[h].x - xv[h].x, sv[h].x), fma(wold, cv[h].y - xv[h].y, sv[h].y), both);
            }
            if (nx.mode) {                                          // proposal of step + 1 from the new state
                Philox pn;
                pn.gen(a.seed, 2 * (uint64_t)(step + 1), ctr_of(a.chain0 + b, 0, (uint64_t)pair));
                double za, zb;
                normal2(pn, za, zb);
                // same association as k_propose / k_apply_delta: (cur + first term) + second term
                const double o0 = nx.mode == 1 ? fma(nx.c1, z0n, fma(iv[h].x, za, v.x)) : fma(nx.s_iso, za, v.x + iv[h].x);
                const double o1 = nx.mode == 1 ? fma(nx.c1, z0n, fma(iv[h].y, zb, v.y)) : fma(nx.s_iso, zb, v.y + iv[h].y);
                st2(nx.out + base + e, o0, o1, both);
            }
        }
        pair0 += stride;
        if (pair0 < npair) load(pair0);                             // (p > 2 * APT * 64 * ABLK: further passes)
    }
    if (part == 0 && threadIdx.x == 0) {
        if (hist) {
            kcur[sn] = knew;                                        // knew >= kcap: history full, the host checks
            if (take) {
                if (knew < a.kcap) mult[(int64_t)b * a.kcap + knew] = 1;
            } else if (kc < a.kcap) {
                mult[(int64_t)b * a.kcap + kc] += 1;
            }
        }
        cur_lp[sn] = nlp;
        best_lp[sn] = better ? nlp : blp;
        lps[(int64_t)b * (a.nmcmc + 1) + step + 1] = nlp;
        alphas[(int64_t)b * (a.nmcmc + 1) + step + 1] = mh;
        if (take) nacc[b] += 1;
        if (b == 0) step_ptr[1 - a.par] = step + 1;
    }
}


// ---- Hamiltonian Monte Carlo on the device (quinn/mcmc/hmc.py:43-66 around the batched gradient kernel).
// Elementwise over [C, p], HBM-bound: begin 2 reads + 2 writes, leap 3 reads + 2 writes of 8 B per element.  A
// chain's elements are spread over HPARTS(p) workgroups of HBLK threads x HUB elements; each writes ONE partial sum
// of squares (fixed-order tree inside the block), the accept kernel adds the partials left to right: kinetic energies
// are bitwise reproducible and depend only on p, never on how many chains a launch or a rank holds.
constexpr int HBLK = 256;
constexpr int HUB = 4;
__device__ __forceinline__ double block_sum_256(double v, double* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
// momentum z ~ N(0, I) (Philox stream 2*step, keyed by the global chain id), K_cur partials = sum z^2,
// half kick with the cached gradient of the current state, first drift:
//   mom = z + (eps/2) * gs * g_cur;   q = cur + eps * mom           (gs = -0.5 / sigma^2: d logpost = gs * d SSE)
__global__ __launch_bounds__(HBLK) void k_hmc_begin(const double* __restrict__ cur, const double* __restrict__ gcur,
                                                    double half_kick, double eps, int chain0, int64_t p, uint64_t seed,
                                                    const int64_t* __restrict__ step_ptr, double* __restrict__ mom,
                                                    double* __restrict__ q, double* __restrict__ kin_parts) {
    __shared__ double red[4];
    const int b = blockIdx.y;
    const uint64_t step = (uint64_t)*step_ptr;
    const int64_t base = (int64_t)b * p;
    const int64_t npair = (p + 1) / 2;
    const int64_t pchunk = (npair + gridDim.x - 1) / gridDim.x;
    const int64_t lo = blockIdx.x * pchunk, hi = lo + pchunk < npair ? lo + pchunk : npair;
    double ss = 0.0;
    for (int64_t j = lo + threadIdx.x; j < hi; j += HBLK) {
        Philox ph;
        ph.gen(seed, 2 * step, ctr_of(chain0 + b, 0, (uint64_t)j));
        double za, zb;
        normal2(ph, za, zb);
        const int64_t e = 2 * j;
        const double m0 = fma(half_kick, gcur[base + e], za);
        mom[base + e] = m0;
        q[base + e] = fma(eps, m0, cur[base + e]);
        ss = fma(za, za, ss);
        if (e + 1 < p) {
            const double m1 = fma(half_kick, gcur[base + e + 1], zb);
            mom[base + e + 1] = m1;
            q[base + e + 1] = fma(eps, m1, cur[base + e + 1]);
            ss = fma(zb, zb, ss);
        }
    }
    const double tot = block_sum_256(ss, red);
    if (threadIdx.x == 0) kin_parts[(int64_t)b * gridDim.x + blockIdx.x] = tot;
}
// kick with the gradient at q, then (inner step) drift, or (last step) the proposal's kinetic energy partials:
//   mom += kick * g;   last ? K_prop partial = sum mom^2 : q += eps * mom
template <typename TG>
__global__ __launch_bounds__(HBLK) void k_hmc_leap(const TG* __restrict__ g, double kick, double eps, int last, int64_t p,
                                                   double* __restrict__ mom, double* __restrict__ q,
                                                   double* __restrict__ kin_parts) {
    __shared__ double red[4];
    const int b = blockIdx.y;
    const int64_t base = (int64_t)b * p;
    const int64_t chunk = (p + gridDim.x - 1) / gridDim.x;
    const int64_t lo = blockIdx.x * chunk, hi = lo + chunk < p ? lo + chunk : p;
    double ss = 0.0;
    for (int64_t e0 = lo + threadIdx.x; e0 < hi; e0 += (int64_t)HUB * HBLK) {
        double gv[HUB], mv[HUB], qv[HUB];
#pragma unroll
        for (int u = 0; u < HUB; ++u) {
            const int64_t e = e0 + (int64_t)u * HBLK;
            const bool in = e < hi;
            gv[u] = in ? (double)g[base + e] : 0.0;
            mv[u] = in ? mom[base + e] : 0.0;
            qv[u] = (in && !last) ? q[base + e] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < HUB; ++u) {
            const int64_t e = e0 + (int64_t)u * HBLK;
            if (e >= hi) break;
            const double m = fma(kick, gv[u], mv[u]);
            mom[base + e] = m;
            if (last) ss = fma(m, m, ss);
            else q[base + e] = fma(eps, m, qv[u]);
        }
    }
    if (last) {
        const double tot = block_sum_256(ss, red);
        if (threadIdx.x == 0) kin_parts[(int64_t)b * gridDim.x + blockIdx.x] = tot;
    }
}
}  // namespace

extern "C" int qn_mcmc_propose(const double* cur, const double* sd, double c1, int C, int chain0, int64_t p,
                               uint64_t seed, const int64_t* step_ptr, double* out, void* stream) {
    if (!out || !step_ptr || C <= 0 || p <= 0 || C > 65535 || chain0 < 0 || (cur && !sd)) {
        qn_set_error("qn_mcmc_propose: bad argument");
        return QN_EINVAL;
    }
    int gx = (int)(((p + 1) / 2 + BLK - 1) / BLK);
    if (gx > 64) gx = 64;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_propose, dim3(gx, C), dim3(BLK), 0, static_cast<hipStream_t>(stream), cur, sd, c1, chain0, p,
                       seed, step_ptr, out);
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}

extern "C" int qn_mcmc_propose_hist(const double* cur, const void* hist, const float* wsnap, const int32_t* ksnap,
                                    const double* msnap, const double* hscale, double s_lr, double s_iso, int C, int chain0, int64_t p,
                                    int64_t pstride, int kcap, uint64_t seed, const int64_t* step_ptr, double* out,
                                    void* stream) {
    if (!cur || !hist || !wsnap || !ksnap || !msnap || !step_ptr || !out || C <= 0 || C > 65535 || chain0 < 0 || p <= 0 ||
        kcap <= 0 || pstride < p || (pstride & 1)) {
        qn_set_error("qn_mcmc_propose_hist: bad argument (pstride must be even and >= p)");
        return QN_EINVAL;
    }
    HistArgs a;
    a.chain0 = chain0; a.kcap = kcap; a.p = p; a.pstride = pstride; a.s_lr = s_lr; a.s_iso = s_iso; a.seed = seed;
    a.hscale = hscale;
    const int gx = (int)(((p + 1) / 2 + BLK - 1) / BLK);
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_propose_hist, dim3(gx, C), dim3(BLK), 0, static_cast<hipStream_t>(stream), a, cur,
                       static_cast<const half_t*>(hist), wsnap, ksnap, msnap, step_ptr, out);
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}

extern "C" int qn_mcmc_hist_block_steps(void) { return TB; }

static int hist_kstride(int kcap) { return (kcap + 7) / 8 * 8; }
extern "C" size_t qn_mcmc_hist_block_coef_bytes(int C, int kcap) {
    return (C > 0 && kcap > 0) ? (size_t)C * TB * hist_kstride(kcap) * sizeof(half_t) : 0;
}
extern "C" int qn_mcmc_propose_hist_block(const void* hist, const float* wsnap, const int32_t* ksnap,
                                          const double* msnap, const double* hscale, double s_lr, double s_iso, int C, int chain0,
                                          int64_t p, int64_t pstride, int kcap, uint64_t seed, int64_t step0,
                                          const int64_t* step_ptr, void* coef, double* delta, const int32_t* order,
                                          void* stream) {
    if (!hist || !wsnap || !ksnap || !msnap || !coef || !delta || C <= 0 || C > 65535 || chain0 < 0 || p <= 0 || kcap <= 0 ||
        pstride < p || (pstride & 3) || step0 < 0) {
        qn_set_error("qn_mcmc_propose_hist_block: bad argument (pstride must be a multiple of 4 and >= p)");
        return QN_EINVAL;
    }
    HistBlockArgs a;
    a.chain0 = chain0; a.kcap = kcap; a.kstride = hist_kstride(kcap); a.p = p; a.pstride = pstride; a.step0 = step0;
    a.step_ptr = step_ptr;
    a.s_lr = s_lr; a.s_iso = s_iso; a.seed = seed; a.hscale = hscale;
    hipStream_t st = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_hist_coef, dim3((a.kstride / 2 + BLK - 1) / BLK, TB, C), dim3(BLK), 0, st, a, wsnap, ksnap,
                       static_cast<half_t*>(coef));
    hipLaunchKernelGGL(k_hist_block16, dim3((int)((p + 4 * BLK / 2 - 1) / (4 * BLK / 2)), C), dim3(BLK), 0, st, a,
                       static_cast<const half_t*>(hist), static_cast<const half_t*>(coef), ksnap, msnap, order, delta);
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}

extern "C" int qn_mcmc_apply_delta(const double* cur, const double* delta, int t, double s_iso, int C, int chain0,
                                   int64_t p, uint64_t seed, const int64_t* step_ptr, double* out, void* stream) {
    if (!cur || !delta || !out || !step_ptr || t < 0 || t >= TB || C <= 0 || C > 65535 || chain0 < 0 || p <= 0) {
        qn_set_error("qn_mcmc_apply_delta: bad argument");
        return QN_EINVAL;
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_apply_delta, dim3((int)(((p + 1) / 2 + BLK - 1) / BLK), C), dim3(BLK), 0,
                       static_cast<hipStream_t>(stream), cur, delta, t, chain0, p, s_iso, seed, step_ptr, out);
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}

static int accept_parts(int64_t p) {                   // workgroups per chain: APT pairs of elements per thread and pass
    const int64_t n = ((p + 1) / 2 + APT * ABLK - 1) / (APT * ABLK);
    return n > APARTS_MAX ? APARTS_MAX : (n < 1 ? 1 : (int)n);
}
extern "C" int qn_mcmc_accept(const double* prop, const double* sse_prop, double sigma, int n_rows, int C, int chain0,
                              int64_t p, int nmcmc, uint64_t seed, double* cur, double* cur_lp, double* best, double* best_lp,
                              double* chain, double* lps, double* alphas, int64_t* nacc, const double* x0,
                              void* hist, const double* hscale, int32_t* mult, int32_t* kcur, double* sumx, int kcap,
                              int64_t pstride, int64_t* step_ptr, int parity, int nparts, void* stream) {
    if (!prop || !sse_prop || !cur || !cur_lp || !best || !best_lp || !lps || !alphas || !nacc || !step_ptr ||
        C <= 0 || C > 65535 || chain0 < 0 || p <= 0 || sigma <= 0.0 || (parity != 0 && parity != 1) || nparts < 1 ||
        (hist && (!x0 || !mult || !kcur || !sumx || kcap <= 0 || pstride < p))) {
        qn_set_error("qn_mcmc_accept: bad argument");
        return QN_EINVAL;
    }
    AcceptArgs a;
    a.half_inv_sig2 = 0.5 / (sigma * sigma);
    a.lp_const = 0.5 * n_rows * std::log(2.0 * M_PI) + n_rows * std::log(sigma);
    a.chain0 = chain0; a.nmcmc = nmcmc; a.kcap = kcap; a.p = p; a.pstride = pstride; a.seed = seed;
    a.par = parity; a.C = C; a.nparts = nparts; a.hscale = hscale;
    a.nkin = 0; a.kin_cur = a.kin_prop = a.gprop = nullptr; a.gcur = nullptr;
    (void)hipGetLastError();
    NextArgs nx;
    nx.mode = 0; nx.t = 0; nx.c1 = 0.0; nx.s_iso = 0.0; nx.sd = nullptr; nx.delta = nullptr; nx.out = nullptr;
    hipLaunchKernelGGL(k_accept, dim3(accept_parts(p), C), dim3(ABLK), 0, static_cast<hipStream_t>(stream), a, prop, sse_prop, cur, cur_lp,
                       best, best_lp, chain, lps, alphas, nacc, x0, static_cast<half_t*>(hist), mult, kcur, sumx, step_ptr, nx);
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}

extern "C" int qn_mcmc_accept_propose(const double* prop, const double* sse_prop, double sigma, int n_rows, int C,
                                      int chain0, int64_t p, int nmcmc, uint64_t seed, double* cur, double* cur_lp,
                                      double* best, double* best_lp, double* chain, double* lps, double* alphas,
                                      int64_t* nacc, const double* x0, void* hist, const double* hscale, int32_t* mult,
                                      int32_t* kcur, double* sumx, int kcap, int64_t pstride, int64_t* step_ptr, int next_mode,
                                      const double* sd, double c1, const double* delta, int t_next, double s_iso,
                                      double* prop_next, int parity, int nparts, void* stream) {
    if (!prop || !sse_prop || !cur || !cur_lp || !best || !best_lp || !lps || !alphas || !nacc || !step_ptr ||
        C <= 0 || C > 65535 || chain0 < 0 || p <= 0 || sigma <= 0.0 || (parity != 0 && parity != 1) || nparts < 1 || (hist && (!x0 || !mult || !kcur || !sumx || kcap <= 0 || pstride < p)) ||
        next_mode < 0 || next_mode > 2 || (next_mode && !prop_next) || (next_mode == 1 && !sd) ||
        (next_mode == 2 && (!delta || t_next < 0 || t_next >= TB))) {
        qn_set_error("qn_mcmc_accept_propose: bad argument");
        return QN_EINVAL;
    }
    AcceptArgs a;
    a.half_inv_sig2 = 0.5 / (sigma * sigma);
    a.lp_const = 0.5 * n_rows * std::log(2.0 * M_PI) + n_rows * std::log(sigma);
    a.chain0 = chain0; a.nmcmc = nmcmc; a.kcap = kcap; a.p = p; a.pstride = pstride; a.seed = seed;
    a.par = parity; a.C = C; a.nparts = nparts; a.hscale = hscale;
    a.nkin = 0; a.kin_cur = a.kin_prop = a.gprop = nullptr; a.gcur = nullptr;
    NextArgs nx;
    nx.mode = next_mode; nx.t = t_next; nx.c1 = c1; nx.s_iso = s_iso; nx.sd = sd; nx.delta = delta; nx.out = prop_next;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_accept, dim3(accept_parts(p), C), dim3(ABLK), 0, static_cast<hipStream_t>(stream), a, prop, sse_prop, cur, cur_lp,
                       best, best_lp, chain, lps, alphas, nacc, x0, static_cast<half_t*>(hist), mult, kcur, sumx, step_ptr, nx);
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}

// ---------------------------------------------------------------------------------------------- HMC entry points
static int hmc_parts(int64_t p) {                      // workgroups per chain: a function of p alone (see k_hmc_begin)
    const int64_t n = (p + HBLK * HUB - 1) / (HBLK * HUB);
    return n > 64 ? 64 : (n < 1 ? 1 : (int)n);
}
extern "C" int qn_hmc_parts(int64_t p) { return p > 0 ? hmc_parts(p) : QN_EINVAL; }

extern "C" int qn_hmc_begin(const double* cur, const double* grad_cur, double sigma, double epsilon, int C, int chain0,
                            int64_t p, uint64_t seed, const int64_t* step_ptr, double* mom, double* q, double* kin_cur_parts,
                            void* stream) {
    if (!cur || !grad_cur || !step_ptr || !mom || !q || !kin_cur_parts || C <= 0 || C > 65535 || chain0 < 0 || p <= 0 ||
        !(sigma > 0.0)) {
        qn_set_error("qn_hmc_begin: bad argument");
        return QN_EINVAL;
    }
    const double gs = -0.5 / (sigma * sigma);
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_hmc_begin, dim3(hmc_parts(p), C), dim3(HBLK), 0, static_cast<hipStream_t>(stream), cur, grad_cur,
                       0.5 * epsilon * gs, epsilon, chain0, p, seed, step_ptr, mom, q, kin_cur_parts);
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}

extern "C" int qn_hmc_leap(const void* grad_q, int dtype, double sigma, double epsilon, int last, int C, int64_t p, double* mom,
                           double* q, double* kin_prop_parts, void* stream) {
    if (!grad_q || !mom || !q || (last && !kin_prop_parts) || C <= 0 || C > 65535 || p <= 0 || !(sigma > 0.0) ||
        (dtype != QN_F64 && dtype != QN_F32)) {
        qn_set_error("qn_hmc_leap: bad argument");
        return QN_EINVAL;
    }
    const double gs = -0.5 / (sigma * sigma);
    const double kick = (last ? 0.5 : 1.0) * epsilon * gs;
    hipStream_t st = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    if (dtype == QN_F32)
        hipLaunchKernelGGL(k_hmc_leap<float>, dim3(hmc_parts(p), C), dim3(HBLK), 0, st, (const float*)grad_q, kick, epsilon,
                           last ? 1 : 0, p, mom, q, kin_prop_parts);
    else
        hipLaunchKernelGGL(k_hmc_leap<double>, dim3(hmc_parts(p), C), dim3(HBLK), 0, st, (const double*)grad_q, kick, epsilon,
                           last ? 1 : 0, p, mom, q, kin_prop_parts);
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}

extern "C" int qn_hmc_accept(const double* q, const double* grad_q, const double* sse_q, const double* kin_cur_parts,
                             const double* kin_prop_parts, double sigma, int n_rows, int C, int chain0, int64_t p, int nmcmc,
                             uint64_t seed, double* cur, double* grad_cur, double* cur_lp, double* best, double* best_lp,
                             double* chain, double* lps, double* alphas, int64_t* nacc, int64_t* step_ptr, int parity,
                             void* stream) {
    if (!q || !grad_q || !sse_q || !kin_cur_parts || !kin_prop_parts || !cur || !grad_cur || !cur_lp || !best || !best_lp ||
        !lps || !alphas || !nacc || !step_ptr || C <= 0 || C > 65535 || chain0 < 0 || p <= 0 || !(sigma > 0.0) ||
        (parity != 0 && parity != 1)) {
        qn_set_error("qn_hmc_accept: bad argument");
        return QN_EINVAL;
    }
    AcceptArgs a;
    a.half_inv_sig2 = 0.5 / (sigma * sigma);
    a.lp_const = 0.5 * n_rows * std::log(2.0 * M_PI) + n_rows * std::log(sigma);
    a.chain0 = chain0; a.nmcmc = nmcmc; a.kcap = 0; a.p = p; a.pstride = p; a.seed = seed;
    a.par = parity; a.C = C; a.nparts = 1; a.hscale = nullptr;
    a.nkin = hmc_parts(p); a.kin_cur = kin_cur_parts; a.kin_prop = kin_prop_parts; a.gprop = grad_q; a.gcur = grad_cur;
    NextArgs nx;
    nx.mode = 0; nx.t = 0; nx.c1 = 0.0; nx.s_iso = 0.0; nx.sd = nullptr; nx.delta = nullptr; nx.out = nullptr;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_accept, dim3(accept_parts(p), C), dim3(ABLK), 0, static_cast<hipStream_t>(stream), a, q, sse_q, cur,
                       cur_lp, best, best_lp, chain, lps, alphas, nacc, (const double*)nullptr, (half_t*)nullptr,
                       (int32_t*)nullptr, (int32_t*)nullptr, (double*)nullptr, step_ptr, nx);
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}
