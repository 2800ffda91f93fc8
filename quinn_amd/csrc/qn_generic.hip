// Generic (any-shape) batched MLP sum-of-squared-errors forward / backward for gfx950.
//
// One weight vector b, one data row n per lane; activations live feature-major
// ([B][h][Nb], row index fastest) in the caller's workspace so that every global access of
// a wave is 64 consecutive elements.  Weights are wave-uniform operands (scalar loads).
// Per layer: forward kernel (bias + activation fused; the last layer also forms the
// residual, the optional prediction and the per-block SSE partial), dA kernel (activation
// derivative fused) and dW kernel (fixed-order block reduction -> bitwise reproducible).
// This family is the always-correct fallback; qn_fused.hip is the MFMA path for
// LDS-resident weights.
#include "qn_common.h"
#include "qn_fused_args.h"
#include "qn_math.h"
#include <type_traits>
#include <algorithm>

namespace {

constexpr int BLK = 256;

template <typename T> __device__ __forceinline__ T qn_tanh(T x);
template <> __device__ __forceinline__ double qn_tanh<double>(double x) { return qn_tanh_f64(x); }
template <> __device__ __forceinline__ float qn_tanh<float>(float x) { return qn_tanh_f32(x); }
// valid for every non-NaN argument (qn_math.h)
template <typename T> __device__ __forceinline__ T qn_tanh_finite(T x);
template <> __device__ __forceinline__ double qn_tanh_finite<double>(double x) { return qn_tanh_f64_finite(x); }
template <> __device__ __forceinline__ float qn_tanh_finite<float>(float x) { return qn_tanh_f32(x); }

template <typename T> __device__ __forceinline__ T apply_act(T z, int act) {
    if (act == QN_ACT_TANH) return qn_tanh<T>(z);
    if (act == QN_ACT_RELU) return qn_relu<T>(z);
    return z;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

struct LayerArgs {
    int64_t p, offW, offB;
    int has_bias, h_in, h_out, act, Nb, first, d, o;
    int Nsz, Nsa;   // k_dW: row strides of dz [B][h_out][Nsz] and a_prev [B][h_in][Nsa] (= Nb except for the padded stashes of the wide int8 path)
    int eye;        // the layer is the identity map (a residual network without pre / post layer): COPY the input -- a product
                    // with an identity matrix turns an infinite input into NaN (0 . Inf) where the reference passes it on
};

// ---- hidden layer forward: out[b][j][n] = act(b_j + sum_k W[j][k] * in[b][k][n])
template <typename T, int JB>
__global__ __launch_bounds__(BLK) void k_fwd_hidden(LayerArgs a, const T* __restrict__ W,
                                                    const T* __restrict__ in, const T* __restrict__ X,
                                                    const int32_t* __restrict__ row_idx, T* __restrict__ out) {
    const int b = blockIdx.z;
    const int j0 = blockIdx.y * JB;
    const int n = blockIdx.x * BLK + threadIdx.x;
    if (n >= a.Nb) return;
    const T* Wl = W + (int64_t)b * a.p + a.offW;
    T acc[JB];
#pragma unroll
    for (int jj = 0; jj < JB; ++jj)
        acc[jj] = (a.has_bias && j0 + jj < a.h_out) ? W[(int64_t)b * a.p + a.offB + j0 + jj] : T(0);
    int64_t row = n;
    if (a.first && row_idx) row = row_idx[(int64_t)b * a.Nb + n];
    if (a.eye) {
#pragma unroll
        for (int jj = 0; jj < JB; ++jj)
            if (j0 + jj < a.h_out) acc[jj] = a.first ? X[row * a.d + j0 + jj] : in[((int64_t)b * a.h_in + j0 + jj) * a.Nb + n];
    } else {
        for (int k = 0; k < a.h_in; ++k) {
            const T v = a.first ? X[row * a.d + k] : in[((int64_t)b * a.h_in + k) * a.Nb + n];
#pragma unroll
            for (int jj = 0; jj < JB; ++jj)
                if (j0 + jj < a.h_out) acc[jj] = fma(Wl[(int64_t)(j0 + jj) * a.h_in + k], v, acc[jj]);
        }
    }
#pragma unroll
    for (int jj = 0; jj < JB; ++jj)
        if (j0 + jj < a.h_out) out[((int64_t)b * a.h_out + j0 + jj) * a.Nb + n] = apply_act(acc[jj], a.act);
}

// ---- last layer forward: pred, dz_last = 2*(pred - y), per-block SSE partial
template <typename T>
__global__ __launch_bounds__(BLK) void k_fwd_last(LayerArgs a, const T* __restrict__ W, const T* __restrict__ in,
                                                  const T* __restrict__ X, const T* __restrict__ Y,
                                                  const int32_t* __restrict__ row_idx, T* __restrict__ dz_last,
                                                  T* __restrict__ pred, double* __restrict__ partial, int nblk) {
    __shared__ double red[BLK / 64];
    const int b = blockIdx.y;
    const int n = blockIdx.x * BLK + threadIdx.x;
    const bool live = n < a.Nb;
    const T* Wl = W + (int64_t)b * a.p + a.offW;
    double mine = 0.0;
    if (live) {
        int64_t row = n;
        if (row_idx) row = row_idx[(int64_t)b * a.Nb + n];
        for (int j = 0; j < a.h_out; ++j) {
            T acc = a.has_bias ? W[(int64_t)b * a.p + a.offB + j] : T(0);
            if (a.eye) {
                acc = a.first ? X[row * a.d + j] : in[((int64_t)b * a.h_in + j) * a.Nb + n];
            } else {
                for (int k = 0; k < a.h_in; ++k) {
                    const T v = a.first ? X[row * a.d + k] : in[((int64_t)b * a.h_in + k) * a.Nb + n];
                    acc = fma(Wl[(int64_t)j * a.h_in + k], v, acc);
                }
            }
            const T r = acc - Y[row * a.o + j];
            if (pred) pred[((int64_t)b * a.Nb + n) * a.o + j] = acc;
            if (dz_last) dz_last[((int64_t)b * a.h_out + j) * a.Nb + n] = T(2) * r;
            mine += (double)r * (double)r;
        }
    }
    mine = wave_sum(mine);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < BLK / 64; ++w) s += red[w];
        partial[(int64_t)b * nblk + blockIdx.x] = s;
    }
}

__global__ void k_sse_final(const double* __restrict__ partial, int nblk, int B, double* __restrict__ sse) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double s = 0.0;
    for (int i = 0; i < nblk; ++i) s += partial[(int64_t)b * nblk + i];
    sse[b] = s;
}

// ---- backward through a layer's weights and the previous activation:
// dz_prev[b][k][n] = act'(a_prev[b][k][n]) * sum_j W[j][k] * dz[b][j][n]
template <typename T, int KB>
__global__ __launch_bounds__(BLK) void k_bwd_dA(LayerArgs a, const T* __restrict__ W, const T* __restrict__ dz,
                                                const T* __restrict__ a_prev, T* __restrict__ dz_prev) {
    const int b = blockIdx.z;
    const int k0 = blockIdx.y * KB;
    const int n = blockIdx.x * BLK + threadIdx.x;
    if (n >= a.Nb) return;
    const T* Wl = W + (int64_t)b * a.p + a.offW;
    T acc[KB];
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) acc[kk] = T(0);
    for (int j = 0; j < a.h_out; ++j) {
        const T g = dz[((int64_t)b * a.h_out + j) * a.Nb + n];
#pragma unroll
        for (int kk = 0; kk < KB; ++kk)
            if (k0 + kk < a.h_in) acc[kk] = fma(Wl[(int64_t)j * a.h_in + k0 + kk], g, acc[kk]);
    }
#pragma unroll
    for (int kk = 0; kk < KB; ++kk)
        if (k0 + kk < a.h_in) {
            const int64_t idx = ((int64_t)b * a.h_in + k0 + kk) * a.Nb + n;
            dz_prev[idx] = qn_act_bwd<T>(acc[kk], a_prev[idx], a.act);
        }
}

// ---- weight / bias gradient of one layer: dW[j][k] = sum_n dz[j][n] * a_prev[k][n]
template <typename T, int TJ, int TK>
__global__ __launch_bounds__(BLK) void k_dW(LayerArgs a, const T* __restrict__ dz, const T* __restrict__ a_prev,
                                            const T* __restrict__ X, const int32_t* __restrict__ row_idx,
                                            T* __restrict__ gradW) {
    __shared__ double red[BLK / 64][TJ * TK + TJ];
    const int b = blockIdx.z;
    const int j0 = blockIdx.y * TJ;
    const int k0 = blockIdx.x * TK;
    T acc[TJ][TK];
    T accb[TJ];
#pragma unroll
    for (int jj = 0; jj < TJ; ++jj) {
        accb[jj] = T(0);
#pragma unroll
        for (int kk = 0; kk < TK; ++kk) acc[jj][kk] = T(0);
    }
    // (thin shapes stream one operand from HBM: two row steps in flight per thread keep enough loads outstanding; four measured
    // the same at the cfg3 / cfg4 gradient: 4.18 / 17.5 ms either way)
#pragma unroll 2
    for (int n = threadIdx.x; n < a.Nb; n += BLK) {
        T g[TJ], v[TK];
#pragma unroll
        for (int jj = 0; jj < TJ; ++jj)
            g[jj] = (j0 + jj < a.h_out) ? dz[((int64_t)b * a.h_out + j0 + jj) * a.Nsz + n] : T(0);
        if (a.first) {
            int64_t row = n;
            if (row_idx) row = row_idx[(int64_t)b * a.Nb + n];
#pragma unroll
            for (int kk = 0; kk < TK; ++kk) v[kk] = (k0 + kk < a.h_in) ? X[row * a.d + k0 + kk] : T(0);
        } else {
#pragma unroll
            for (int kk = 0; kk < TK; ++kk)
                v[kk] = (k0 + kk < a.h_in) ? a_prev[((int64_t)b * a.h_in + k0 + kk) * a.Nsa + n] : T(0);
        }
#pragma unroll
        for (int jj = 0; jj < TJ; ++jj) {
            accb[jj] += g[jj];
#pragma unroll
            for (int kk = 0; kk < TK; ++kk) acc[jj][kk] = fma(g[jj], v[kk], acc[jj][kk]);
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int jj = 0; jj < TJ; ++jj) {
#pragma unroll
        for (int kk = 0; kk < TK; ++kk) {
            const double s = wave_sum((double)acc[jj][kk]);
            if (lane == 0) red[wave][jj * TK + kk] = s;
        }
        const double sb = wave_sum((double)accb[jj]);
        if (lane == 0) red[wave][TJ * TK + jj] = sb;
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < TJ * TK + TJ) {
        double s = 0.0;
        for (int w = 0; w < BLK / 64; ++w) s += red[w][t];
        if (t < TJ * TK) {
            const int jj = t / TK, kk = t % TK;
            if (j0 + jj < a.h_out && k0 + kk < a.h_in)
                gradW[(int64_t)b * a.p + a.offW + (int64_t)(j0 + jj) * a.h_in + k0 + kk] = (T)s;
        } else if (a.has_bias && blockIdx.x == 0) {
            const int jj = t - TJ * TK;
            if (j0 + jj < a.h_out) gradW[(int64_t)b * a.p + a.offB + j0 + jj] = (T)s;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// float64 MFMA GEMM for hidden -> hidden layers whose widths are multiples of 64 (cfg3..5:
// h = 128 / 256, where one chain's weights no longer fit LDS).  One kernel, three operand maps:
//   FWD  out[j][n]  = act( b_j + sum_i W[j][i] * in[i][n] )                 M=h_out, N'=rows, K=h_in
//   DA   dzp[i][n]  = act'(a[i][n]) * sum_j W[j][i] * dz[j][n]              M=h_in,  N'=rows, K=h_out
//   DW   dW[j][i]  += sum_n dz[j][n] * a[i][n]   (split over K = rows)      M=h_out, N'=h_in, K=rows
// Workgroup = 4 waves = one 64x64 output tile, each wave 2x2 v_mfma_f64_16x16x4_f64 tiles; K in steps
// of 16 through double-buffered LDS tiles Ps[64][16+2], Qs[16][64+16] (strides 18 / 80 doubles make
// the A-fragment (16 rows x 2 k) and B-fragment (2 k x 16 cols) ds_read_b64 conflict-free); the next
// K-step's global loads are issued before the MFMAs of the current one.
template <typename T> struct mfma16;
template <> struct mfma16<double> {
    typedef double v4 __attribute__((ext_vector_type(4)));
    // C/D: reg r of a 16x16 tile = row (lane>>4) + 4 r
    static __device__ __forceinline__ int row(int q, int r) { return q + 4 * r; }
    static constexpr int RSTEP = 4;
    static __device__ __forceinline__ v4 run(double a, double b, v4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
};
template <> struct mfma16<float> {
    typedef float v4 __attribute__((ext_vector_type(4)));
    // v_mfma_f32_16x16x4_f32: reg r = row 4 (lane>>4) + r
    static __device__ __forceinline__ int row(int q, int r) { return 4 * q + r; }
    static constexpr int RSTEP = 1;
    static __device__ __forceinline__ v4 run(float a, float b, v4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
};
enum { GEMM_FWD = 0, GEMM_DA = 1, GEMM_DW = 2 };
constexpr int GKB = 16, GSP = GKB + 2, GSQ = 80;   // K-step 32 halves the residency (76 KB LDS) and measured 18 % slower

struct GemmArgs {
    int64_t p, offW, offB, out_stride_b, out_stride_k;
    int h_in, h_out, Nb, act, has_bias, ksplit, kchunk;
    int inner, outer_total, per_b;      // XCD-aware 1-D grid, see gemm_grid()
};
// Workgroups are dealt round-robin to the 8 XCDs (workgroup L runs on XCD L % 8, each with its own L2).
// Output tiles that read the same operand slab -- FWD / DA: the M-tiles of one column block (same
// activations in[K][64]); DW: the (j, i) tiles of one row chunk (same dz / a rows) -- form an "inner"
// group that is laid out on ONE XCD, back to back in its dispatch order, so the slab comes from HBM once
// and from that XCD's L2 afterwards.  outer = which slab (column block or row chunk, times chain).
inline unsigned gemm_grid(GemmArgs& g, int inner, int per_b, int B) {
    g.inner = inner; g.per_b = per_b; g.outer_total = per_b * B;
    return (unsigned)(((g.outer_total + 7) / 8) * 8 * inner);
}

template <typename T, int MODE>
__global__ __launch_bounds__(BLK) void k_gemm64(GemmArgs g, const T* __restrict__ W, const T* __restrict__ in0,
                                                const T* __restrict__ in1, T* __restrict__ out) {
    typedef typename mfma16<T>::v4 gv4;
    __shared__ __attribute__((aligned(16))) T Ps[2][64 * GSP];
    __shared__ __attribute__((aligned(16))) T Qs[2][GKB * GSQ];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, q = lane >> 4, c = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    const int Nb = g.Nb;
    // XCD-aware decode of the 1-D grid (gemm_grid)
    const int seq = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int inner_i = seq % g.inner, outer = (seq / g.inner) * 8 + xcd;
    if (outer >= g.outer_total) return;
    const int b = outer / g.per_b, slab = outer % g.per_b;
    int m0, n0, kbeg, kend;
    if (MODE == GEMM_DW) {
        const int tiles_i = g.h_in / 64;
        m0 = (inner_i / tiles_i) * 64;             // j0
        n0 = (inner_i % tiles_i) * 64;             // i0
        kbeg = slab * g.kchunk;
        kend = kbeg + g.kchunk < Nb ? kbeg + g.kchunk : Nb;
    } else {
        m0 = inner_i * 64;
        n0 = slab * 64;
        kbeg = 0;
        kend = MODE == GEMM_FWD ? g.h_in : g.h_out;
    }
    const T* Wl = W + (int64_t)b * g.p + g.offW;
    const T* I0 = in0 + (int64_t)b * (MODE == GEMM_FWD ? g.h_in : g.h_out) * Nb;
    const T* I1 = in1 ? in1 + (int64_t)b * g.h_in * Nb : nullptr;

    // Each thread moves 8 + 8 doubles per K-step.  FWD / DA read out-of-range data columns (n >= Nb)
    // from a clamped, valid address instead of masking: those output columns are never stored.
    constexpr int NE = GKB / 4;                 // doubles per thread and operand per K-step
    T pr[NE], qr[NE];
    // DW, i-tile 0 only: the bias gradient db[j] = sum_n dz[j][n] is the row sum of the P operand, which passes
    // through this thread's registers anyway (thread = row tid>>2, a quarter of each K-step)
    const bool want_rowsum = MODE == GEMM_DW && g.has_bias && n0 == 0;
    T rsum = T(0);
    const int ncl = Nb - 1;
    auto gload = [&](int k0) {
        if (MODE == GEMM_FWD) {          // P[m][k] = W[j0+m][k0+k];  Q[k][n] = in[k0+k][n0+n]
            const int m = tid >> 2, kq = (tid & 3) * NE;
            const T* src = Wl + (int64_t)(m0 + m) * g.h_in + k0 + kq;
#pragma unroll
            for (int u = 0; u < NE; ++u) pr[u] = src[u];
            const int k = tid / (64 / NE), nq = (tid % (64 / NE)) * NE;
            const T* qs = I0 + (int64_t)(k0 + k) * Nb;
#pragma unroll
            for (int u = 0; u < NE; ++u) { const int n = n0 + nq + u; qr[u] = qs[n < Nb ? n : ncl]; }
        } else if (MODE == GEMM_DA) {    // P[m][k] = W[k0+k][i0+m];  Q[k][n] = dz[k0+k][n0+n]
            const int m = tid & 63, kq = (tid >> 6) * NE;
#pragma unroll
            for (int u = 0; u < NE; ++u) pr[u] = Wl[(int64_t)(k0 + kq + u) * g.h_in + m0 + m];
            const int k = tid / (64 / NE), nq = (tid % (64 / NE)) * NE;
            const T* qs = I0 + (int64_t)(k0 + k) * Nb;
#pragma unroll
            for (int u = 0; u < NE; ++u) { const int n = n0 + nq + u; qr[u] = qs[n < Nb ? n : ncl]; }
        } else {                         // P[m][k] = dz[j0+m][k0+k];  Q[k][n] = a[i0+n][k0+k]   (k = data row)
            const int m = tid >> 2, kq = (tid & 3) * NE;
            const int nn = tid & 63, kq2 = (tid >> 6) * NE;
            const T* ps = I0 + (int64_t)(m0 + m) * Nb + k0 + kq;
            const T* qs = I1 + (int64_t)(n0 + nn) * Nb + k0 + kq2;
            if (k0 + GKB <= kend) {      // wave-uniform: only the last K-step of the last chunk is ragged
#pragma unroll
                for (int u = 0; u < NE; ++u) pr[u] = ps[u];
#pragma unroll
                for (int u = 0; u < NE; ++u) qr[u] = qs[u];
            } else {
#pragma unroll
                for (int u = 0; u < NE; ++u) pr[u] = k0 + kq + u < kend ? ps[u] : T(0);
#pragma unroll
                for (int u = 0; u < NE; ++u) qr[u] = k0 + kq2 + u < kend ? qs[u] : T(0);
            }
            if (want_rowsum) {
#pragma unroll
                for (int u = 0; u < NE; ++u) rsum += pr[u];
            }
        }
    };
    auto lstore = [&](int buf) {
        if (MODE == GEMM_FWD) {
            const int m = tid >> 2, kq = (tid & 3) * NE;
#pragma unroll
            for (int u = 0; u < NE; ++u) Ps[buf][m * GSP + kq + u] = pr[u];
            const int k = tid / (64 / NE), nq = (tid % (64 / NE)) * NE;
#pragma unroll
            for (int u = 0; u < NE; ++u) Qs[buf][k * GSQ + nq + u] = qr[u];
        } else if (MODE == GEMM_DA) {
            const int m = tid & 63, kq = (tid >> 6) * NE;
#pragma unroll
            for (int u = 0; u < NE; ++u) Ps[buf][m * GSP + kq + u] = pr[u];
            const int k = tid / (64 / NE), nq = (tid % (64 / NE)) * NE;
#pragma unroll
            for (int u = 0; u < NE; ++u) Qs[buf][k * GSQ + nq + u] = qr[u];
        } else {
            const int m = tid >> 2, kq = (tid & 3) * NE;
#pragma unroll
            for (int u = 0; u < NE; ++u) Ps[buf][m * GSP + kq + u] = pr[u];
            const int nn = tid & 63, kq2 = (tid >> 6) * NE;
#pragma unroll
            for (int u = 0; u < NE; ++u) Qs[buf][(kq2 + u) * GSQ + nn] = qr[u];
        }
    };

    gv4 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = (gv4){T(0), T(0), T(0), T(0)};
    if (kbeg < kend) {
        gload(kbeg);
        lstore(0);
    }
    __syncthreads();
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += GKB) {
        const bool more = k0 + GKB < kend;
        if (more) gload(k0 + GKB);                       // global loads in flight under the MFMAs
        const T* pa = &Ps[buf][(16 * 2 * wm + c) * GSP + q];
        const T* pb = &Qs[buf][q * GSQ + 16 * 2 * wn + c];
        // fragments of k-step kk+1 are requested before the MFMAs of k-step kk are issued
        T af[2], bf[2], an[2], bn[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) an[mi] = pa[mi * 16 * GSP];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) bn[ni] = pb[16 * ni];
#pragma unroll
        for (int kk = 0; kk < GKB / 4; ++kk) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) af[mi] = an[mi];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) bf[ni] = bn[ni];
            if (kk + 1 < GKB / 4) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) an[mi] = pa[mi * 16 * GSP + 4 * (kk + 1)];
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) bn[ni] = pb[4 * (kk + 1) * GSQ + 16 * ni];
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = mfma16<T>::run(af[mi], bf[ni], acc[mi][ni]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) lstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // ---- epilogue; C layout: reg r of tile (mi, ni) = row 16*(2wm+mi) + mfma16<T>::row(q, r), col 16*(2wn+ni) + c.
    // Every VALU instruction here competes with the MFMAs for the same pipe, so row pointers are
    // formed once per (mi, r) from one vector base plus wave-uniform offsets, and the activation is chosen per wave.
    constexpr int RS = mfma16<T>::RSTEP;
    const int mrow = m0 + 32 * wm + mfma16<T>::row(q, 0);          // row of (mi = 0, r = 0)
    const int ncol = n0 + 32 * wn + c;                             // column of ni = 0
    if (MODE == GEMM_DW) {
        T* o = out + (int64_t)b * g.out_stride_b + (int64_t)slab * g.out_stride_k + (int64_t)mrow * g.h_in + ncol;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                T* orow = o + (int64_t)(16 * mi + RS * r) * g.h_in;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) orow[16 * ni] = acc[mi][ni][r];
            }
        if (want_rowsum) {          // 4 adjacent lanes hold the quarters of one row; bias block follows the weights
            rsum += __shfl_xor(rsum, 1, 64);
            rsum += __shfl_xor(rsum, 2, 64);
            if ((tid & 3) == 0)
                out[(int64_t)b * g.out_stride_b + (int64_t)slab * g.out_stride_k + (int64_t)g.h_out * g.h_in + m0 +
                    (tid >> 2)] = rsum;
        }
        return;
    }
    const int hrows = MODE == GEMM_FWD ? g.h_out : g.h_in;
    const int64_t base = ((int64_t)b * hrows + mrow) * Nb + ncol;
    const bool ok[2] = {ncol < Nb, ncol + 16 < Nb};
    if (MODE == GEMM_FWD) {
        T v[2][2][4];
        const T* bias = W + (int64_t)b * g.p + g.offB + mrow;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const T bj = g.has_bias ? bias[16 * mi + RS * r] : T(0);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) v[mi][ni][r] = acc[mi][ni][r] + bj;
            }
        if (g.act == QN_ACT_TANH) {
            // a NaN pre-activation makes the sum NaN (so does inf - inf, which only costs the slower variant):
            // one wave-uniform test instead of a NaN mask per element (qn_math.h)
            T sum = T(0);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sum += v[mi][ni][r];
            if constexpr (std::is_same<T, double>::value) {
                // float64: table-assisted tanh (qn_math.h); the operand tiles are free after the K loop, so the
                // 321-entry table goes into Ps (workgroup-uniform branch: every thread reaches the barrier)
                double* tab = reinterpret_cast<double*>(&Ps[0][0]);
                qn_tanh_table_stage(tab, tid, BLK);
                __syncthreads();
                if (__any(sum != sum)) {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[mi][ni][r] = qn_tanh_f64_tab<true>(v[mi][ni][r], tab);
                } else {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[mi][ni][r] = qn_tanh_f64_tab<false>(v[mi][ni][r], tab);
                }
            } else {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[mi][ni][r] = qn_tanh<T>(v[mi][ni][r]);
            }
        } else if (g.act == QN_ACT_RELU) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[mi][ni][r] = qn_relu<T>(v[mi][ni][r]);
        }
        T* o = out + base;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                T* orow = o + (int64_t)(16 * mi + RS * r) * Nb;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    if (ok[ni]) orow[16 * ni] = v[mi][ni][r];
            }
    } else {
        const T* a = in1 + base;
        T* o = out + base;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t ro = (int64_t)(16 * mi + RS * r) * Nb;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    if (ok[ni]) o[ro + 16 * ni] = qn_act_bwd<T>(acc[mi][ni][r], a[ro + 16 * ni], g.act);
            }
    }
}

// gradW[b][off + e] = sum_k slab[b][k][e]
template <typename T>
__global__ __launch_bounds__(BLK) void k_slab_reduce(const T* __restrict__ slab, int ksplit, int64_t n, int64_t p,
                                                     int64_t off, T* __restrict__ gradW) {
    const int b = blockIdx.y;
    for (int64_t e = (int64_t)blockIdx.x * BLK + threadIdx.x; e < n; e += (int64_t)gridDim.x * BLK) {
        double s = 0.0;
        for (int k = 0; k < ksplit; ++k) s += (double)slab[((int64_t)b * ksplit + k) * n + e];
        gradW[(int64_t)b * p + off + e] = (T)s;
    }
}

#if defined(QN_NO_I8_WIDE) || defined(QN_NO_I8_WIDE_BWD) || defined(QN_NO_I8_DW) || defined(QN_DW_UNB_OFF)
#define QN_NO_STASH_PAD               // (A/B builds that send part of the wide gradient through the layer-wise kernels: those take no stride)
#endif
inline int wide_stash_stride(int Nb) { return (Nb + 15) / 16 * 16; }
inline bool gemm_layer(const qn_desc* d, int l) {
    return l >= 1 && l + 1 < d->nlayers && d->dims[l] % 64 == 0 && d->dims[l + 1] % 64 == 0;
}
// split-K factor of the dW GEMM: enough workgroups for ~4 resident per CU (LDS-limited) over several rounds
#ifndef QN_DW_I8_TARGET_WGS
#define QN_DW_I8_TARGET_WGS 1024
#endif
// target: 4096 workgroups for the float64 GEMM (4 resident per CU); the int8-slice kernel holds a CU alone (129 KB of LDS)
// and pays ~8 us of set-up per workgroup (the first group's exponents, three loads in sequence), so it gets long slabs:
// 1024 workgroups = 4 whole rounds on 256 CUs (A/B in one call against 4096 / 2048: cfg3 gradient 47.5 / 48.6 / 49.2 TFLOP/s)
inline int dw_ksplit(int B, int tiles, int Nb, int target = 4096) {
    int ks = (target + B * tiles - 1) / (B * tiles);
    const int kmax = (Nb + 255) / 256;
    if (ks > kmax) ks = kmax;
    if (ks > 16) ks = 16;
    return ks < 1 ? 1 : ks;
}

struct Carve {
    char* base; size_t off, cap;
    template <typename U> U* take(size_t n) {
        U* ptr = reinterpret_cast<U*>(base + off);
        off += qn_align(n * sizeof(U));
        return ptr;
    }
};

template <typename T>
int run_generic(const qn_desc* d, const T* W, const T* X, const T* Y, const int32_t* row_idx, int B, int N,
                int Nb, double* sse, T* pred, T* gradW, void* ws, size_t ws_bytes, hipStream_t st) {
    const int L = d->nlayers;
    const bool grad = gradW != nullptr;
    (void)hipGetLastError();   // drop any stale error of this thread before our launches
    Carve c{static_cast<char*>(ws), 0, ws_bytes};
    // Row stride of the activation / dZ stashes [B][h][Ns].  The gradient of the wide int8 path (qn_wide_i8.hip, qn_dw_i8.hip: every
    // kernel that touches them takes the stride) pads it to a multiple of 16 rows: a feature's rows then start on a 128-byte line
    // whatever the row count -- an unpadded odd count cost those kernels ~12 %, one that is not a multiple of 16 ~5 %
    // (profiles/r04_ragged_rows_dw_ab.txt).  From 64 rows on, where the hidden matrices' weight gradient is the int8 kernel's.
    int Ns = Nb;
    if constexpr (std::is_same<T, double>::value) {
#ifndef QN_NO_STASH_PAD
        if (grad && Nb >= 64 && d->path == QN_PATH_AUTO && qn_i8_wide_applies(d)) Ns = wide_stash_stride(Nb);
#endif
    }
    std::vector<T*> act(L, nullptr);            // act[l] = output of layer l (l < L-1)
    for (int l = 0; l + 1 < L; ++l) act[l] = c.take<T>((size_t)B * d->dims[l + 1] * Ns);
    T* dz_last = grad ? c.take<T>((size_t)B * d->dims[L] * Nb) : nullptr;
    T* dzbuf[2] = {nullptr, nullptr};
    if (grad && L > 1) {
        dzbuf[0] = c.take<T>((size_t)B * d->hmax * Ns);
        dzbuf[1] = c.take<T>((size_t)B * d->hmax * Ns);
        // the fused int8-slice backward writes dZ of EVERY hidden layer before the weight-gradient kernels run: L - 1
        // consecutive buffers, the two above being the first
        if constexpr (std::is_same<T, double>::value)
            if (qn_i8_wide_applies(d))
                for (int l = 2; l < L - 1; ++l) c.take<T>((size_t)B * d->hmax * Ns);
    }
    const int nblk = (Nb + BLK - 1) / BLK;
    double* partial = c.take<double>((size_t)B * nblk);
    // split-K slabs of the MFMA dW GEMM (float64 hidden->hidden layers with widths % 64 == 0)
    T* dwslab = nullptr;
    if (grad) {
        size_t need = 0;
        for (int l = 1; l + 1 < L; ++l)
            if (gemm_layer(d, l)) {
                const int tiles = (d->dims[l] / 64) * (d->dims[l + 1] / 64);
                const int ks = dw_ksplit(B, tiles, Nb);          // (the larger of the two targets: an upper bound for the int8 path)
                if (ks > 1) need = std::max(need, (size_t)B * ks * ((size_t)d->dims[l] * d->dims[l + 1] + d->dims[l + 1]));
            }
        if (need) dwslab = c.take<T>(need);
    }
    // float64 tanh networks whose hidden widths are multiples of 64: the first layer and the hidden->hidden layers of
    // the FORWARD pass run as sliced int8 products (qn_fused_i8.hip: qn_i8_layers_forward), unless a kernel family is
    // forced on the descriptor (QN_PATH_GENERIC stays the exact float64 reference of the tests)
    bool i8_fwd = false, wide = false;
    void* i8_ws = nullptr;
    void* wide_ws = nullptr;
    if constexpr (std::is_same<T, double>::value) {
#ifndef QN_NO_I8_LAYERS
        i8_fwd = d->path == QN_PATH_AUTO && qn_i8_layers_apply(d);
#endif
        const size_t nb8 = qn_i8_wide_applies(d) ? 0 : qn_i8_layers_workspace(d, B, Nb);      // (the fused kernels take precedence)
        if (nb8) i8_ws = c.take<char>(nb8);
#ifndef QN_NO_I8_WIDE
        // uniform 128 / 256-wide networks with one output: the whole forward pass is ONE launch (qn_wide_i8.hip)
        wide = d->path == QN_PATH_AUTO && qn_i8_wide_applies(d);
#endif
        const size_t nbw = qn_i8_wide_workspace(d, B, Nb, grad);
        if (nbw) wide_ws = c.take<char>(nbw);
    }
    if (c.off > ws_bytes) {
        qn_set_error("workspace too small: need %zu bytes, got %zu", c.off, ws_bytes);
        return QN_EWORKSPACE;
    }
    constexpr int JB = 8;
    auto largs = [&](int l) {
        LayerArgs a;
        a.p = d->p; a.offW = d->offW[l]; a.offB = d->offB[l]; a.has_bias = d->has_bias;
        a.h_in = d->dims[l]; a.h_out = d->dims[l + 1]; a.act = d->act; a.Nb = Nb; a.first = (l == 0);
        a.d = d->dims[0]; a.o = d->dims[L]; a.eye = 0; a.Nsz = Nb; a.Nsa = Nb;
        return a;
    };
    auto gargs = [&](int l) {
        GemmArgs g;
        g.p = d->p; g.offW = d->offW[l]; g.offB = d->offB[l]; g.out_stride_b = 0; g.out_stride_k = 0;
        g.h_in = d->dims[l]; g.h_out = d->dims[l + 1]; g.Nb = Nb; g.act = d->act; g.has_bias = d->has_bias;
        g.ksplit = 1; g.kchunk = Nb;
        return g;
    };
    if constexpr (std::is_same<T, double>::value) {
        if (wide) {
            // activations are written only for the backward pass; act[l] are consecutive, equally sized blocks
            if (int rc = qn_i8_wide_forward(d, W, X, Y, row_idx, B, Nb, grad ? act[0] : nullptr, act[1] - act[0], Ns, dz_last,
                                            pred, sse, wide_ws, st))
                return rc;
        } else if (i8_fwd) {
            if (int rc = qn_i8_layers_forward(d, W, X, row_idx, B, Nb, act.data(), i8_ws, st)) return rc;
        }
    }
    for (int l = 0; l + 1 < L && !i8_fwd && !wide; ++l) {
        if (gemm_layer(d, l)) {
            GemmArgs g = gargs(l);
            const unsigned grid = gemm_grid(g, g.h_out / 64, (Nb + 63) / 64, B);
            hipLaunchKernelGGL((k_gemm64<T, GEMM_FWD>), dim3(grid), dim3(BLK), 0, st, g, W, (const T*)act[l - 1],
                               (const T*)nullptr, act[l]);
            continue;
        }
        LayerArgs a = largs(l);
        dim3 grid(nblk, (a.h_out + JB - 1) / JB, B);
        hipLaunchKernelGGL((k_fwd_hidden<T, JB>), grid, dim3(BLK), 0, st, a, W, l ? act[l - 1] : (const T*)nullptr,
                           X, row_idx, act[l]);
    }
    if (!wide) {
        LayerArgs a = largs(L - 1);
        dim3 grid(nblk, B);
        hipLaunchKernelGGL((k_fwd_last<T>), grid, dim3(BLK), 0, st, a, W, L > 1 ? act[L - 2] : (const T*)nullptr, X, Y,
                           row_idx, dz_last, pred, partial, nblk);
        hipLaunchKernelGGL(k_sse_final, dim3((B + 63) / 64), dim3(64), 0, st, partial, nblk, B, sse);
    }
    if (grad) {
        constexpr int TJ = 8, TK = 8, KB = 8;
        const T* dz = dz_last;
        // fused int8-slice backward through the hidden layers (qn_wide_i8.hip): dZ_l of every hidden layer in one launch
        bool wide_bwd = false;
        int last_done = 0;                                   // the output layer's weight gradient came out of the fused backward (h = 128)
        const int64_t dz_stride = (int64_t)(qn_align((size_t)B * d->hmax * Ns * sizeof(T)) / sizeof(T));
        if constexpr (std::is_same<T, double>::value) {
#ifndef QN_NO_I8_WIDE_BWD
            wide_bwd = wide;
#endif
            if (wide_bwd) {
                if (int rc = qn_i8_wide_backward(d, W, X, row_idx, B, Nb, act[0], act[1] - act[0], Ns, dz_last, dzbuf[0], dz_stride,
                                                 wide_ws, gradW, &last_done, st))
                    return rc;
            }
        }
        // the hidden matrices' weight gradient as sliced int8 products (qn_dw_i8.hip): tanh networks; relu / identity ones through
        // the forward's row scales, from 64 rows on (else the float64-MFMA product below)
        const double* rowsc = nullptr;
        bool i8_dw_ok = false;
        if constexpr (std::is_same<T, double>::value) {
            if (wide_bwd) {
                rowsc = qn_i8_wide_rowscale(d, B, Nb, 1, wide_ws);
                i8_dw_ok = d->act == QN_ACT_TANH || (rowsc != nullptr && Nb >= 64);
#ifdef QN_DW_UNB_OFF
                i8_dw_ok = d->act == QN_ACT_TANH;                        // (A/B)
#endif
            }
        }
        for (int l = L - 1; l >= 0; --l) {
            LayerArgs a = largs(l);
            if (wide_bwd && l < L - 1) dz = dzbuf[0] + (int64_t)l * dz_stride;
            if (wide_bwd) {                                      // (k_dW of the thin layers: the stashes' row stride; dz_last is [B][o][Nb])
                a.Nsa = Ns;
                if (l < L - 1) a.Nsz = Ns;
            }
            if (l == L - 1 && last_done) continue;
            if (gemm_layer(d, l)) {
                GemmArgs g = gargs(l);
                const int tiles = (g.h_in / 64) * (g.h_out / 64);
                const int ks = dw_ksplit(B, tiles, Nb, wide_bwd && i8_dw_ok ? QN_DW_I8_TARGET_WGS : 4096);
                // weights and (if any) the bias block behind them: contiguous in the flat layout and in a slab
                const int64_t nW = (int64_t)g.h_in * g.h_out + (d->has_bias ? g.h_out : 0);
                g.ksplit = ks;
                g.kchunk = ((Nb + ks - 1) / ks + GKB - 1) / GKB * GKB;
                T* dst = gradW + d->offW[l];
                g.out_stride_b = d->p; g.out_stride_k = 0;
                if (ks > 1) { dst = dwslab; g.out_stride_b = (int64_t)ks * nW; g.out_stride_k = nW; }
                bool dw_done = false;
                if constexpr (std::is_same<T, double>::value) {
#ifndef QN_NO_I8_DW
                    // (sliced int8 products, qn_dw_i8.hip; K-slabs in whole 64-row chunks)
                    if (wide_bwd && i8_dw_ok) {
                        const int kc64 = ((Nb + ks - 1) / ks + 63) / 64 * 64;
                        const int rc = qn_i8_dw(g.h_in, g.h_out, d->has_bias, dz, act[l - 1], B, Nb, Ns, dst, g.out_stride_b,
                                                g.out_stride_k, ks, kc64, rowsc ? rowsc + (int64_t)(l - 1) * B * Nb : nullptr, st);
                        if (rc == QN_OK) dw_done = true;
                        else if (!(rc == QN_EUNSUPPORTED && rowsc && Ns == Nb)) return rc;   // (row scales in a build without the group-scale kernel: the float64 product below, which knows no padded stride)
                    }
#endif
                }
                if (!dw_done)
                    hipLaunchKernelGGL((k_gemm64<T, GEMM_DW>), dim3(gemm_grid(g, tiles, ks, B)), dim3(BLK), 0, st, g, W, dz,
                                       (const T*)act[l - 1], dst);
                if (ks > 1) {
                    int gx = (int)((nW + BLK - 1) / BLK);
                    if (gx > 64) gx = 64;
                    hipLaunchKernelGGL(k_slab_reduce<T>, dim3(gx, B), dim3(BLK), 0, st, (const T*)dwslab, ks, nW, d->p,
                                       d->offW[l], gradW);
                }
                if constexpr (std::is_same<T, double>::value) {
                    if (dw_done)            // (the int8 kernel took the whole 64-row chunks: the last Nb % 64 rows in float64)
                        if (int rc = qn_i8_dw_tail(g.h_in, g.h_out, d->has_bias, dz, act[l - 1], B, Nb, Ns, gradW + d->offW[l], d->p, st))
                            return rc;
                }
                if (wide_bwd) continue;
                T* dzp = dzbuf[l & 1];
                g.ksplit = 1; g.kchunk = Nb;
                hipLaunchKernelGGL((k_gemm64<T, GEMM_DA>), dim3(gemm_grid(g, g.h_in / 64, (Nb + 63) / 64, B)), dim3(BLK), 0,
                                   st, g, W, dz, (const T*)act[l - 1], dzp);
                dz = dzp;
                continue;
            }
            // tile shape by layer shape: a thin input (d <= 2 / 4) or a single output wastes most of an 8 x 8 tile's loads
            // and accumulators (same sums in the same order whatever the tile: every (j, k) is summed independently)
            auto launch_dw = [&](auto tj_tag, auto tk_tag) {
                constexpr int tj = decltype(tj_tag)::value, tk = decltype(tk_tag)::value;
                dim3 gridw((a.h_in + tk - 1) / tk, (a.h_out + tj - 1) / tj, B);
                hipLaunchKernelGGL((k_dW<T, tj, tk>), gridw, dim3(BLK), 0, st, a, dz, l ? act[l - 1] : (const T*)nullptr,
                                   X, row_idx, gradW);
            };
            using std::integral_constant;
            if (a.h_out == 1 && a.h_in >= 16) launch_dw(integral_constant<int, 1>{}, integral_constant<int, 16>{});
            else if (a.h_in <= 2 && a.h_out >= 8) launch_dw(integral_constant<int, 8>{}, integral_constant<int, 2>{});
            else if (a.h_in <= 4 && a.h_out >= 8) launch_dw(integral_constant<int, 8>{}, integral_constant<int, 4>{});
            else launch_dw(integral_constant<int, TJ>{}, integral_constant<int, TK>{});
            if (l > 0 && !wide_bwd) {
                T* dzp = dzbuf[l & 1];
                dim3 grida(nblk, (a.h_in + KB - 1) / KB, B);
                hipLaunchKernelGGL((k_bwd_dA<T, KB>), grida, dim3(BLK), 0, st, a, W, dz, act[l - 1], dzp);
                dz = dzp;
            }
        }
    }
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}

// ---------------------------------------------------------------------------------------------
// Residual network (quinn/nns/rnet.py:130-165), layer-wise.  Every weight parameterisation of the
// reference (Const / Lin / Quad / Cubic / Poly / NonPar, rnet.py:217-380) is linear in its
// parameters, W_i = sum_k coef[i][k] * ww_k, so the per-step weights are expanded once per call
// into Weff[b][i][r*r + r]; the layer kernels above then run on Weff (stride / offsets through
// LayerArgs), and the step gradients are contracted back with the same coefficients.
//   OUT_0 = act(Wpre x + bpre) | x ;  TH_i = act(Weff_i OUT_i + beff_i) ;
//   OUT_{i+1} = mlp ? TH_i : OUT_i + h * TH_i ;  pred = Wpost OUT_S + bpost | OUT_S.
struct RnCoef { double c[QN_MAX_LAYERS * QN_MAX_LAYERS]; unsigned char use[QN_MAX_LAYERS * QN_MAX_LAYERS]; };

template <typename T>
__global__ __launch_bounds__(BLK) void k_rn_expand(RnCoef cf, const T* __restrict__ W, int64_t p, int64_t offWW,
                                                   int64_t offBB, int r, int steps, int npar, int has_bias,
                                                   T* __restrict__ Weff) {
    const int b = blockIdx.y;
    const int rr = r * r, per = rr + r, tot = steps * per;
    const T* Wb = W + (int64_t)b * p;
    for (int e = blockIdx.x * BLK + threadIdx.x; e < tot; e += gridDim.x * BLK) {
        const int i = e / per, q = e % per;
        T s = T(0);
        if (q < rr) {
            for (int k = 0; k < npar; ++k)
                if (cf.use[i * npar + k]) s = fma((T)cf.c[i * npar + k], Wb[offWW + (int64_t)k * rr + q], s);
        } else if (has_bias) {
            for (int k = 0; k < npar; ++k)
                if (cf.use[i * npar + k]) s = fma((T)cf.c[i * npar + k], Wb[offBB + (int64_t)k * r + q - rr], s);
        }
        Weff[(int64_t)b * tot + e] = s;
    }
}

template <typename T>
__global__ __launch_bounds__(BLK) void k_rn_contract(RnCoef cf, const T* __restrict__ dWeff, int r, int steps,
                                                     int npar, int has_bias, int64_t p, int64_t offWW,
                                                     int64_t offBB, T* __restrict__ gradW) {
    const int b = blockIdx.y;
    const int rr = r * r, per = rr + r, tot = npar * per;
    for (int e = blockIdx.x * BLK + threadIdx.x; e < tot; e += gridDim.x * BLK) {
        const int k = e / per, q = e % per;
        if (q >= rr && !has_bias) continue;
        double s = 0.0;
        for (int i = 0; i < steps; ++i)
            if (cf.use[i * npar + k]) s = fma(cf.c[i * npar + k], (double)dWeff[((int64_t)b * steps + i) * per + q], s);
        if (q < rr) gradW[(int64_t)b * p + offWW + (int64_t)k * rr + q] = (T)s;
        else gradW[(int64_t)b * p + offBB + (int64_t)k * r + q - rr] = (T)s;
    }
}

template <typename T>
__global__ __launch_bounds__(BLK) void k_rn_eye(int r, T* __restrict__ I) {
    const int e = blockIdx.x * BLK + threadIdx.x;
    if (e < r * r) I[e] = (e / r == e % r) ? T(1) : T(0);
}
// out = a + h * th
template <typename T>
__global__ __launch_bounds__(BLK) void k_rn_axpy(const T* __restrict__ a, const T* __restrict__ th, T h, int64_t n,
                                                 T* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (e < n) out[e] = a[e] + h * th[e];
}
// dz = s * g * act'(th)
template <typename T>
__global__ __launch_bounds__(BLK) void k_rn_dz(const T* __restrict__ g, const T* __restrict__ th, T s, int act,
                                               int64_t n, T* __restrict__ dz) {
    const int64_t e = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (e < n) dz[e] = qn_act_bwd<T>(s * g[e], th[e], act);
}
// t += g
template <typename T>
__global__ __launch_bounds__(BLK) void k_rn_add(const T* __restrict__ g, int64_t n, T* __restrict__ t) {
    const int64_t e = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (e < n) t[e] += g[e];
}

template <typename T>
int run_rnet(const qn_desc* d, const T* W, const T* X, const T* Y, const int32_t* row_idx, int B, int N, int Nb,
             double* sse, T* pred, T* gradW, void* ws, size_t ws_bytes, hipStream_t st) {
    (void)N;
    const bool grad = gradW != nullptr;
    const int r = d->rn_r, S = d->rn_steps, din = d->dims[0], o = d->dims[2];
    const int per = r * r + r;
    const int64_t nact = (int64_t)B * r * Nb;
    (void)hipGetLastError();
    Carve c{static_cast<char*>(ws), 0, ws_bytes};
    std::vector<T*> OUT(S + 1), TH(S);
    OUT[0] = c.take<T>(nact);
    for (int i = 0; i < S; ++i) {
        TH[i] = c.take<T>(nact);
        OUT[i + 1] = d->rn_mlp ? TH[i] : c.take<T>(nact);
    }
    T* Weff = c.take<T>((size_t)B * S * per);
    T* eye = c.take<T>((size_t)r * r);
    const int nblk = (Nb + BLK - 1) / BLK;
    double* partial = c.take<double>((size_t)B * nblk);
    T *dz_last = nullptr, *dWeff = nullptr, *buf[3] = {nullptr, nullptr, nullptr};
    if (grad) {
        dz_last = c.take<T>((size_t)B * o * Nb);
        for (int i = 0; i < 3; ++i) buf[i] = c.take<T>(nact);
        dWeff = c.take<T>((size_t)B * S * per);
    }
    if (c.off > ws_bytes) {
        qn_set_error("workspace too small: need %zu bytes, got %zu", c.off, ws_bytes);
        return QN_EWORKSPACE;
    }
    RnCoef cf;
    for (int i = 0; i < S * d->rn_npar; ++i) cf.c[i] = d->rn_coef[i];
    for (int i = 0; i < S * d->rn_npar; ++i) cf.use[i] = d->rn_uses[i];
    constexpr int JB = 8, TJ = 8, TK = 8, KB = 8;
    const int egrid = (int)((nact + BLK - 1) / BLK);
    auto base = [&]() {
        LayerArgs a;
        a.p = d->p; a.offW = 0; a.offB = 0; a.has_bias = d->has_bias; a.h_in = r; a.h_out = r; a.act = d->act;
        a.Nb = Nb; a.first = 0; a.d = din; a.o = o; a.eye = 0; a.Nsz = Nb; a.Nsa = Nb;
        return a;
    };
    LayerArgs apre = base();     // x -> OUT_0
    apre.first = 1; apre.h_in = din;
    const T* Wpre = W;
    if (d->rn_pre) { apre.offW = d->rn_offWpre; apre.offB = d->rn_offBpre; apre.has_bias = 1; }
    else { Wpre = eye; apre.p = 0; apre.has_bias = 0; apre.act = QN_ACT_IDENTITY; apre.eye = 1; }
    LayerArgs apost = base();    // OUT_S -> pred
    apost.h_out = o;
    const T* Wpost = W;
    if (d->rn_post) { apost.offW = d->rn_offWpost; apost.offB = d->rn_offBpost; apost.has_bias = 1; }
    else { Wpost = eye; apost.p = 0; apost.has_bias = 0; apost.eye = 1; }
    auto astep = [&](int i) {    // OUT_i -> TH_i on the expanded weights
        LayerArgs a = base();
        a.p = (int64_t)S * per; a.offW = (int64_t)i * per; a.offB = a.offW + r * r;
        return a;
    };
    if (!d->rn_pre || !d->rn_post)
        hipLaunchKernelGGL(k_rn_eye<T>, dim3((r * r + BLK - 1) / BLK), dim3(BLK), 0, st, r, eye);
    hipLaunchKernelGGL(k_rn_expand<T>, dim3((S * per + BLK - 1) / BLK, B), dim3(BLK), 0, st, cf, W, d->p,
                       d->rn_offWW, d->rn_offBB, r, S, d->rn_npar, d->has_bias, Weff);
    const dim3 gridh(nblk, (r + JB - 1) / JB, B);
    hipLaunchKernelGGL((k_fwd_hidden<T, JB>), gridh, dim3(BLK), 0, st, apre, Wpre, (const T*)nullptr, X, row_idx,
                       OUT[0]);
    for (int i = 0; i < S; ++i) {
        hipLaunchKernelGGL((k_fwd_hidden<T, JB>), gridh, dim3(BLK), 0, st, astep(i), (const T*)Weff,
                           (const T*)OUT[i], X, row_idx, TH[i]);
        if (!d->rn_mlp)
            hipLaunchKernelGGL(k_rn_axpy<T>, dim3(egrid), dim3(BLK), 0, st, (const T*)OUT[i], (const T*)TH[i],
                               (T)(1.0 / S), nact, OUT[i + 1]);
    }
    hipLaunchKernelGGL((k_fwd_last<T>), dim3(nblk, B), dim3(BLK), 0, st, apost, Wpost, (const T*)OUT[S], X, Y,
                       row_idx, dz_last, pred, partial, nblk);
    hipLaunchKernelGGL(k_sse_final, dim3((B + 63) / 64), dim3(64), 0, st, partial, nblk, B, sse);
    if (grad) {
        const dim3 grida(nblk, (r + KB - 1) / KB, B);
        const dim3 gridw((r + TK - 1) / TK, (r + TJ - 1) / TJ, B);
        // G = d SSE / d OUT_S
        const T* G = dz_last;
        int nb = 0;              // next free scratch buffer
        if (d->rn_post) {
            hipLaunchKernelGGL((k_dW<T, TJ, TK>), dim3((r + TK - 1) / TK, (o + TJ - 1) / TJ, B), dim3(BLK), 0, st,
                               apost, (const T*)dz_last, (const T*)OUT[S], X, row_idx, gradW);
            LayerArgs a = apost; a.act = QN_ACT_IDENTITY;
            hipLaunchKernelGGL((k_bwd_dA<T, KB>), grida, dim3(BLK), 0, st, a, Wpost, (const T*)dz_last,
                               (const T*)OUT[S], buf[0]);
            G = buf[0]; nb = 1;
        }
        const T sc = d->rn_mlp ? T(1) : (T)(1.0 / S);
        for (int i = S - 1; i >= 0; --i) {
            T* dz = buf[nb]; nb = (nb + 1) % 3;
            hipLaunchKernelGGL(k_rn_dz<T>, dim3(egrid), dim3(BLK), 0, st, G, (const T*)TH[i], sc, d->act, nact, dz);
            LayerArgs a = astep(i);
            hipLaunchKernelGGL((k_dW<T, TJ, TK>), gridw, dim3(BLK), 0, st, a, (const T*)dz, (const T*)OUT[i], X,
                               row_idx, dWeff);
            if (i > 0 || d->rn_pre) {
                T* t = buf[nb]; nb = (nb + 1) % 3;
                a.act = QN_ACT_IDENTITY;
                hipLaunchKernelGGL((k_bwd_dA<T, KB>), grida, dim3(BLK), 0, st, a, (const T*)Weff, (const T*)dz,
                                   (const T*)OUT[i], t);
                if (!d->rn_mlp) hipLaunchKernelGGL(k_rn_add<T>, dim3(egrid), dim3(BLK), 0, st, G, nact, t);
                G = t;
            }
        }
        if (d->rn_pre) {
            T* dz = buf[nb];
            hipLaunchKernelGGL(k_rn_dz<T>, dim3(egrid), dim3(BLK), 0, st, G, (const T*)OUT[0], T(1), d->act, nact, dz);
            hipLaunchKernelGGL((k_dW<T, TJ, TK>), dim3((din + TK - 1) / TK, (r + TJ - 1) / TJ, B), dim3(BLK), 0, st,
                               apre, (const T*)dz, (const T*)nullptr, X, row_idx, gradW);
        }
        const int tot = d->rn_npar * per;
        hipLaunchKernelGGL(k_rn_contract<T>, dim3((tot + BLK - 1) / BLK, B), dim3(BLK), 0, st, cf, (const T*)dWeff, r,
                           S, d->rn_npar, d->has_bias, d->p, d->rn_offWW, d->rn_offBB, gradW);
    }
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}

}  // namespace

size_t qn_generic_workspace(const qn_desc* d, int B, int Nb, int want_grad, int dtype) {
    const size_t e = dtype == QN_F64 ? 8 : 4;
    const int L = d->nlayers;
    size_t tot = 0;
    // (the wide int8 gradient path pads the stashes' row stride: sized for it whatever path the descriptor is forced to)
    const int Ns = (dtype == QN_F64 && want_grad && qn_i8_wide_applies(d)) ? wide_stash_stride(Nb) : Nb;
    for (int l = 0; l + 1 < L; ++l) tot += qn_align((size_t)B * d->dims[l + 1] * Ns * e);
    if (want_grad) {
        tot += qn_align((size_t)B * d->dims[L] * Nb * e);
        if (L > 1) tot += 2 * qn_align((size_t)B * d->hmax * Ns * e);
        if (dtype == QN_F64 && qn_i8_wide_applies(d))
            for (int l = 2; l < L - 1; ++l) tot += qn_align((size_t)B * d->hmax * Ns * e);
    }
    tot += qn_align((size_t)B * ((Nb + BLK - 1) / BLK) * sizeof(double));
    if (want_grad) {
        size_t need = 0;
        for (int l = 1; l + 1 < L; ++l)
            if (gemm_layer(d, l)) {
                const int tiles = (d->dims[l] / 64) * (d->dims[l + 1] / 64);
                const int ks = dw_ksplit(B, tiles, Nb);
                if (ks > 1) need = std::max(need, (size_t)B * ks * ((size_t)d->dims[l] * d->dims[l + 1] + d->dims[l + 1]));
            }
        tot += qn_align(need * e);
    }
    if (dtype == QN_F64 && !qn_i8_wide_applies(d)) tot += qn_align(qn_i8_layers_workspace(d, B, Nb));      // layer-wise int8-slice forward (0 if it does not apply)
    if (dtype == QN_F64) tot += qn_align(qn_i8_wide_workspace(d, B, Nb, want_grad));        // fused int8-slice forward (0 if it does not apply)
    return tot + 256;
}

int qn_generic_run(const qn_desc* d, int dtype, const void* W, const void* X, const void* Y,
                   const int32_t* row_idx, int B, int N, int Nb, double* sse, void* pred, void* gradW,
                   void* ws, size_t ws_bytes, hipStream_t st) {
    if (dtype == QN_F64)
        return run_generic<double>(d, (const double*)W, (const double*)X, (const double*)Y, row_idx, B, N, Nb, sse,
                                   (double*)pred, (double*)gradW, ws, ws_bytes, st);
    return run_generic<float>(d, (const float*)W, (const float*)X, (const float*)Y, row_idx, B, N, Nb, sse,
                              (float*)pred, (float*)gradW, ws, ws_bytes, st);
}

size_t qn_rnet_workspace(const qn_desc* d, int B, int Nb, int want_grad, int dtype) {
    const size_t e = dtype == QN_F64 ? 8 : 4;
    const int r = d->rn_r, S = d->rn_steps, o = d->dims[2];
    const size_t nact = qn_align((size_t)B * r * Nb * e);
    const size_t weff = qn_align((size_t)B * S * (r * r + r) * e);
    size_t tot = nact * (1 + (d->rn_mlp ? S : 2 * S)) + weff + qn_align((size_t)r * r * e);
    tot += qn_align((size_t)B * ((Nb + BLK - 1) / BLK) * sizeof(double));
    if (want_grad) tot += qn_align((size_t)B * o * Nb * e) + 3 * nact + weff;
    return tot + 256;
}

int qn_rnet_run(const qn_desc* d, int dtype, const void* W, const void* X, const void* Y, const int32_t* row_idx,
                int B, int N, int Nb, double* sse, void* pred, void* gradW, void* ws, size_t ws_bytes,
                hipStream_t st) {
    if (dtype == QN_F64)
        return run_rnet<double>(d, (const double*)W, (const double*)X, (const double*)Y, row_idx, B, N, Nb, sse,
                                (double*)pred, (double*)gradW, ws, ws_bytes, st);
    return run_rnet<float>(d, (const float*)W, (const float*)X, (const float*)Y, row_idx, B, N, Nb, sse,
                           (float*)pred, (float*)gradW, ws, ws_bytes, st);
}
