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
#include "qn_math.h"

namespace {

constexpr int BLK = 256;

template <typename T> __device__ __forceinline__ T qn_tanh(T x);
template <> __device__ __forceinline__ double qn_tanh<double>(double x) { return qn_tanh_f64(x); }
template <> __device__ __forceinline__ float qn_tanh<float>(float x) { return qn_tanh_f32(x); }

template <typename T> __device__ __forceinline__ T apply_act(T z, int act) {
    if (act == QN_ACT_TANH) return qn_tanh<T>(z);
    if (act == QN_ACT_RELU) return z > T(0) ? z : T(0);
    return z;
}
// derivative expressed through the stored OUTPUT a = act(z)
template <typename T> __device__ __forceinline__ T act_deriv(T a, int act) {
    if (act == QN_ACT_TANH) return T(1) - a * a;
    if (act == QN_ACT_RELU) return a > T(0) ? T(1) : T(0);
    return T(1);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

struct LayerArgs {
    int64_t p, offW, offB;
    int has_bias, h_in, h_out, act, Nb, first, d, o;
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
    for (int k = 0; k < a.h_in; ++k) {
        const T v = a.first ? X[row * a.d + k] : in[((int64_t)b * a.h_in + k) * a.Nb + n];
#pragma unroll
        for (int jj = 0; jj < JB; ++jj)
            if (j0 + jj < a.h_out) acc[jj] = fma(Wl[(int64_t)(j0 + jj) * a.h_in + k], v, acc[jj]);
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
            for (int k = 0; k < a.h_in; ++k) {
                const T v = a.first ? X[row * a.d + k] : in[((int64_t)b * a.h_in + k) * a.Nb + n];
                acc = fma(Wl[(int64_t)j * a.h_in + k], v, acc);
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
            dz_prev[idx] = acc[kk] * act_deriv(a_prev[idx], a.act);
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
    for (int n = threadIdx.x; n < a.Nb; n += BLK) {
        T g[TJ], v[TK];
#pragma unroll
        for (int jj = 0; jj < TJ; ++jj)
            g[jj] = (j0 + jj < a.h_out) ? dz[((int64_t)b * a.h_out + j0 + jj) * a.Nb + n] : T(0);
        if (a.first) {
            int64_t row = n;
            if (row_idx) row = row_idx[(int64_t)b * a.Nb + n];
#pragma unroll
            for (int kk = 0; kk < TK; ++kk) v[kk] = (k0 + kk < a.h_in) ? X[row * a.d + k0 + kk] : T(0);
        } else {
#pragma unroll
            for (int kk = 0; kk < TK; ++kk)
                v[kk] = (k0 + kk < a.h_in) ? a_prev[((int64_t)b * a.h_in + k0 + kk) * a.Nb + n] : T(0);
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
    std::vector<T*> act(L, nullptr);            // act[l] = output of layer l (l < L-1)
    for (int l = 0; l + 1 < L; ++l) act[l] = c.take<T>((size_t)B * d->dims[l + 1] * Nb);
    T* dz_last = grad ? c.take<T>((size_t)B * d->dims[L] * Nb) : nullptr;
    T* dzbuf[2] = {nullptr, nullptr};
    if (grad && L > 1) {
        dzbuf[0] = c.take<T>((size_t)B * d->hmax * Nb);
        dzbuf[1] = c.take<T>((size_t)B * d->hmax * Nb);
    }
    const int nblk = (Nb + BLK - 1) / BLK;
    double* partial = c.take<double>((size_t)B * nblk);
    if (c.off > ws_bytes) {
        qn_set_error("workspace too small: need %zu bytes, got %zu", c.off, ws_bytes);
        return QN_EWORKSPACE;
    }
    constexpr int JB = 8;
    auto largs = [&](int l) {
        LayerArgs a;
        a.p = d->p; a.offW = d->offW[l]; a.offB = d->offB[l]; a.has_bias = d->has_bias;
        a.h_in = d->dims[l]; a.h_out = d->dims[l + 1]; a.act = d->act; a.Nb = Nb; a.first = (l == 0);
        a.d = d->dims[0]; a.o = d->dims[L];
        return a;
    };
    for (int l = 0; l + 1 < L; ++l) {
        LayerArgs a = largs(l);
        dim3 grid(nblk, (a.h_out + JB - 1) / JB, B);
        hipLaunchKernelGGL((k_fwd_hidden<T, JB>), grid, dim3(BLK), 0, st, a, W, l ? act[l - 1] : (const T*)nullptr,
                           X, row_idx, act[l]);
    }
    {
        LayerArgs a = largs(L - 1);
        dim3 grid(nblk, B);
        hipLaunchKernelGGL((k_fwd_last<T>), grid, dim3(BLK), 0, st, a, W, L > 1 ? act[L - 2] : (const T*)nullptr, X, Y,
                           row_idx, dz_last, pred, partial, nblk);
        hipLaunchKernelGGL(k_sse_final, dim3((B + 63) / 64), dim3(64), 0, st, partial, nblk, B, sse);
    }
    if (grad) {
        constexpr int TJ = 8, TK = 8, KB = 8;
        const T* dz = dz_last;
        for (int l = L - 1; l >= 0; --l) {
            LayerArgs a = largs(l);
            dim3 gridw((a.h_in + TK - 1) / TK, (a.h_out + TJ - 1) / TJ, B);
            hipLaunchKernelGGL((k_dW<T, TJ, TK>), gridw, dim3(BLK), 0, st, a, dz, l ? act[l - 1] : (const T*)nullptr,
                               X, row_idx, gradW);
            if (l > 0) {
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

}  // namespace

size_t qn_generic_workspace(const qn_desc* d, int B, int Nb, int want_grad, int dtype) {
    const size_t e = dtype == QN_F64 ? 8 : 4;
    const int L = d->nlayers;
    size_t tot = 0;
    for (int l = 0; l + 1 < L; ++l) tot += qn_align((size_t)B * d->dims[l + 1] * Nb * e);
    if (want_grad) {
        tot += qn_align((size_t)B * d->dims[L] * Nb * e);
        if (L > 1) tot += 2 * qn_align((size_t)B * d->hmax * Nb * e);
    }
    tot += qn_align((size_t)B * ((Nb + BLK - 1) / BLK) * sizeof(double));
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
