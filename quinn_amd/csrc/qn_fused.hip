// Fused batched-MLP kernels for gfx950 with the chain's weights resident in LDS.
//
// Work decomposition.  grid = (nsplit, B): workgroup (s, b) stages weight vector b into LDS once
// and walks its share of the data rows; B x nsplit is sized to put >= 2 workgroups on every one
// of the 256 CUs.  A workgroup is 4 waves; a wave processes G groups of 16 data rows at a time.
//
// Layout of a layer in registers (float64 path, v_mfma_f64_16x16x4_f64).  Every hidden layer is
// computed TRANSPOSED, Z^T[h_out x rows] = W[h_out x h_in] . A^T[h_in x rows]:
//   A operand  = a 16x4 tile of W, read from LDS      (lane: row = lane&15, k = lane>>4)
//   B operand  = a 4x16 tile of the activations A^T   (lane: k = lane>>4, col = lane&15 = data row)
//   C/D        = 16 features x 16 data rows           (lane: col = lane&15, row = (lane>>4) + 4*reg)
// The C/D row map of this instruction equals the B operand's k map, so accumulator register r of
// output tile t IS the B operand of k-step 4t+r of the next layer: activations never leave the
// register file between layers (bias = initial accumulator, tanh applied in place).
//
// LDS image of a hidden->hidden weight matrix: row-major [h_out][S] with the column index
// XOR-swizzled, col' = col ^ f(row), f(j) = ((j&1)<<4) | (((j>>1)&7)<<1).  With ds_read_b64
// (64 banks x 4 B, conflicts per 32-lane half) both the forward fragment read (16 rows x 2 cols)
// and the transposed read used by the backward pass (2 rows x 16 cols) are conflict-free.
//
// First layer (d <= 4 inputs) and last layer (o <= 4 outputs) are thin and run on the VALU; the
// last layer's dot product is finished with two cross-lane adds (lanes l, l^16, l^32 hold the same
// data row).  SSE partials are reduced in a fixed order (bitwise reproducible).
#include "qn_common.h"
#include "qn_math.h"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int WG = 256;      // 4 waves
constexpr int DMAX = 4;
constexpr int OMAX = 4;

struct FusedArgs {
    int64_t p;
    int B, N, Nb, d, o, nhid, act, has_bias;
    int nsplit, rows_per_split, iters;
};

__host__ __device__ constexpr int swz(int j) { return ((j & 1) << 4) | (((j >> 1) & 7) << 1); }
__host__ __device__ constexpr int stride_of(int H) { return (H + 31) / 32 * 32; }

// LDS image (in doubles): W0 [H][DP] | b0 [H] | (NH-1) x { W [H][S] swizzled | b [H] } | Wl [o][H] | bl [o]
__host__ __device__ inline int lds_doubles(int H, int dp, int o, int nhid) {
    return H * dp + H + (nhid - 1) * (H * stride_of(H) + H) + o * H + o + 8;   // + reduction scratch
}
inline int padded_d(int d) { return d <= 2 ? 2 : 4; }

template <int ACT> __device__ __forceinline__ double act_apply(double z) {
    if constexpr (ACT == QN_ACT_TANH) return qn_tanh_f64(z);
    else if constexpr (ACT == QN_ACT_RELU) return z > 0.0 ? z : 0.0;
    else return z;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Copy weight vector `Wb` into the LDS image.  Loads are issued in batches (all of a layer's
// loads in flight before the first LDS write): a load->wait->write loop costs one memory round
// trip per 2 KB and was ~15 % of the kernel.
template <int H, int DP>
__device__ __forceinline__ void stage_weights(double* __restrict__ lds, const double* __restrict__ Wb,
                                              const FusedArgs& a) {
    constexpr int S = stride_of(H);
    constexpr int PER = (H * H + WG - 1) / WG;
    const int tid = threadIdx.x;
    const int d = a.d, o = a.o;
    const int nb = a.has_bias ? 1 : 0;
    // thin pieces first (W0 padded to DP columns, biases, last layer): at most a few loads per thread
    {
        const int64_t gW0 = 0, gb0 = (int64_t)H * d;
        const int64_t gHH = gb0 + nb * H;                                  // first hidden->hidden block
        const int64_t gWl = gHH + (int64_t)(a.nhid - 1) * (H * H + nb * H);
        const int64_t gbl = gWl + (int64_t)o * H;
        const int lW0 = 0, lb0 = H * DP, lHH = lb0 + H;
        const int lWl = lHH + (a.nhid - 1) * (H * S + H), lbl = lWl + o * H;
        for (int e = tid; e < H * DP; e += WG) {
            const int j = e / DP, k = e % DP;
            lds[lW0 + e] = k < d ? Wb[gW0 + j * d + k] : 0.0;
        }
        for (int e = tid; e < H; e += WG) lds[lb0 + e] = nb ? Wb[gb0 + e] : 0.0;
        for (int layer = 1; layer < a.nhid; ++layer)
            for (int e = tid; e < H; e += WG)
                lds[lHH + (layer - 1) * (H * S + H) + H * S + e] =
                    nb ? Wb[gHH + (int64_t)(layer - 1) * (H * H + H) + H * H + e] : 0.0;
        for (int e = tid; e < o * H; e += WG) lds[lWl + e] = Wb[gWl + e];
        for (int e = tid; e < o; e += WG) lds[lbl + e] = nb ? Wb[gbl + e] : 0.0;
    }
    // hidden->hidden matrices, swizzled
    int64_t g = (int64_t)H * d + nb * H;
    int l = H * DP + H;
    for (int layer = 1; layer < a.nhid; ++layer) {
        double v[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = tid + u * WG;
            v[u] = (H * H % WG == 0 || e < H * H) ? Wb[g + e] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = tid + u * WG;
            if (H * H % WG == 0 || e < H * H) {
                const int j = e / H, i = e % H;
                lds[l + j * S + (i ^ swz(j))] = v[u];
            }
        }
        g += H * H + nb * H;
        l += H * S + H;
    }
}

template <int H, int G, int ACT, int DP>
__global__ __launch_bounds__(WG, 2) void k_fused_fwd_f64(FusedArgs a, const double* __restrict__ W,
                                                      const double* __restrict__ X, const double* __restrict__ Y,
                                                      const int32_t* __restrict__ row_idx,
                                                      double* __restrict__ pred_out, double* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    constexpr int T = H / 16;
    constexpr int S = stride_of(H);
    const int b = blockIdx.y, split = blockIdx.x;
    const int d = a.d, o = a.o, NH = a.nhid;
    const int offW0 = 0, offb0 = H * DP, offHH = offb0 + H;
    const int offWl = offHH + (NH - 1) * (H * S + H), offbl = offWl + o * H;
    double* red = lds + ((offbl + o + 1) & ~1);      // 4 doubles behind the weight image

    stage_weights<H, DP>(lds, W + (int64_t)b * a.p, a);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, c = lane & 15;
    const int fl = swz(c);
    double sse = 0.0;

    for (int it = 0; it < a.iters; ++it) {
        const int nbase = split * a.rows_per_split + (it * (WG / 64) + wave) * 16 * G;
        double act[G][T][4];
        int64_t rrow[G];
        int nrow[G];
        bool valid[G];
        // ---- first layer (VALU): a_1 = act(W0 x + b0)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int n = nbase + 16 * g + c;
            valid[g] = n < a.Nb;
            nrow[g] = n;
            const int nn = valid[g] ? n : 0;
            rrow[g] = row_idx ? (int64_t)row_idx[(int64_t)b * a.Nb + nn] : (int64_t)nn;
            double xk[DP];
#pragma unroll
            for (int k = 0; k < DP; ++k) xk[k] = k < d ? X[rrow[g] * d + k] : 0.0;
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int j = 16 * t + q + 4 * i;
                    double z = lds[offb0 + j];
#pragma unroll
                    for (int k = 0; k < DP; ++k) z = fma(lds[offW0 + j * DP + k], xk[k], z);
                    act[g][t][i] = act_apply<ACT>(z);
                }
        }
        // ---- hidden -> hidden layers on the matrix cores
        for (int layer = 1; layer < NH; ++layer) {
            const double* Wl = lds + offHH + (layer - 1) * (H * S + H);
            const double* bl = Wl + H * S;
            v4d acc[G][T];
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double bj = bl[16 * t + q + 4 * i];
#pragma unroll
                    for (int g = 0; g < G; ++g) acc[g][t][i] = bj;
                }
#pragma unroll
            for (int s = 0; s < H / 4; ++s) {
                const int col = (4 * s + q) ^ fl;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const double aw = Wl[(16 * t + c) * S + col];
#pragma unroll
                    for (int g = 0; g < G; ++g)
                        acc[g][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, act[g][s >> 2][s & 3], acc[g][t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) act[g][t][i] = act_apply<ACT>(acc[g][t][i]);
        }
        // ---- last layer (VALU + 2 cross-lane adds), residual, SSE
#pragma unroll
        for (int g = 0; g < G; ++g) {
            for (int qo = 0; qo < o; ++qo) {
                double part = 0.0;
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        part = fma(lds[offWl + qo * H + 16 * t + q + 4 * i], act[g][t][i], part);
                part += __shfl_xor(part, 16, 64);
                part += __shfl_xor(part, 32, 64);
                const double pr = part + lds[offbl + qo];
                const double res = pr - Y[rrow[g] * o + qo];
                if (valid[g] && q == 0) {
                    sse += res * res;
                    if (pred_out) pred_out[((int64_t)b * a.Nb + nrow[g]) * o + qo] = pr;
                }
            }
        }
    }
    sse = wave_sum(sse);
    if (lane == 0) red[wave] = sse;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < WG / 64; ++w) s += red[w];
        partial[(int64_t)b * a.nsplit + split] = s;
    }
}

__global__ void k_sum_partials(const double* __restrict__ partial, int n, int B, double* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += partial[(int64_t)b * n + i];
    out[b] = s;
}

bool uniform_hidden(const qn_desc* d, int* H, int* nhid) {
    if (d->nlayers < 2) return false;
    const int h = d->dims[1];
    for (int l = 1; l < d->nlayers; ++l)
        if (d->dims[l] != h) return false;
    *H = h;
    *nhid = d->nlayers - 1;
    return true;
}

constexpr int G_FWD = 2;

void plan(const qn_desc* d, int B, int Nb, int G, FusedArgs* a) {
    const int rows_it = (WG / 64) * 16 * G;              // rows one workgroup covers per iteration
    const int max_split = (Nb + rows_it - 1) / rows_it;
    int nsplit = (512 + B - 1) / B;                      // aim at >= 2 workgroups per CU
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    int rps = (Nb + nsplit - 1) / nsplit;
    rps = (rps + rows_it - 1) / rows_it * rows_it;
    nsplit = (Nb + rps - 1) / rps;
    a->nsplit = nsplit;
    a->rows_per_split = rps;
    a->iters = rps / rows_it;
}

}  // namespace

bool qn_fused_supported(const qn_desc* d, int B, int Nb, int want_grad, int dtype) {
    int H, nhid;
    if (dtype != QN_F64 || want_grad) return false;
    if (!uniform_hidden(d, &H, &nhid)) return false;
    if (H != 16 && H != 32 && H != 64) return false;
    if (d->dims[0] > DMAX || d->dims[d->nlayers] > OMAX) return false;
    const size_t bytes = (size_t)lds_doubles(H, padded_d(d->dims[0]), d->dims[d->nlayers], nhid) * sizeof(double);
    return bytes <= 150 * 1024;
}

size_t qn_fused_workspace(const qn_desc* d, int B, int Nb, int want_grad, int dtype) {
    FusedArgs a;
    plan(d, B, Nb, G_FWD, &a);
    return qn_align((size_t)B * a.nsplit * sizeof(double)) + 256;
}

int qn_fused_run(const qn_desc* d, int dtype, const void* W, const void* X, const void* Y, const int32_t* row_idx,
                 int B, int N, int Nb, double* sse, void* pred, void* gradW, void* ws, size_t ws_bytes,
                 hipStream_t st) {
    int H, nhid;
    if (!uniform_hidden(d, &H, &nhid) || gradW || dtype != QN_F64) {
        qn_set_error("qn_fused_run: unsupported configuration");
        return QN_EUNSUPPORTED;
    }
    FusedArgs a;
    a.p = d->p; a.B = B; a.N = N; a.Nb = Nb; a.d = d->dims[0]; a.o = d->dims[d->nlayers]; a.nhid = nhid;
    a.act = d->act; a.has_bias = d->has_bias;
    plan(d, B, Nb, G_FWD, &a);
    const size_t need = qn_align((size_t)B * a.nsplit * sizeof(double));
    if (need > ws_bytes) {
        qn_set_error("workspace too small: need %zu bytes, got %zu", need, ws_bytes);
        return QN_EWORKSPACE;
    }
    double* partial = static_cast<double*>(ws);
    const int dp = padded_d(a.d);
    const size_t lds_bytes = (size_t)lds_doubles(H, dp, a.o, nhid) * sizeof(double);
    dim3 grid(a.nsplit, B);
    (void)hipGetLastError();
    using fwd_fn = void (*)(FusedArgs, const double*, const double*, const double*, const int32_t*, double*, double*);
    fwd_fn kern = nullptr;
#define QN_PICK(HH, AA, DD) if (H == HH && a.act == AA && dp == DD) kern = k_fused_fwd_f64<HH, G_FWD, AA, DD>;
#define QN_PICK_H(HH)                                                                          \
    QN_PICK(HH, QN_ACT_TANH, 2) QN_PICK(HH, QN_ACT_TANH, 4) QN_PICK(HH, QN_ACT_RELU, 2)        \
    QN_PICK(HH, QN_ACT_RELU, 4) QN_PICK(HH, QN_ACT_IDENTITY, 2) QN_PICK(HH, QN_ACT_IDENTITY, 4)
    QN_PICK_H(16) QN_PICK_H(32) QN_PICK_H(64)
#undef QN_PICK_H
#undef QN_PICK
    if (!kern) {
        qn_set_error("qn_fused_run: no kernel instance for H=%d act=%d", H, a.act);
        return QN_EUNSUPPORTED;
    }
    QN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024));
    hipLaunchKernelGGL(kern, grid, dim3(WG), lds_bytes, st, a, (const double*)W, (const double*)X, (const double*)Y,
                       row_idx, (double*)pred, partial);
    hipLaunchKernelGGL(k_sum_partials, dim3((B + 63) / 64), dim3(64), 0, st, partial, a.nsplit, B, sse);
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}
