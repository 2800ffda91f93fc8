// Fused batched-MLP kernels for gfx950 with the chain's weights resident in LDS.
//
// Work decomposition.  1-D XCD-aware grid over (split s, chain b) (qn_fused_args.h): workgroup (s, b) stages weight vector b into LDS once
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
// First layer (d <= 4 inputs; forward kernel: <= 16) and last layer (o <= 4 outputs; forward: <= 16) are thin and run on the VALU; the
// last layer's dot product is finished with two cross-lane adds (lanes l, l^16, l^32 hold the same
// data row).  SSE partials are reduced in a fixed order (bitwise reproducible).
#include "qn_common.h"
#include "qn_fused_args.h"
#include "qn_math.h"
#include <cstdlib>
#include <type_traits>
#include <mutex>
#include <unordered_set>

// Compiled as two objects: part 0 (this file) holds the kernels' instances with up to 4 inputs (forward, tanh: up to 16) and all
// the host code; part 1 (qn_fused_d8.hip: `#define QN_FUSED_PART 1` + `#include` of this file) the instances for networks with
// 5..16 inputs -- the gradient kernel k_fused_bwd_f64<H, NH, 8 | 16, UNB> and the relu / identity forward k_fused_fwd_f64<H, G, ACT, 8 | 16>
// (round 4; until then such gradients ran on the layer-wise kernels) -- behind two pick functions; part 2 (qn_fused_o16.hip) the gradient
// kernel's instances for 5..16 outputs.
#ifndef QN_FUSED_PART
#define QN_FUSED_PART 0
#endif

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int WG = 256;      // 4 waves
constexpr int DMAX = 4;
constexpr int OMAX = 4;

__host__ __device__ constexpr int swz(int j) { return ((j & 1) << 4) | (((j >> 1) & 7) << 1); }
__host__ __device__ constexpr int stride_of(int H) { return (H + 31) / 32 * 32; }

// LDS image (in doubles): W0 [H][DP] | b0 [H] | (NH-1) x { W [H][S] swizzled | b [H] } | Wl [o][H] | bl [o]
__host__ __device__ inline int lds_doubles(int H, int dp, int o, int nhid) {
    return H * dp + H + (nhid - 1) * (H * stride_of(H) + H) + o * H + o + 8;   // + reduction scratch
}
constexpr int TANH_TAB = QN_TANH_LDS_DOUBLES;           // doubles reserved for the tanh table behind an image
inline int padded_d(int d) { return d <= 2 ? 2 : d <= 4 ? 4 : d <= 8 ? 8 : 16; }
constexpr int DWIDE = 16, OWIDE = 16;    // forward kernel (tanh): up to 16 inputs / outputs; backward: DMAX / OMAX

template <int ACT, bool NANSAFE = true> __device__ __forceinline__ double act_apply(double z, const double* tab) {
    if constexpr (ACT == QN_ACT_TANH) return qn_tanh_f64_tab<NANSAFE>(z, tab);
    else if constexpr (ACT == QN_ACT_RELU) return qn_relu<double>(z);
    else return z;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Workgroup-wide OR through a word of the kernel's dynamic LDS (__syncthreads_or would add a static
// LDS variable on top of the 160 KB the backward kernel asks for).  Ends with a barrier.
__device__ __forceinline__ bool block_or(int mine, double* slot) {
    int* flag = reinterpret_cast<int*>(slot);
    if (threadIdx.x == 0) *flag = 0;
    __syncthreads();
    if (mine) *flag = 1;
    __syncthreads();
    return *flag != 0;
}

// Copy weight vector `Wb` into the LDS image.  Loads are issued in batches (all of a layer's
// loads in flight before the first LDS write): a load->wait->write loop costs one memory round
// trip per 2 KB and was ~15 % of the kernel.
// Returns (per thread) whether any weight it copied is NaN / inf / >= 2^500 in magnitude: the caller ORs
// this over the workgroup and then knows whether the NaN-free tanh may be used (qn_math.h).
template <int H, int DP, int NT = WG>
__device__ __forceinline__ int stage_weights(double* __restrict__ lds, const double* __restrict__ Wb,
                                             const FusedArgs& a) {
    int bad = 0;
    auto chk = [&](double v) { bad |= !qn_bounded(v); return v; };
    constexpr int S = stride_of(H);
    constexpr int PER = (H * H + NT - 1) / NT;
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
        for (int e = tid; e < H * DP; e += NT) {
            const int j = e / DP, k = e % DP;
            lds[lW0 + e] = k < d ? chk(Wb[gW0 + j * d + k]) : 0.0;
        }
        for (int e = tid; e < H; e += NT) lds[lb0 + e] = nb ? chk(Wb[gb0 + e]) : 0.0;
        for (int layer = 1; layer < a.nhid; ++layer)
            for (int e = tid; e < H; e += NT)
                lds[lHH + (layer - 1) * (H * S + H) + H * S + e] =
                    nb ? chk(Wb[gHH + (int64_t)(layer - 1) * (H * H + H) + H * H + e]) : 0.0;
        for (int e = tid; e < o * H; e += NT) lds[lWl + e] = chk(Wb[gWl + e]);
        for (int e = tid; e < o; e += NT) lds[lbl + e] = nb ? chk(Wb[gbl + e]) : 0.0;
    }
    // hidden->hidden matrices, swizzled
    int64_t g = (int64_t)H * d + nb * H;
    int l = H * DP + H;
    for (int layer = 1; layer < a.nhid; ++layer) {
        double v[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = tid + u * NT;
            v[u] = (H * H % NT == 0 || e < H * H) ? Wb[g + e] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = tid + u * NT;
            if (H * H % NT == 0 || e < H * H) {
                const int j = e / H, i = e % H;
                lds[l + j * S + (i ^ swz(j))] = chk(v[u]);
            }
        }
        g += H * H + nb * H;
        l += H * S + H;
    }
    return bad;
}

// OM = 4: targets prefetched with the inputs; OM = 16 (more than 4 outputs): targets read where they are used
template <int H, int G, int ACT, int DP, int NT, int OM = OMAX>
__global__ __launch_bounds__(NT, NT / 128) void k_fused_fwd_f64(FusedArgs a, const double* __restrict__ W,
                                                      const double* __restrict__ X, const double* __restrict__ Y,
                                                      const int32_t* __restrict__ row_idx,
                                                      double* __restrict__ pred_out, double* __restrict__ partial,
                                                      unsigned long long* __restrict__ arrive, double* __restrict__ sse_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    constexpr int T = H / 16;
    constexpr int S = stride_of(H);
    int b, split;
    if (!qn_fused_wg(a.nsplit, a.B, &b, &split)) return;
    const int d = a.d, o = a.o, NH = a.nhid;
    const int offW0 = 0, offb0 = H * DP, offHH = offb0 + H;
    const int offWl = offHH + (NH - 1) * (H * S + H), offbl = offWl + o * H;
    double* red = lds + ((offbl + o + 1) & ~1);      // 4 doubles behind the weight image
    const double* tanh_tab = lds + ((lds_doubles(H, DP, o, NH) + 1) & ~1);
    qn_tanh_table_stage(lds + ((lds_doubles(H, DP, o, NH) + 1) & ~1), threadIdx.x, NT);   // barrier: block_or below

    // NaN-free tanh only if nothing can produce a NaN: all weights of this chain bounded (checked while
    // staging) and, per wave iteration, all inputs of its rows bounded (checked in fetch)
    const bool w_unbounded = block_or(stage_weights<H, DP, NT>(lds, W + (int64_t)b * a.p, a), red + 6);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, c = lane & 15;
    const int fl = swz(c);
    double sse = 0.0;

    // data of the NEXT iteration is fetched while the current one computes (x, y and row indices come
    // from HBM/L2; their latency would otherwise be exposed once per iteration)
    constexpr bool YPRE = OM <= OMAX;             // prefetch the targets (few outputs) or read them late
    constexpr int OY = YPRE ? OM : 1;
    constexpr bool XPRE = DP <= DMAX;             // likewise the inputs: more than 4 are read at the start of the iteration
    constexpr int DX = XPRE ? DP : 1;
    double xn[G][DX], yn[G][OY];
    int64_t rr_n[G];
    int nrow_n[G];
    bool valid_n[G];
    int xbad_n = 0;
    auto fetch = [&](int it) {
        xbad_n = 0;
        const int nbase = split * a.rows_per_split + (it * (NT / 64) + wave) * 16 * G;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int n = nbase + 16 * g + c;
            valid_n[g] = n < a.Nb;
            nrow_n[g] = n;
            const int nn = valid_n[g] ? n : 0;
            const int64_t rr = row_idx ? (int64_t)row_idx[(int64_t)b * a.Nb + nn] : (int64_t)nn;
#pragma unroll
            for (int k = 0; k < DX; ++k) {
                xn[g][k] = (XPRE && k < d) ? X[rr * d + k] : 0.0;
                xbad_n |= !qn_bounded(xn[g][k]);
            }
#pragma unroll
            for (int qo = 0; qo < OY; ++qo) yn[g][qo] = (YPRE && qo < o) ? Y[rr * o + qo] : 0.0;
            rr_n[g] = rr;
        }
    };
    fetch(0);
    for (int it = 0; it < a.iters; ++it) {
        double act[G][T][4];
        double yk[G][OY];
        int64_t rrk[G];
        int nrow[G];
        bool valid[G];
        double xk[G][DP];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            valid[g] = valid_n[g];
            nrow[g] = nrow_n[g];
            rrk[g] = rr_n[g];
#pragma unroll
            for (int k = 0; k < DP; ++k) {
                if constexpr (XPRE) xk[g][k] = xn[g][k];
                else {
                    xk[g][k] = k < d ? X[rrk[g] * d + k] : 0.0;
                    xbad_n |= !qn_bounded(xk[g][k]);
                }
            }
#pragma unroll
            for (int qo = 0; qo < OY; ++qo) yk[g][qo] = yn[g][qo];
        }
        const bool nan_possible = w_unbounded || __any(xbad_n);
        if (it + 1 < a.iters) fetch(it + 1);
        // ---- first layer (VALU): a_1 = act(W0 x + b0)
        auto first_layer = [&](auto tag) {
            constexpr bool NS = decltype(tag)::value;
#pragma unroll
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int j = 16 * t + q + 4 * i;
                        double z = lds[offb0 + j];
#pragma unroll
                        for (int k = 0; k < DP; ++k) z = fma(lds[offW0 + j * DP + k], xk[g][k], z);
                        act[g][t][i] = act_apply<ACT, NS>(z, tanh_tab);
                        // keep the scheduler from hoisting every element's DP weight reads at once (without it the
                        // 4-input 3x64 kernel spilled 316 bytes per lane: 400 k -> 493 k evals/s at 64 chains x N=4096)
                        if constexpr (DP > DMAX || ((DP > 2 || ACT != QN_ACT_TANH) && H == 64)) __builtin_amdgcn_sched_barrier(0);
                    }
            }
        };
        if (ACT == QN_ACT_TANH && !nan_possible) first_layer(std::false_type{});
        else first_layer(std::true_type{});
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
            auto epilogue = [&](auto tag) {
                constexpr bool NS = decltype(tag)::value;
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int t = 0; t < T; ++t)
#pragma unroll
                        for (int i = 0; i < 4; ++i) act[g][t][i] = act_apply<ACT, NS>(acc[g][t][i], tanh_tab);
            };
            if (ACT == QN_ACT_TANH && !nan_possible) epilogue(std::false_type{});
            else epilogue(std::true_type{});
        }
        // ---- last layer (VALU + 2 cross-lane adds), residual, SSE
#pragma unroll
        for (int g = 0; g < G; ++g) {
            auto output = [&](int qo, double yv) {
                double part = 0.0;
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        part = fma(lds[offWl + qo * H + 16 * t + q + 4 * i], act[g][t][i], part);
                part += __shfl_xor(part, 16, 64);
                part += __shfl_xor(part, 32, 64);
                const double pr = part + lds[offbl + qo];
                const double res = pr - yv;
                if (valid[g] && q == 0) {
                    sse += res * res;
                    if (pred_out) pred_out[((int64_t)b * a.Nb + nrow[g]) * o + qo] = pr;
                }
            };
            if constexpr (YPRE) {
#pragma unroll
                for (int qo = 0; qo < OM; ++qo) {
                    if (qo >= o) break;
                    output(qo, yk[g][qo]);
                }
            } else {
                for (int qo = 0; qo < o; ++qo) output(qo, Y[rrk[g] * o + qo]);     // more than 4 outputs: a plain loop
            }
        }
    }
    sse = wave_sum(sse);
    if (lane == 0) red[wave] = sse;
    __syncthreads();
    if (threadIdx.x < 64) {                        // (the first wave, every lane with the same sum)
        double s = 0.0;
        for (int w = 0; w < NT / 64; ++w) s += red[w];
        qn_sse_finish(partial, arrive, sse_out, b, split, a.nsplit, s);
    }
}


// =============================================================================================
// Streaming forward for H = 128 (float64).  One 128 x 128 float64 matrix is 128 KB, so the image of
// ALL layers no longer fits the 160 KB of LDS; this kernel keeps the thin pieces (first / last layer,
// every bias) resident and streams the hidden matrices through ONE 128 KB buffer.  A workgroup is
// 8 waves (2 per SIMD), 128 data rows per iteration, one workgroup per CU; activations stay in
// registers between layers exactly as in k_fused_fwd_f64.
constexpr int HS = 128;
constexpr int NTS = 512;
__host__ __device__ inline int stream_thin_doubles(int dp, int o, int nhid) {
    return HS * dp + HS + (nhid - 1) * HS + o * HS + o + 16;      // + 8 wave sums, flag, padding
}
__host__ __device__ inline int stream_lds_doubles(int dp, int o, int nhid) {
    return ((stream_thin_doubles(dp, o, nhid) + 1) & ~1) + ((TANH_TAB + 1) & ~1) + HS * HS;
}

// The matrix buffer is two K-halves ([128 rows][64 columns] each) filled by
// LDS-DMA (global_load_lds_dwordx4: no registers, asynchronous): while the matrix cores work on one half,
// the other half receives the columns needed next.  One wave-instruction writes 1 KiB = 2 rows of a
// half, lane-linear, so the XOR swizzle is applied to the SOURCE column of each lane.  Two barriers
// per layer (one per half); every DMA has a whole half-phase (~16 k cycles) to land.
constexpr int HK = HS / 2;

__device__ __forceinline__ void glds16(const double* src, double* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int ACT, int DP>
__global__ __launch_bounds__(NTS, 1) void k_fused_fwd_stream_f64(FusedArgs a, const double* __restrict__ W,
                                                                   const double* __restrict__ X,
                                                                   const double* __restrict__ Y,
                                                                   const int32_t* __restrict__ row_idx,
                                                                   double* __restrict__ pred_out,
                                                                   double* __restrict__ partial,
                                                                   unsigned long long* __restrict__ arrive,
                                                                   double* __restrict__ sse_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    constexpr int H = HS, T = H / 16, NT = NTS;
    int b, split;
    if (!qn_fused_wg(a.nsplit, a.B, &b, &split)) return;
    const int d = a.d, o = a.o, NH = a.nhid;
    const int nb = a.has_bias ? 1 : 0;
    const int offW0 = 0, offb0 = H * DP, offbh = offb0 + H;
    const int offWl = offbh + (NH - 1) * H, offbl = offWl + o * H;
    double* red = lds + ((offbl + o + 1) & ~1);
    double* tab = lds + ((stream_thin_doubles(DP, o, NH) + 1) & ~1);
    double* Wbuf = tab + ((TANH_TAB + 1) & ~1);
    const double* tanh_tab = tab;
    const double* Wb = W + (int64_t)b * a.p;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int64_t gb0 = (int64_t)H * d, gHH = gb0 + nb * H, blk = (int64_t)H * H + nb * H;
    const int64_t gWl = gHH + (int64_t)(NH - 1) * blk, gbl = gWl + (int64_t)o * H;

    // one K-half of a hidden matrix: 64 wave-instructions of 2 rows x 64 columns, 8 per wave
    auto dma_half = [&](int layer, int half) {
        const double* src = Wb + gHH + (int64_t)(layer - 1) * blk + half * HK;
        double* dst = Wbuf + half * (H * HK);
        int ll = lane;
        asm volatile("" : "+v"(ll));              // source offsets are recomputed here, not kept live
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int chunk = wave * 8 + u;
            const int j = 2 * chunk + (ll >> 5);
            const int kk = (2 * (ll & 31)) ^ swz(j);
            glds16(src + j * H + kk, dst + chunk * 128);
        }
    };
    const bool restage = NH > 2;                  // with one hidden matrix it is staged once
    if (NH > 1) {
        dma_half(1, 0);
        if (!restage) dma_half(1, 1);
    }

    qn_tanh_table_stage(tab, tid, NT);
    int bad = 0;
    {
        auto chk = [&](double v) { bad |= !qn_bounded(v); return v; };
        for (int e = tid; e < H * DP; e += NT) {
            const int j = e / DP, k = e % DP;
            lds[offW0 + e] = k < d ? chk(Wb[j * d + k]) : 0.0;
        }
        for (int e = tid; e < H; e += NT) lds[offb0 + e] = nb ? chk(Wb[gb0 + e]) : 0.0;
        for (int layer = 1; layer < NH; ++layer)
            for (int e = tid; e < H; e += NT)
                lds[offbh + (layer - 1) * H + e] = nb ? chk(Wb[gHH + (layer - 1) * blk + H * H + e]) : 0.0;
        for (int e = tid; e < o * H; e += NT) lds[offWl + e] = chk(Wb[gWl + e]);
        for (int e = tid; e < o; e += NT) lds[offbl + e] = nb ? chk(Wb[gbl + e]) : 0.0;
        // the hidden matrices reach LDS by DMA, not through registers: scan them once for unbounded values
        for (int layer = 1; layer < NH; ++layer) {
            const double* src = Wb + gHH + (int64_t)(layer - 1) * blk;
#pragma unroll 8
            for (int e = tid; e < H * H; e += NT) bad |= !qn_bounded(src[e]);
        }
    }
    const bool w_unbounded = block_or(bad, red + 8);       // ends with a barrier (and vmcnt(0): first DMA landed)

    const int q = lane >> 4, c = lane & 15;
    const int fl = swz(c);
    double sse = 0.0;

    double xn[DP], yn[OMAX];
    int nrow_n = 0, xbad_n = 0;
    bool valid_n = false;
    auto fetch = [&](int it) {
        xbad_n = 0;
        const int n = split * a.rows_per_split + (it * (NT / 64) + wave) * 16 + c;
        valid_n = n < a.Nb;
        nrow_n = n;
        const int nn = valid_n ? n : 0;
        const int64_t rr = row_idx ? (int64_t)row_idx[(int64_t)b * a.Nb + nn] : (int64_t)nn;
#pragma unroll
        for (int k = 0; k < DP; ++k) {
            xn[k] = k < d ? X[rr * d + k] : 0.0;
            xbad_n |= !qn_bounded(xn[k]);
        }
#pragma unroll
        for (int qo = 0; qo < OMAX; ++qo) yn[qo] = qo < o ? Y[rr * o + qo] : 0.0;
    };
    fetch(0);
    for (int it = 0; it < a.iters; ++it) {
        double act[T][4];
        double xk[DP], yk[OMAX];
        const bool valid = valid_n;
        const int nrow = nrow_n;
#pragma unroll
        for (int k = 0; k < DP; ++k) xk[k] = xn[k];
#pragma unroll
        for (int qo = 0; qo < OMAX; ++qo) yk[qo] = yn[qo];
        const bool nan_possible = w_unbounded || __any(xbad_n);
        if (it + 1 < a.iters) fetch(it + 1);
        auto first_layer = [&](auto tag) {
            constexpr bool NS = decltype(tag)::value;
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int j = 16 * t + q + 4 * i;
                    double z = lds[offb0 + j];
#pragma unroll
                    for (int k = 0; k < DP; ++k) z = fma(lds[offW0 + j * DP + k], xk[k], z);
                    act[t][i] = act_apply<ACT, NS>(z, tanh_tab);
                }
        };
        if (ACT == QN_ACT_TANH && !nan_possible) first_layer(std::false_type{});
        else first_layer(std::true_type{});
        for (int layer = 1; layer < NH; ++layer) {
            const double* bl = lds + offbh + (layer - 1) * H;
            int fls = fl;
            asm volatile("" : "+v"(fls));         // the 16 swizzled fragment columns are recomputed per layer
            v4d acc[T];
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][i] = bl[16 * t + q + 4 * i];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (restage) {
                    // this half has landed for every wave; every wave is done reading the other half
                    __syncthreads();
                    if (half == 0) dma_half(layer, 1);
                    else if (layer + 1 < NH || it + 1 < a.iters) dma_half(layer + 1 < NH ? layer + 1 : 1, 0);
                }
                const double* Wh = Wbuf + half * (H * HK);
#pragma unroll
                for (int s = 0; s < HK / 4; ++s) {
                    const int col = (4 * s + q) ^ fls;
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        const double aw = Wh[(16 * t + c) * HK + col];
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, act[(16 * half + s) >> 2][s & 3], acc[t], 0, 0, 0);
                    }
                }
            }
            auto epilogue = [&](auto tag) {
                constexpr bool NS = decltype(tag)::value;
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) act[t][i] = act_apply<ACT, NS>(acc[t][i], tanh_tab);
            };
            if (ACT == QN_ACT_TANH && !nan_possible) epilogue(std::false_type{});
            else epilogue(std::true_type{});
        }
#pragma unroll
        for (int qo = 0; qo < OMAX; ++qo) {
            if (qo >= o) break;
            double part = 0.0;
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) part = fma(lds[offWl + qo * H + 16 * t + q + 4 * i], act[t][i], part);
            part += __shfl_xor(part, 16, 64);
            part += __shfl_xor(part, 32, 64);
            const double pr = part + lds[offbl + qo];
            const double res = pr - yk[qo];
            if (valid && q == 0) {
                sse += res * res;
                if (pred_out) pred_out[((int64_t)b * a.Nb + nrow) * o + qo] = pr;
            }
        }
    }
    sse = wave_sum(sse);
    if (lane == 0) red[wave] = sse;
    __syncthreads();
    if (threadIdx.x < 64) {                        // (the first wave, every lane with the same sum)
        double s = 0.0;
        for (int w = 0; w < NT / 64; ++w) s += red[w];
        qn_sse_finish(partial, arrive, sse_out, b, split, a.nsplit, s);
    }
}

// =============================================================================================
// Fused forward + backward (float64): gradient of the SSE w.r.t. every weight, per chain.
//
// One workgroup = 4 waves = 64 data rows per iteration (one 16-row group per wave), one chain.
// Forward as above, but every hidden activation a_1..a_NH stays in registers.  Backward, per layer:
//   dA^T = W^T . dZ^T        MFMA; A operand = transposed W fragment straight from the swizzled LDS
//                            image (conflict-free), B operand = dZ in accumulator layout -> registers
//   dW  += dZ^T . A          the contraction runs over DATA ROWS, which sit on the lane index in the
//                            accumulator layout, so dZ and A are transposed through two LDS stashes
//                            SD, SA ([feature][64 rows + 2], stride 66 -> conflict-free fragment
//                            reads); wave w owns output tiles {w, w+4, ...} and accumulates them in
//                            registers over all iterations of the workgroup
//   db, dW_first, dW_last    column sums over the stashes on the VALU (thread = (feature, row part))
// At the end each workgroup writes its partial gradient to a slab [B][nsplit][p]; k_grad_reduce sums
// the slabs in a fixed order (bitwise reproducible).  LDS at 3x64: 68.7 KB image + 2 x 33.8 KB
// stashes -> one workgroup per CU, one wave per SIMD (the DP pipe is serial anyway, section 4.1 of
// DESIGN.md).
constexpr int NHMAX = 4;
#ifdef QN_BWD_STAMPS
// diagnostic build only (tools/bwd_stamps.sh): accumulate per-phase cycles of wave 0 in scalar sums
#define QN_STAMP(k)                                                                          \
    do {                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        const long long now_ = __builtin_amdgcn_s_memtime();                                 \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                  \
        stamp_acc[k] += now_ - stamp_prev;                                                   \
        stamp_prev = now_;                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    } while (0)
#else
#define QN_STAMP(k) do { } while (0)
#endif
constexpr int ROWS_IT = 64;          // rows per workgroup iteration in the backward kernel
constexpr int NSP = ROWS_IT + 2;     // stash row stride (doubles)

__host__ __device__ inline int bwd_lds_doubles(int H, int dp, int o, int nhid, int om = OMAX) {
    return lds_doubles(H, dp, o, nhid) + 2 * H * NSP + ROWS_IT * dp + ROWS_IT * om + 8;
}

// activation of one 16x16 tile (4 values per lane); the switch is wave-uniform and sits OUTSIDE the
// element loop, the sched_barrier keeps the scheduler from interleaving more than one tile's tanh
// chains (register pressure)
__device__ __forceinline__ void act_tile(const v4d& z, double (&out)[4], int act, bool nan_possible,
                                         const double* tanh_tab) {
    if (act == QN_ACT_TANH) {
        if (nan_possible) {
#pragma unroll
            for (int i = 0; i < 4; ++i) out[i] = qn_tanh_f64_tab<true>(z[i], tanh_tab);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) out[i] = qn_tanh_f64_tab<false>(z[i], tanh_tab);
        }
    } else if (act == QN_ACT_RELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = qn_relu<double>(z[i]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = z[i];
    }
    __builtin_amdgcn_sched_barrier(0);
}

// UNB: the activation is unbounded (relu / identity): the masked rows of a ragged tail are zeroed outright (below).  A
// template parameter, not a test of the runtime activation: the test alone cost the tanh instance 3.8 % (tools/ab_grad_mask.sh).
// OM: padded output count of the thin last layer (OMAX = 4; OWIDE = 16 for networks with 5..16 outputs: part 2, qn_fused_o16.hip)
template <int H, int NH, int DP, bool UNB, int OM = OMAX>
__global__ __launch_bounds__(WG, 1) void k_fused_bwd_f64(FusedArgs a, const double* __restrict__ W,
                                                         const double* __restrict__ X, const double* __restrict__ Y,
                                                         const int32_t* __restrict__ row_idx,
                                                         double* __restrict__ pred_out, double* __restrict__ partial,
                                                         double* __restrict__ slab, const int* __restrict__ only_flagged,
                                                         double* __restrict__ gradW, double* __restrict__ sse_out,
                                                         unsigned long long* __restrict__ arrive) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    constexpr int T = H / 16;
    constexpr int S = stride_of(H);
    constexpr int TT = T * T;
    constexpr int TPW = (TT + 3) / 4;            // dW tiles per wave
    constexpr int TPF = WG / H;                  // threads per feature in the column sums
    constexpr int RPT = ROWS_IT / TPF;           // rows per such thread
    int b, split;
    if (!qn_fused_wg(a.nsplit, a.B, &b, &split)) return;
    if (only_flagged) {
        // second pass behind k_fused_bwd_i8 (qn_fused_bwd_i8.hip): only chains with a (chain, split) that left its fast path
        int any = 0;
        for (int k = 0; k < a.nsplit; ++k) any |= only_flagged[b * a.nsplit + k];
        if (!any) {
            // (round 4) this pass also is the slab reduction of the gradient (one launch fewer per gradient evaluation: every
            // kernel boundary costs ~4.4 us of serial time): the chain's nsplit workgroups each sum their share of its elements
            // over the slabs k_fused_bwd_i8 wrote, in the fixed order of k_grad_reduce; workgroup 0 adds the SSE partials
            if (gradW) {
                const int64_t chunk = (a.p + a.nsplit - 1) / a.nsplit, lo = (int64_t)split * chunk, hi = lo + chunk < a.p ? lo + chunk : a.p;
                for (int64_t e = lo + threadIdx.x; e < hi; e += WG) {
                    double sacc = 0.0;
                    for (int k = 0; k < a.nsplit; ++k) sacc += slab[((int64_t)b * a.nsplit + k) * a.p + e];
                    gradW[(int64_t)b * a.p + e] = sacc;
                }
                if (split == 0 && threadIdx.x == 0) {
                    double sacc = 0.0;
                    for (int i = 0; i < a.nsplit; ++i) sacc += partial[(int64_t)b * a.nsplit + i];
                    sse_out[b] = sacc;
                }
            }
            return;
        }
    }
    const int d = a.d, o = a.o, act_kind = a.act;
    const int nb = a.has_bias ? 1 : 0;
    const int offW0 = 0, offb0 = H * DP, offHH = offb0 + H;
    const int offWl = offHH + (NH - 1) * (H * S + H), offbl = offWl + o * H;
    const int img = lds_doubles(H, DP, o, NH);
    double* SA = lds + ((img + 1) & ~1);
    double* SD = SA + H * NSP;
    double* Sx = SD + H * NSP;
    double* Sdl = Sx + ROWS_IT * DP;
    double* red = Sdl + ROWS_IT * OM;
    const double* tanh_tab = lds + ((bwd_lds_doubles(H, DP, o, NH, OM) + 1) & ~1);
    qn_tanh_table_stage(lds + ((bwd_lds_doubles(H, DP, o, NH, OM) + 1) & ~1), threadIdx.x, WG);   // barrier: block_or below

    const bool w_unbounded = block_or(stage_weights<H, DP>(lds, W + (int64_t)b * a.p, a), red + 6);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, c = lane & 15;
    const int fl = swz(c);
    int swq[4];                                   // swz(4*s + q) for s & 3 = 0..3
#pragma unroll
    for (int m = 0; m < 4; ++m) swq[m] = swz(4 * m + q);
    const int fj = tid / TPF, part = tid % TPF;   // column-sum role
    const int wrow = wave * 16 + c;               // this lane's row inside the 64-row tile

    double sse = 0.0;
#ifdef QN_BWD_STAMPS
    long long stamp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long stamp_prev = __builtin_amdgcn_s_memtime();
#endif
    v4d dWacc[NH > 1 ? NH - 1 : 1][TPW];
#pragma unroll
    for (int l = 0; l < (NH > 1 ? NH - 1 : 1); ++l)
#pragma unroll
        for (int u = 0; u < TPW; ++u) dWacc[l][u] = (v4d){0.0, 0.0, 0.0, 0.0};
    double dbacc[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) dbacc[k] = 0.0;
    double dW0acc[DP], dWlacc[OM], dblacc[OM] = {0.0, 0.0, 0.0, 0.0};
    // o == 1 and d == 1 (every BASELINE config): the thin layers' gradients are accumulated lane-locally in the
    // accumulator layout over all iterations and reduced across lanes / waves ONCE at the end, instead of two
    // stash round trips (4 barriers + column sums) per iteration
    const bool thin_fast = (o == 1 && d == 1);
    double accWl[T][4], accW0[T][4], accB0[T][4], accBl = 0.0;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) accWl[t][i] = accW0[t][i] = accB0[t][i] = 0.0;
#pragma unroll
    for (int k = 0; k < DP; ++k) dW0acc[k] = 0.0;
#pragma unroll
    for (int k = 0; k < OM; ++k) dWlacc[k] = 0.0;

    for (int it = 0; it < a.iters; ++it) {
        const int n = split * a.rows_per_split + it * ROWS_IT + wrow;
        const bool valid = n < a.Nb;
        const int nn = valid ? n : 0;
        const int64_t rrow = row_idx ? (int64_t)row_idx[(int64_t)b * a.Nb + nn] : (int64_t)nn;
        double xk[DP];
        int xbad = 0;
#pragma unroll
        for (int k = 0; k < DP; ++k) {
            xk[k] = k < d ? X[rrow * d + k] : 0.0;
            xbad |= !qn_bounded(xk[k]);
        }
        const bool nan_possible = w_unbounded || __any(xbad);
        // relu / identity activations are unbounded: the masked rows of a ragged tail (they re-run row 0 with a zero
        // residual) must contribute 0 . 0, not 0 . Inf = NaN, so their inputs, activations and dz are zeroed outright
        // (tanh: activations are bounded and the zero residual is enough)
        const bool maskrow = UNB && !valid;
        if (UNB && maskrow) {
#pragma unroll
            for (int k = 0; k < DP; ++k) xk[k] = 0.0;
        }
        QN_STAMP(0);                                       // 0: loop top + x load
        // ------------------------------------------------------------------ forward
        double act[NH][T][4];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            v4d z;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = 16 * t + q + 4 * i;
                z[i] = lds[offb0 + j];
#pragma unroll
                for (int k = 0; k < DP; ++k) z[i] = fma(lds[offW0 + j * DP + k], xk[k], z[i]);
            }
            act_tile(z, act[0][t], act_kind, nan_possible, tanh_tab);
            if (UNB && maskrow) {
#pragma unroll
                for (int i = 0; i < 4; ++i) act[0][t][i] = 0.0;
            }
        }
#pragma unroll
        for (int layer = 1; layer < NH; ++layer) {
            {
                const double* Wl = lds + offHH + (layer - 1) * (H * S + H);
                const double* bl = Wl + H * S;
                v4d acc[T];
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[t][i] = bl[16 * t + q + 4 * i];
                double wf[T], wn[T];
#pragma unroll
                for (int t = 0; t < T; ++t) wn[t] = Wl[(16 * t + c) * S + (q ^ fl)];
#pragma unroll
                for (int s = 0; s < H / 4; ++s) {       // fragments of step s+1 are in flight under the MFMAs of step s
#pragma unroll
                    for (int t = 0; t < T; ++t) wf[t] = wn[t];
                    if (s + 1 < H / 4) {
                        const int col = (4 * (s + 1) + q) ^ fl;
#pragma unroll
                        for (int t = 0; t < T; ++t) wn[t] = Wl[(16 * t + c) * S + col];
                    }
#pragma unroll
                    for (int t = 0; t < T; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(wf[t], act[layer - 1][s >> 2][s & 3], acc[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    act_tile(acc[t], act[layer][t], act_kind, nan_possible, tanh_tab);
                    if (UNB && maskrow) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) act[layer][t][i] = 0.0;
                    }
                }
            }
        }
        double (&alast)[T][4] = act[NH - 1];
        QN_STAMP(1);                                       // 1: forward layers
        // ------------------------------------------------------------------ last layer, residual
        double delta[OM];
#pragma unroll
        for (int qo = 0; qo < OM; ++qo) {
            delta[qo] = 0.0;
            if (qo < o) {
                double pd = 0.0;
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) pd = fma(lds[offWl + qo * H + 16 * t + q + 4 * i], alast[t][i], pd);
                pd += __shfl_xor(pd, 16, 64);
                pd += __shfl_xor(pd, 32, 64);
                const double pr = pd + lds[offbl + qo];
                const double res = pr - Y[rrow * o + qo];
                if (valid) {
                    delta[qo] = 2.0 * res;
                    if (q == 0) {
                        sse += res * res;
                        if (pred_out) pred_out[((int64_t)b * a.Nb + n) * o + qo] = pr;
                    }
                }
            }
        }
        QN_STAMP(2);                                       // 2: last layer + residual
        // ------------------------------------------------------------------ backward: last layer
        if (thin_fast) {
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) accWl[t][i] = fma(delta[0], alast[t][i], accWl[t][i]);
            if (q == 0) accBl += delta[0];
        } else {
            __syncthreads();                                   // previous iteration's stash readers are done
            QN_STAMP(3);                                       // 3: barrier A (last stage)
    #pragma unroll
            for (int t = 0; t < T; ++t)
    #pragma unroll
                for (int i = 0; i < 4; ++i) SA[(16 * t + q + 4 * i) * NSP + wrow] = alast[t][i];
            if (q == 0) {
    #pragma unroll
                for (int qo = 0; qo < OM; ++qo) Sdl[wrow * OM + qo] = delta[qo];
    #pragma unroll
                for (int k = 0; k < DP; ++k) Sx[wrow * DP + k] = xk[k];
            }
            __syncthreads();
            QN_STAMP(4);                                       // 4: stash write + barrier B (last stage)
            {
                double sum[OM] = {0.0, 0.0, 0.0, 0.0}, sdl[OM] = {0.0, 0.0, 0.0, 0.0};
    #pragma unroll 4
                for (int rr = 0; rr < RPT; ++rr) {
                    const int row = part + TPF * rr;          // interleaved rows: conflict-free Sdl / Sx reads
                    const double av = SA[fj * NSP + row];
    #pragma unroll
                    for (int qo = 0; qo < OM; ++qo) {
                        const double dv = Sdl[row * OM + qo];
                        sum[qo] = fma(av, dv, sum[qo]);
                        sdl[qo] += dv;                        // every feature's threads see all deltas: feature 0 keeps the bias sum
                    }
                }
    #pragma unroll
                for (int qo = 0; qo < OM; ++qo) {
                    dWlacc[qo] += sum[qo];
                    dblacc[qo] += sdl[qo];
                }
            }
        }
        double dz[T][4];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) dz[t][i] = 0.0;
#pragma unroll
        for (int qo = 0; qo < OM; ++qo) {
            if (qo < o) {                                  // one uniform branch per output, 16 LDS reads in flight
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        dz[t][i] = fma(lds[offWl + qo * H + 16 * t + q + 4 * i], delta[qo], dz[t][i]);
            }
        }
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) dz[t][i] = (UNB && maskrow) ? 0.0 : qn_act_bwd<double>(dz[t][i], alast[t][i], act_kind);
        QN_STAMP(5);                                       // 5: last-stage column sums + dz_NH
        // ------------------------------------------------------------------ backward: hidden -> hidden layers
#pragma unroll
        for (int layer = NH - 1; layer >= 1; --layer) {
            {
                const double* Wl = lds + offHH + (layer - 1) * (H * S + H);
                __syncthreads();
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        SD[(16 * t + q + 4 * i) * NSP + wrow] = dz[t][i];
                        SA[(16 * t + q + 4 * i) * NSP + wrow] = act[layer - 1][t][i];
                    }
                __syncthreads();
                QN_STAMP(6);                               // 6: hidden stage barriers + stash writes
                // dW_layer tiles owned by this wave, contraction over the 64 rows of the tile; two tiles
                // are advanced together (independent accumulators back to back), fragments of the next
                // 2 k-steps are in flight under the MFMAs of the current ones
                constexpr int UP = TPW >= 2 ? 2 : 1;
#pragma unroll
                for (int u0 = 0; u0 < TPW; u0 += UP) {
                    const double* pa[UP];
                    const double* pb[UP];
                    bool live[UP];
#pragma unroll
                    for (int v = 0; v < UP; ++v) {
                        const int tile = wave + 4 * (u0 + v);
                        live[v] = tile < TT;
                        const int tj = live[v] ? tile / T : 0, ti = live[v] ? tile % T : 0;
                        pa[v] = SD + (16 * tj + c) * NSP + q;
                        pb[v] = SA + (16 * ti + c) * NSP + q;
                    }
                    constexpr int KB = 2;                              // k-steps per batch
                    double fa[UP][KB], fb[UP][KB], na[UP][KB], nb2[UP][KB];
#pragma unroll
                    for (int v = 0; v < UP; ++v)
#pragma unroll
                        for (int m = 0; m < KB; ++m) { na[v][m] = pa[v][4 * m]; nb2[v][m] = pb[v][4 * m]; }
#pragma unroll
                    for (int sb = 0; sb < ROWS_IT / 4 / KB; ++sb) {
#pragma unroll
                        for (int v = 0; v < UP; ++v)
#pragma unroll
                            for (int m = 0; m < KB; ++m) { fa[v][m] = na[v][m]; fb[v][m] = nb2[v][m]; }
                        if (sb + 1 < ROWS_IT / 4 / KB) {
#pragma unroll
                            for (int v = 0; v < UP; ++v)
#pragma unroll
                                for (int m = 0; m < KB; ++m) {
                                    na[v][m] = pa[v][4 * (KB * (sb + 1) + m)];
                                    nb2[v][m] = pb[v][4 * (KB * (sb + 1) + m)];
                                }
                        }
#pragma unroll
                        for (int m = 0; m < KB; ++m)
#pragma unroll
                            for (int v = 0; v < UP; ++v)
                                if (live[v])
                                    dWacc[layer - 1][u0 + v] = __builtin_amdgcn_mfma_f64_16x16x4f64(
                                        fa[v][m], fb[v][m], dWacc[layer - 1][u0 + v], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                QN_STAMP(7);                               // 7: dW MFMAs
                {   // bias gradient of this layer: column sums of dZ
                    double sb = 0.0;
#pragma unroll 4
                    for (int rr = 0; rr < RPT; ++rr) sb += SD[fj * NSP + part + TPF * rr];
                    dbacc[layer] += sb;
                }
                QN_STAMP(8);                               // 8: db column sums
                // dA = W^T dZ (transposed fragment reads of the same LDS image), then dZ of the layer below
                v4d nd[T];
#pragma unroll
                for (int t = 0; t < T; ++t) nd[t] = (v4d){0.0, 0.0, 0.0, 0.0};
                double tf[T], tn[T];
#pragma unroll
                for (int t = 0; t < T; ++t) tn[t] = Wl[q * S + ((16 * t + c) ^ swq[0])];
#pragma unroll
                for (int s = 0; s < H / 4; ++s) {
#pragma unroll
                    for (int t = 0; t < T; ++t) tf[t] = tn[t];
                    if (s + 1 < H / 4) {
                        const double* wrow_p = Wl + (4 * (s + 1) + q) * S;
                        const int sw = swq[(s + 1) & 3];
#pragma unroll
                        for (int t = 0; t < T; ++t) tn[t] = wrow_p[(16 * t + c) ^ sw];
                    }
#pragma unroll
                    for (int t = 0; t < T; ++t)
                        nd[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(tf[t], dz[s >> 2][s & 3], nd[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) dz[t][i] = (UNB && maskrow) ? 0.0 : qn_act_bwd<double>(nd[t][i], act[layer - 1][t][i], act_kind);
                QN_STAMP(9);                               // 9: dA MFMAs + dz
            }
        }
        // ------------------------------------------------------------------ backward: first layer
        if (thin_fast) {
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    accW0[t][i] = fma(dz[t][i], xk[0], accW0[t][i]);
                    accB0[t][i] += dz[t][i];
                }
        } else {
            __syncthreads();
    #pragma unroll
            for (int t = 0; t < T; ++t)
    #pragma unroll
                for (int i = 0; i < 4; ++i) SD[(16 * t + q + 4 * i) * NSP + wrow] = dz[t][i];
            __syncthreads();
            {
                double sb = 0.0, sw0[DP];
    #pragma unroll
                for (int k = 0; k < DP; ++k) sw0[k] = 0.0;
    #pragma unroll 4
                for (int rr = 0; rr < RPT; ++rr) {
                    const int row = part + TPF * rr;
                    const double g = SD[fj * NSP + row];
                    sb += g;
    #pragma unroll
                    for (int k = 0; k < DP; ++k) sw0[k] = fma(g, Sx[row * DP + k], sw0[k]);
                }
                dbacc[0] += sb;
    #pragma unroll
                for (int k = 0; k < DP; ++k) dW0acc[k] += sw0[k];
            }
        }
        QN_STAMP(10);                                      // 10: first-layer stage (2 barriers + sums)
    }
#ifdef QN_BWD_STAMPS
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        long long* dbg = reinterpret_cast<long long*>(partial + a.dbg_off);
        for (int k = 0; k < 12; ++k) dbg[k] = stamp_acc[k];
    }
#endif

    // ---------------------------------------------------------------------- write the partial gradient
    double* out = slab + ((int64_t)b * a.nsplit + split) * a.p;
    const int64_t gW0 = 0, gb0 = (int64_t)H * d, gHH = gb0 + nb * H;
    const int64_t gWl = gHH + (int64_t)(NH - 1) * (H * H + nb * H), gbl = gWl + (int64_t)o * H;
    // column-sum accumulators: add up the TPF threads of a feature (adjacent lanes)
#pragma unroll
    for (int m = 1; m < TPF; m <<= 1) {
#pragma unroll
        for (int k = 0; k < NH; ++k) dbacc[k] += __shfl_xor(dbacc[k], m, 64);
#pragma unroll
        for (int k = 0; k < DP; ++k) dW0acc[k] += __shfl_xor(dW0acc[k], m, 64);
#pragma unroll
        for (int k = 0; k < OM; ++k) {
            dWlacc[k] += __shfl_xor(dWlacc[k], m, 64);
            dblacc[k] += __shfl_xor(dblacc[k], m, 64);
        }
    }
    if (part == 0) {
        if (!thin_fast) {
#pragma unroll
            for (int k = 0; k < DP; ++k)
                if (k < d) out[gW0 + (int64_t)fj * d + k] = dW0acc[k];
        }
        if (nb) {
            if (!thin_fast) out[gb0 + fj] = dbacc[0];
#pragma unroll
            for (int layer = 1; layer < NH; ++layer)
                out[gHH + (int64_t)(layer - 1) * (H * H + H) + H * H + fj] = dbacc[layer];
        }
        if (!thin_fast) {
#pragma unroll
            for (int qo = 0; qo < OM; ++qo)
                if (qo < o) out[gWl + (int64_t)qo * H + fj] = dWlacc[qo];
        }
    }
    if (!thin_fast && nb && tid == 0) {
#pragma unroll
        for (int qo = 0; qo < OM; ++qo)
            if (qo < o) out[gbl + qo] = dblacc[qo];
    }
    if (thin_fast) {
        // lane-local partials: sum over the 16 row lanes of each q-group, then over the 4 waves through LDS
        __syncthreads();                                   // stashes are free now: reuse SA as scratch
        double* R = SA;                                    // [3][4 waves][H] + [4]
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double v0 = accWl[t][i], v1 = accW0[t][i], v2 = accB0[t][i];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) {
                    v0 += __shfl_xor(v0, m, 64);
                    v1 += __shfl_xor(v1, m, 64);
                    v2 += __shfl_xor(v2, m, 64);
                }
                if (c == 0) {
                    const int f = 16 * t + q + 4 * i;
                    R[(0 * 4 + wave) * H + f] = v0;
                    R[(1 * 4 + wave) * H + f] = v1;
                    R[(2 * 4 + wave) * H + f] = v2;
                }
            }
        const double bl_w = wave_sum(accBl);
        if (lane == 0) R[12 * H + wave] = bl_w;
        __syncthreads();
        if (tid < H) {
            double s0 = 0.0, s1 = 0.0, s2 = 0.0;
            for (int w = 0; w < 4; ++w) {
                s0 += R[(0 * 4 + w) * H + tid];
                s1 += R[(1 * 4 + w) * H + tid];
                s2 += R[(2 * 4 + w) * H + tid];
            }
            out[gWl + tid] = s0;
            out[gW0 + tid] = s1;
            if (nb) out[gb0 + tid] = s2;
        }
        if (tid == 0 && nb) out[gbl] = R[12 * H] + R[12 * H + 1] + R[12 * H + 2] + R[12 * H + 3];
    }
#pragma unroll
    for (int layer = 1; layer < NH; ++layer) {
        {
            double* og = out + gHH + (int64_t)(layer - 1) * (H * H + nb * H);
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                const int tile = wave + 4 * u;
                if (tile < TT) {
                    const int tj = tile / T, ti = tile % T;
#pragma unroll
                    for (int r = 0; r < 4; ++r) og[(int64_t)(16 * tj + q + 4 * r) * H + 16 * ti + c] = dWacc[layer - 1][u][r];
                }
            }
        }
    }
    sse = wave_sum(sse);
    __syncthreads();
    if (lane == 0) red[wave] = sse;
    __syncthreads();
    if (tid == 0) {
        double sum = 0.0;
        for (int w = 0; w < WG / 64; ++w) sum += red[w];
        partial[(int64_t)b * a.nsplit + split] = sum;
    }
    if (only_flagged && gradW) {
        // a FLAGGED chain (rare): its workgroups have just rewritten every split's slab row; the last of them to arrive reduces the
        // chain.  Agent-scope release / acquire around the arrival (the cost does not matter here).
        __threadfence();
        __syncthreads();
        if (tid == 0) reinterpret_cast<unsigned*>(red)[0] = qn_arrive_tagged(arrive, b);
        __syncthreads();
        if (reinterpret_cast<unsigned*>(red)[0] == (unsigned)a.nsplit) {
            __threadfence();
            for (int64_t e = tid; e < a.p; e += WG) {
                double sacc = 0.0;
                for (int k = 0; k < a.nsplit; ++k)
                    sacc += __hip_atomic_load(&slab[((int64_t)b * a.nsplit + k) * a.p + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                gradW[(int64_t)b * a.p + e] = sacc;
            }
            if (tid == 0) {
                double sacc = 0.0;
                for (int i = 0; i < a.nsplit; ++i) sacc += __hip_atomic_load(&partial[(int64_t)b * a.nsplit + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sse_out[b] = sacc;
                __hip_atomic_store(&arrive[b], QN_ARRIVE_MAGIC << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// gradW[b][e] = sum over the nsplit slabs, fixed order; block (0, b) also adds up the chain's SSE partials (left to right, as
// k_sum_partials does: one launch fewer per gradient evaluation)
__global__ __launch_bounds__(256) void k_grad_reduce(const double* __restrict__ slab, int nsplit, int64_t p, int B,
                                                     double* __restrict__ gradW, const double* __restrict__ partial,
                                                     double* __restrict__ sse) {
    const int b = blockIdx.y;
    if (partial && blockIdx.x == 0 && threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < nsplit; ++i) s += partial[(int64_t)b * nsplit + i];
        sse[b] = s;
    }
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < p; e += (int64_t)gridDim.x * 256) {
        double sacc = 0.0;
        for (int k = 0; k < nsplit; ++k) sacc += slab[((int64_t)b * nsplit + k) * p + e];
        gradW[(int64_t)b * p + e] = sacc;
    }
}

__global__ void k_sum_partials(const double* __restrict__ partial, int n, int B, double* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += partial[(int64_t)b * n + i];
    out[b] = s;
}

constexpr int G_FWD = 2;

#if QN_FUSED_PART == 2
}  // namespace

// gradient kernel for networks with 5..16 outputs (OM = OWIDE): 4 or 16 padded input columns
qn_bwd_f64_fn qn_fused_bwd_o16_kernel(int H, int nhid, int act, int dp) {
#define QN_PICK(HH, NN, DD) if (H == HH && nhid == NN && dp == DD) return act == QN_ACT_TANH ? k_fused_bwd_f64<HH, NN, DD, false, OWIDE> : k_fused_bwd_f64<HH, NN, DD, true, OWIDE>;
#define QN_PICK_D(HH, NN) QN_PICK(HH, NN, 4) QN_PICK(HH, NN, 16)
    QN_PICK_D(16, 1) QN_PICK_D(16, 2) QN_PICK_D(16, 3) QN_PICK_D(16, 4)
    QN_PICK_D(32, 1) QN_PICK_D(32, 2) QN_PICK_D(32, 3) QN_PICK_D(32, 4)
    QN_PICK_D(64, 1) QN_PICK_D(64, 2) QN_PICK_D(64, 3)
#undef QN_PICK_D
#undef QN_PICK
    return nullptr;
}
#elif QN_FUSED_PART == 1
}  // namespace

// dp = 8, 16: every (H, NH) the kernel has (no spills; the largest LDS image, 3 x 64 with 16 inputs and 4 outputs, is 158 KB)
qn_bwd_f64_fn qn_fused_bwd_d8_kernel(int H, int nhid, int act, int dp) {
#define QN_PICK(HH, NN, DD) if (H == HH && nhid == NN && dp == DD) return act == QN_ACT_TANH ? k_fused_bwd_f64<HH, NN, DD, false> : k_fused_bwd_f64<HH, NN, DD, true>;
    QN_PICK(16, 1, 8) QN_PICK(16, 2, 8) QN_PICK(16, 3, 8) QN_PICK(16, 4, 8)
    QN_PICK(32, 1, 8) QN_PICK(32, 2, 8) QN_PICK(32, 3, 8) QN_PICK(32, 4, 8)
    QN_PICK(64, 1, 8) QN_PICK(64, 2, 8) QN_PICK(64, 3, 8)
    QN_PICK(16, 1, 16) QN_PICK(16, 2, 16) QN_PICK(16, 3, 16) QN_PICK(16, 4, 16)
    QN_PICK(32, 1, 16) QN_PICK(32, 2, 16) QN_PICK(32, 3, 16) QN_PICK(32, 4, 16)
    QN_PICK(64, 1, 16) QN_PICK(64, 2, 16) QN_PICK(64, 3, 16)
#undef QN_PICK
    return nullptr;
}
// relu / identity forward with a wide first or last layer (tanh: pick_fwd of part 0): more than 4 inputs, or -- `wide_out` -- 5..16 outputs
qn_fwd_fn qn_fused_fwd_d8_kernel(int H, int act, int dp, int wide_out) {
#define QN_PICK(HH, AA, DD) if (H == HH && act == AA && dp == DD && !wide_out) return k_fused_fwd_f64<HH, G_FWD, AA, DD, WG>;
#define QN_PICKO(HH, AA, DD) if (H == HH && act == AA && dp == DD && wide_out) return k_fused_fwd_f64<HH, G_FWD, AA, DD, WG, OWIDE>;
#define QN_PICK_A(HH, DD) QN_PICK(HH, QN_ACT_RELU, DD) QN_PICK(HH, QN_ACT_IDENTITY, DD)
#define QN_PICKO_A(HH, DD) QN_PICKO(HH, QN_ACT_RELU, DD) QN_PICKO(HH, QN_ACT_IDENTITY, DD)
#define QN_PICK_H(HH) QN_PICK_A(HH, 8) QN_PICK_A(HH, 16) QN_PICKO_A(HH, 2) QN_PICKO_A(HH, 4) QN_PICKO_A(HH, 8) QN_PICKO_A(HH, 16)
    QN_PICK_H(16) QN_PICK_H(32) QN_PICK_H(64)
#undef QN_PICK_H
#undef QN_PICKO_A
#undef QN_PICK_A
#undef QN_PICKO
#undef QN_PICK
    return nullptr;
}
#else

bool uniform_hidden(const qn_desc* d, int* H, int* nhid) {
    if (d->nlayers < 2) return false;
    const int h = d->dims[1];
    for (int l = 1; l < d->nlayers; ++l)
        if (d->dims[l] != h) return false;
    *H = h;
    *nhid = d->nlayers - 1;
    return true;
}

bool streams(const qn_desc* d, int want_grad) { return !want_grad && d->nlayers >= 2 && d->dims[1] == HS; }

// the forward of 64-wide tanh networks runs as sliced int8 products (qn_fused_i8.hip) unless the descriptor asks for
// the float64-MFMA kernels (QN_PATH_FUSED_DP)
bool uses_i8(const qn_desc* d, int want_grad) {
    int H, nhid;
    return !want_grad && d->path != QN_PATH_FUSED_DP && uniform_hidden(d, &H, &nhid) &&
           qn_fused_i8_applies(H, nhid, d->act, d->dims[0], d->dims[d->nlayers]);
}

// the gradient of 64-wide tanh networks likewise (qn_fused_bwd_i8.hip), with the float64 kernel behind it for flagged chains
bool uses_i8_bwd(const qn_desc* d) {
    int H, nhid;
    return d->path != QN_PATH_FUSED_DP && uniform_hidden(d, &H, &nhid) &&
           qn_fused_bwd_i8_applies(H, nhid, d->act, d->dims[0], d->dims[d->nlayers]);
}

void plan(const qn_desc* d, int B, int Nb, int want_grad, FusedArgs* a) {
    // rows one workgroup covers per iteration; target workgroups per chip: 2/CU forward, 1/CU backward
    // (and streaming forward: its 128 KB matrix buffer leaves room for one workgroup per CU)
    const bool stream = streams(d, want_grad);
    int rows_it = want_grad ? ROWS_IT : (stream ? (NTS / 64) * 16 : (WG / 64) * 16 * G_FWD);
    if (uses_i8(d, want_grad)) rows_it = qn_fused_i8_rows_per_iteration();
#ifndef QN_FWD_TARGET
#define QN_FWD_TARGET 512
#endif
#ifndef QN_BWD_TARGET
#define QN_BWD_TARGET 256
#endif
    const int target = want_grad || stream ? QN_BWD_TARGET : QN_FWD_TARGET;
    const int max_split = (Nb + rows_it - 1) / rows_it;
    const int Bp = d->plan_batch > B ? d->plan_batch : B;
    int nsplit = (target + Bp - 1) / Bp;
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    int rps = (Nb + nsplit - 1) / nsplit;
    rps = (rps + rows_it - 1) / rows_it * rows_it;
    nsplit = (Nb + rps - 1) / rps;
    a->nsplit = nsplit;
    a->rows_per_split = rps;
    a->iters = rps / rows_it;
}

constexpr int DBWD = 16;                                   // inputs of the gradient kernel: DP = 4 (part 0), 8 or 16 (part 1)
inline int bwd_dp(int d, int o = 1) { return o > OMAX ? (d <= DMAX ? DMAX : 16) : (d <= DMAX ? DMAX : d <= 8 ? 8 : 16); }   // (5..16 outputs: 4 or 16 columns)
inline int bwd_om(int o) { return o > OMAX ? OWIDE : OMAX; }
size_t lds_need(int H, int d, int o, int nhid, int want_grad) {
    if (H == HS) return sizeof(double) * (size_t)(stream_lds_doubles(padded_d(d), o, nhid) + 2);
    return sizeof(double) * (size_t)((want_grad ? bwd_lds_doubles(H, bwd_dp(d, o), o, nhid, bwd_om(o)) : lds_doubles(H, padded_d(d), o, nhid)) +
                                     2 + TANH_TAB);
}

using fwd_fn = qn_fwd_fn;
using bwd_fn = qn_bwd_f64_fn;

// Forward geometry: 4 waves x 2 row groups per workgroup (2 workgroups / CU, 2 waves / SIMD).  The
// alternative 8 waves x 1 row group (4 waves / SIMD, same 128 rows per iteration) measured 3.5 % slower
// at cfg2 (462 k vs 480 k evals/s): occupancy is not the lever on a serial DP pipe.
fwd_fn pick_fwd(int H, int act, int dp, int o) {
    if (dp > DMAX || o > OMAX) {             // wide first / last layer: tanh networks only
#define QN_PICKW(HH, DD)                                                                          \
    if (H == HH && dp == DD)                                                                      \
        return o > OMAX ? k_fused_fwd_f64<HH, G_FWD, QN_ACT_TANH, DD, WG, OWIDE>                  \
                        : k_fused_fwd_f64<HH, G_FWD, QN_ACT_TANH, DD, WG, OMAX>;
        if (act != QN_ACT_TANH) return qn_fused_fwd_d8_kernel(H, act, dp, o > OMAX);
        QN_PICKW(16, 2) QN_PICKW(16, 4) QN_PICKW(16, 8) QN_PICKW(16, 16)
        QN_PICKW(32, 2) QN_PICKW(32, 4) QN_PICKW(32, 8) QN_PICKW(32, 16)
        QN_PICKW(64, 2) QN_PICKW(64, 4) QN_PICKW(64, 8) QN_PICKW(64, 16)
#undef QN_PICKW
        return nullptr;
    }
#define QN_PICK(HH, AA, DD)                                                                    \
    if (H == HH && act == AA && dp == DD)                                                      \
        return k_fused_fwd_f64<HH, G_FWD, AA, DD, WG>;
#define QN_PICK_H(HH)                                                                          \
    QN_PICK(HH, QN_ACT_TANH, 2) QN_PICK(HH, QN_ACT_TANH, 4) QN_PICK(HH, QN_ACT_RELU, 2)        \
    QN_PICK(HH, QN_ACT_RELU, 4) QN_PICK(HH, QN_ACT_IDENTITY, 2) QN_PICK(HH, QN_ACT_IDENTITY, 4)
    QN_PICK_H(16) QN_PICK_H(32) QN_PICK_H(64)
#undef QN_PICK_H
#undef QN_PICK
#define QN_PICK(AA, DD)                                                                        \
    if (H == HS && act == AA && dp == DD) return k_fused_fwd_stream_f64<AA, DD>;
    QN_PICK(QN_ACT_TANH, 2) QN_PICK(QN_ACT_TANH, 4) QN_PICK(QN_ACT_RELU, 2)
    QN_PICK(QN_ACT_RELU, 4) QN_PICK(QN_ACT_IDENTITY, 2) QN_PICK(QN_ACT_IDENTITY, 4)
#undef QN_PICK
    return nullptr;
}

bwd_fn pick_bwd(int H, int nhid, int act = QN_ACT_TANH, int dp = DMAX, int om = OMAX) {
    if (om != OMAX) return qn_fused_bwd_o16_kernel(H, nhid, act, dp);
    if (dp != DMAX) return qn_fused_bwd_d8_kernel(H, nhid, act, dp);
#define QN_PICK(HH, NN) if (H == HH && nhid == NN) return act == QN_ACT_TANH ? k_fused_bwd_f64<HH, NN, 4, false> : k_fused_bwd_f64<HH, NN, 4, true>;
    QN_PICK(16, 1) QN_PICK(16, 2) QN_PICK(16, 3) QN_PICK(16, 4)
    QN_PICK(32, 1) QN_PICK(32, 2) QN_PICK(32, 3) QN_PICK(32, 4)
    QN_PICK(64, 1) QN_PICK(64, 2) QN_PICK(64, 3)
#undef QN_PICK
    return nullptr;
}

// raise the dynamic-LDS limit of a kernel once per process (not per launch: keeps the launch
// function free of non-stream API calls, so it can be captured into a HIP graph)
int arm_lds(const void* fn) {
    static std::mutex mu;
    static std::unordered_set<uint64_t> armed;              // (kernel, device): the attribute is per device
    int dev = 0;
    QN_HIP_CHECK(hipGetDevice(&dev));
    const uint64_t key = (uint64_t)(uintptr_t)fn * 64u + (uint64_t)(dev & 63);
    std::lock_guard<std::mutex> lock(mu);
    if (armed.count(key)) return QN_OK;
    QN_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    armed.insert(key);
    return QN_OK;
}

}  // namespace

bool qn_fused_uses_i8(const qn_desc* d, int want_grad) { return want_grad ? uses_i8_bwd(d) : uses_i8(d, 0); }

bool qn_fused_supported(const qn_desc* d, int B, int Nb, int want_grad, int dtype) {
    int H, nhid;
    if (dtype != QN_F64) return false;
    if (!uniform_hidden(d, &H, &nhid)) return false;
#ifdef QN_NO_STREAM
    if (H == HS) return false;                       // A/B builds: layer-wise path at H = 128
#endif
    if (H == HS && want_grad) return false;          // the streaming kernel is forward-only
    if (H != 16 && H != 32 && H != 64 && H != HS) return false;
    const int din = d->dims[0], dout = d->dims[d->nlayers];
    if (din > DMAX || dout > OMAX) {
        // wide first / last layer, hidden width <= 64: up to 16 inputs and 16 outputs, forward and gradient
        if (H == HS || din > DWIDE || dout > OWIDE) return false;
    }
    if (want_grad && !pick_bwd(H, nhid, d->act, bwd_dp(din, dout), bwd_om(dout))) return false;
    if (!want_grad && !pick_fwd(H, d->act, padded_d(din), dout)) return false;
    return lds_need(H, d->dims[0], d->dims[d->nlayers], nhid, want_grad) <= 160 * 1024;
}

size_t qn_fused_workspace(const qn_desc* d, int B, int Nb, int want_grad, int dtype) {
    FusedArgs a;
    plan(d, B, Nb, want_grad, &a);
    size_t tot = qn_align((size_t)B * a.nsplit * sizeof(double)) + qn_align((size_t)B * sizeof(unsigned long long));    // partials | arrivals
    if (want_grad) tot += qn_align((size_t)B * a.nsplit * d->p * sizeof(double)) + qn_align((size_t)B * a.nsplit * sizeof(int));
    return tot + 1024;              // (+ a scratch area diagnostic builds write their stamps to)
}

int qn_fused_parts(const qn_desc* d, int B, int Nb) {
    FusedArgs a;
    plan(d, B, Nb, 0, &a);
    return a.nsplit;
}

int qn_fused_run(const qn_desc* d, int dtype, const void* W, const void* X, const void* Y, const int32_t* row_idx,
                 int B, int N, int Nb, double* sse, void* pred, void* gradW, void* ws, size_t ws_bytes,
                 hipStream_t st, bool parts_out) {
    int H, nhid;
    const int want_grad = gradW != nullptr;
    if (!uniform_hidden(d, &H, &nhid) || dtype != QN_F64) {
        qn_set_error("qn_fused_run: unsupported configuration");
        return QN_EUNSUPPORTED;
    }
    FusedArgs a;
    a.p = d->p; a.B = B; a.N = N; a.Nb = Nb; a.d = d->dims[0]; a.o = d->dims[d->nlayers]; a.nhid = nhid;
    a.act = d->act; a.has_bias = d->has_bias;
    plan(d, B, Nb, want_grad, &a);
    const size_t npart = qn_align((size_t)B * a.nsplit * sizeof(double)) + qn_align((size_t)B * sizeof(unsigned long long));
    const size_t nslab = want_grad ? qn_align((size_t)B * a.nsplit * d->p * sizeof(double)) : 0;
    const size_t need = npart + nslab + (want_grad ? qn_align((size_t)B * a.nsplit * sizeof(int)) : 0);
    if (need > ws_bytes) {
        qn_set_error("workspace too small: need %zu bytes, got %zu", need, ws_bytes);
        return QN_EWORKSPACE;
    }
    double* partial = (parts_out && !want_grad) ? sse : static_cast<double*>(ws);
    double* slab = reinterpret_cast<double*>(static_cast<char*>(ws) + npart);
    a.dbg_off = (int64_t)(need / sizeof(double));          // the 256 spare bytes behind the slabs
    size_t lds_bytes = lds_need(H, a.d, a.o, nhid, want_grad);
    dim3 grid(qn_fused_grid(a.nsplit, B));
    bool summed = false;
    (void)hipGetLastError();
    if (!want_grad) {
        fwd_fn kern = pick_fwd(H, a.act, padded_d(a.d), a.o);
        if (uses_i8(d, want_grad)) {
            kern = qn_fused_i8_kernel(a.d, a.o, a.act);
            lds_bytes = qn_fused_i8_lds_bytes(a.d, nhid);
#ifdef QN_DEBUG_LDS_PAD
            if (const char* pad = getenv("QN_DEBUG_LDS_PAD")) lds_bytes += (size_t)atoi(pad);     // (occupancy experiments)
#endif
        }
        if (!kern) {
            qn_set_error("qn_fused_run: no forward kernel instance for H=%d act=%d", H, a.act);
            return QN_EUNSUPPORTED;
        }
        if (int rc = arm_lds(reinterpret_cast<const void*>(kern))) return rc;
        // the chain's SSE is summed by the last of its workgroups to finish (qn_sse_finish), unless the caller wants the parts
        unsigned long long* arrive = partial == sse ? nullptr :
            reinterpret_cast<unsigned long long*>(static_cast<char*>(ws) + qn_align((size_t)B * a.nsplit * sizeof(double)));
#ifdef QN_NO_ARRIVE
        arrive = nullptr;                                   // A/B builds: the separate k_sum_partials launch
#endif
        summed = arrive != nullptr;
        hipLaunchKernelGGL(kern, grid, dim3(H == HS ? NTS : WG), lds_bytes, st, a, (const double*)W,
                           (const double*)X, (const double*)Y, row_idx, (double*)pred, partial, arrive, sse);
    } else {
        bwd_fn kern = pick_bwd(H, nhid, d->act, bwd_dp(a.d, a.o), bwd_om(a.o));
        if (!kern) {
            qn_set_error("qn_fused_run: no backward kernel instance for H=%d nhid=%d", H, nhid);
            return QN_EUNSUPPORTED;
        }
        if (int rc = arm_lds(reinterpret_cast<const void*>(kern))) return rc;
        const int* flagged = nullptr;
        if (uses_i8_bwd(d)) {
            // sliced int8 products (qn_fused_bwd_i8.hip); (chain, split)s outside its contract are flagged and their chains
            // recomputed by the float64 kernel right behind it (which returns at once for every other chain)
            int* flags = reinterpret_cast<int*>(static_cast<char*>(ws) + npart + nslab);
            qn_bwd_i8_fn k8 = qn_fused_bwd_i8_kernel(nhid, a.d, a.act);
            if (int rc = arm_lds(reinterpret_cast<const void*>(k8))) return rc;
            hipLaunchKernelGGL(k8, grid, dim3(WG), qn_fused_bwd_i8_lds_bytes(nhid, a.d), st, a, (const double*)W, (const double*)X,
                               (const double*)Y, row_idx, (double*)pred, partial, slab, flags);
            flagged = flags;
        }
        // behind k_fused_bwd_i8 the float64 kernel's launch is the flagged chains' recomputation AND the slab reduction of all the
        // others (one launch instead of two: 221 -> 216 us per cfg2 gradient step); alone it is followed by k_grad_reduce
#ifdef QN_BWD_SEPARATE_REDUCE
        const bool merged = false;                                   // A/B builds
#else
        const bool merged = flagged != nullptr;
#endif
        unsigned long long* arrive =
            reinterpret_cast<unsigned long long*>(static_cast<char*>(ws) + qn_align((size_t)B * a.nsplit * sizeof(double)));
        hipLaunchKernelGGL(kern, grid, dim3(WG), lds_bytes, st, a, (const double*)W, (const double*)X,
                           (const double*)Y, row_idx, (double*)pred, partial, slab, flagged, merged ? (double*)gradW : (double*)nullptr,
                           sse, arrive);
        if (!merged) {
            int gx = (int)((d->p + 255) / 256);
            if (gx > 64) gx = 64;
            hipLaunchKernelGGL(k_grad_reduce, dim3(gx, B), dim3(256), 0, st, slab, a.nsplit, d->p, B, (double*)gradW,
                               (const double*)partial, sse);
        }
    }
    if (!want_grad && partial != sse && !summed) hipLaunchKernelGGL(k_sum_partials, dim3((B + 63) / 64), dim3(64), 0, st, partial, a.nsplit, B, sse);
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}
#endif  // QN_FUSED_PART == 0
