// Fused float64 forward for 64-wide tanh networks with the hidden GEMMs as SLICED EXACT PRODUCTS on the int8 matrix
// pipe (v_mfma_i32_16x16x64_i8) -- the headline kernel of BASELINE configs[1] (64 chains, 3x64, N = 4096).
//
// Why.  On gfx950 the float64 MFMA shares the vector ALUs: k_fused_fwd_f64 pays (MFMA cycles) + (tanh cycles) and
// sits at ~0.5 of the f64 MFMA peak with ~0.57 as its ceiling.  The int8 matrix pipe is a separate
// unit: tools/ubench_i8.hip shows a wave of v_mfma_i32_16x16x64_i8 (16.2 cycles each) running at full rate beside a
// float64 VALU wave on the same SIMD, which keeps 55-75 % of its solo rate; inside one wave an i8 MFMA costs ~5 issue
// cycles in a VALU-bound stream.  A 64x64 float64 layer for 16 data rows is 64 f64 MFMAs = 4096 cycles on the vector
// pipe; as sliced int8 products it is 4 x 26 MFMAs = 1700 cycles on the OTHER pipe, hidden behind the activation.
//
// Arithmetic (an Ozaki-style error-free product, cf. the reference's plain float64 addmm, quinn/nns/mlp.py:92-101).
//   activations a in [-1, 1] (tanh):  m_a = round(a 2^46)           = sum_k da_k 256^k, k = 0..5
//   weights, per output row j:        m_w = round(w 2^(46 - e_j))   = sum_k dw_k 256^k, 2^e_j > max_i |W_ji|
//   digits balanced: d_k in [-128, 127] for k < 5 (bytes of m + 0x8080808080, xor 0x80), top digit in [-64, 64];
//   one fma with the constant 1.5 2^52 + 0x8080808080 leaves all six digits in the low mantissa bytes.
//   z_j = b_j + 2^(e_j - 92) sum_i m_w(j,i) m_a(i):  every digit product dw_p da_q is an exact int8 x int8 -> int32 MFMA
//   over the whole K = 64; products of equal level L = p + q share an accumulator (<= 6 x 64 x 2^14 < 2^23); levels
//   L >= LMIN = 4 are kept (26 of 36 products; the dropped ones are < 2^-51 of the row scale), recombined exactly in
//   pairs (int32) and then in float64.  Error of a pre-activation: ~1e-14 relative to |W_j| (quantisation of a and w to
//   2^-47), against ~2e-16 for the f64 MFMA chain -- both far inside the 1e-11 parity tolerance (tests).
//
// Layout.  Z^T[feature x row] = W A^T per 16-row group: A operand = weight digits from LDS (lane: row = lane & 15,
// 16 bytes = k-slots of lane group lane >> 4), B operand = activation digits (lane: column = data row = lane & 15, same
// k-slots), C/D: lane (q, c) holds features 16 t + 4 q + r (r = 0..3) of row c.  The k-slot <-> feature map is free
// as long as both operands use it: byte j of lane group q is feature 16 (j >> 2) + 4 q + (j & 3), which makes the four
// results of output tile t exactly bytes 4 t .. 4 t + 3 of the NEXT layer's B operand: activations go
// accumulator -> float64 (tanh) -> digits -> operand without leaving the lane.  First layer (d <= 4) and last layer
// (o <= 4) on the VALU as in k_fused_fwd_f64; the last hidden layer's outputs are consumed as float64 (no slicing).
//
// Anything that could make a NaN / inf (a weight or input that is not finite and < 2^500) leaves the fast path: the
// wave evaluates its 32 rows with a plain float64 loop from the original weights (slow_tile), IEEE semantics as the
// layer-wise kernels.
#include "qn_common.h"
#include "qn_fused_args.h"
#include "qn_math.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int H = 64, T = 4, NS = 6, QB = 46, WGT = 256, G = 2;
constexpr int OMAX = 4;
constexpr int SLICE_BYTES = H * H;                  // one digit plane of a layer: [64 rows][64 bytes]
constexpr int LAYER_BYTES = NS * SLICE_BYTES;       // 24 KB
constexpr double kMagic = 6755399441055744.0 + 551911719040.0;     // 1.5 * 2^52 + 0x8080808080 (exact)
constexpr int TANH_TAB = QN_TANH_LDS_DOUBLES;

// LDS image, doubles first: W0 [64][DP] | b0 [64] | Wl [4][64] | bl [4] | red [8] | sb (NH-1) x [64][2] {scale, bias} |
// tanh table | slow-path scratch 4 x 128 | then bytes: (NH-1) x 6 digit planes
__host__ __device__ constexpr int thin_doubles(int dp) { return H * dp + H + OMAX * H + OMAX + 8; }
__host__ __device__ constexpr int head_doubles(int dp, int nhid) {
    return ((thin_doubles(dp) + (nhid - 1) * 2 * H + 1) & ~1) + ((TANH_TAB + 1) & ~1) + 4 * 128;
}

// slot swizzle of the digit planes: 16-byte slot s of row r is stored at slot s ^ hs(r >> 2 & 3); with it the four
// 16-lane groups of a ds_read_b128 each cover all 64 banks (rows r and r + 4 would otherwise collide 2-way)
__device__ __forceinline__ int slot_swz(int row) { return (0x1320 >> (4 * ((row >> 2) & 3))) & 3; }      // {0, 2, 3, 1}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ bool block_or(int mine, double* slot) {
    int* flag = reinterpret_cast<int*>(slot);
    if (threadIdx.x == 0) *flag = 0;
    __syncthreads();
    if (mine) *flag = 1;
    __syncthreads();
    return *flag != 0;
}

// four float64 values in [-1, 1] -> six words, word k = digit k of the four values in bytes 0..3
__device__ __forceinline__ void slice4(const double (&a)[4], int (&S)[NS]) {
    int lo[4], hi[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double x = fma(a[r], 0x1p46, kMagic);
        lo[r] = __double2loint(x);
        hi[r] = __double2hiint(x);
    }
    const int p01 = __builtin_amdgcn_perm(lo[1], lo[0], 0x05010400), q01 = __builtin_amdgcn_perm(lo[1], lo[0], 0x07030602);
    const int p23 = __builtin_amdgcn_perm(lo[3], lo[2], 0x05010400), q23 = __builtin_amdgcn_perm(lo[3], lo[2], 0x07030602);
    const int r01 = __builtin_amdgcn_perm(hi[1], hi[0], 0x05010400), r23 = __builtin_amdgcn_perm(hi[3], hi[2], 0x05010400);
    S[0] = __builtin_amdgcn_perm(p23, p01, 0x05040100) ^ 0x80808080;
    S[1] = __builtin_amdgcn_perm(p23, p01, 0x07060302) ^ 0x80808080;
    S[2] = __builtin_amdgcn_perm(q23, q01, 0x05040100) ^ 0x80808080;
    S[3] = __builtin_amdgcn_perm(q23, q01, 0x07060302) ^ 0x80808080;
    S[4] = __builtin_amdgcn_perm(r23, r01, 0x05040100) ^ 0x80808080;
    S[5] = __builtin_amdgcn_perm(r23, r01, 0x07060302);                 // top digit: two's complement as it stands
}

// Stage chain `Wb`: thin layers as float64, hidden matrices as digit planes.  Returns whether this thread saw a weight
// that is not finite and < 2^500.
template <int DP, int LMIN>
__device__ __forceinline__ int stage(double* __restrict__ lds, unsigned char* __restrict__ wq,
                                     const double* __restrict__ Wb, const FusedArgs& a) {
    int bad = 0;
    auto chk = [&](double v) { bad |= !qn_bounded(v); return v; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = a.d, o = a.o, nb = a.has_bias ? 1 : 0;
    const int64_t gb0 = (int64_t)H * d, gHH = gb0 + nb * H;
    const int64_t gWl = gHH + (int64_t)(a.nhid - 1) * (H * H + nb * H), gbl = gWl + (int64_t)o * H;
    const int lb0 = H * DP, lWl = lb0 + H, lbl = lWl + OMAX * H, lsb = thin_doubles(DP);
    for (int e = tid; e < H * DP; e += WGT) {
        const int j = e / DP, k = e % DP;
        lds[e] = k < d ? chk(Wb[j * d + k]) : 0.0;
    }
    for (int e = tid; e < H; e += WGT) lds[lb0 + e] = nb ? chk(Wb[gb0 + e]) : 0.0;
    for (int e = tid; e < OMAX * H; e += WGT) lds[lWl + e] = e < o * H ? chk(Wb[gWl + e]) : 0.0;
    for (int e = tid; e < OMAX; e += WGT) lds[lbl + e] = (nb && e < o) ? chk(Wb[gbl + e]) : 0.0;
    // hidden matrices: wave w takes rows w, w + 4, ...; a lane is one column, so a row's maximum is a wave reduction
    const int gq = (lane >> 2) & 3, jb = 4 * (lane >> 4) + (lane & 3);           // k-slot of input feature `lane`
    for (int layer = 1; layer < a.nhid; ++layer) {
        const double* Wg = Wb + gHH + (int64_t)(layer - 1) * (H * H + nb * H);
        unsigned char* plane = wq + (layer - 1) * LAYER_BYTES;
        double* sb = lds + lsb + (layer - 1) * 2 * H;
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = chk(Wg[(wave + 4 * u) * H + lane]);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int row = wave + 4 * u;
            const double mx = wave_max(fabs(v[u]));
            int e;
            (void)frexp(mx, &e);                                       // mx < 2^e (e = 0 for an all-zero row)
            e = e < -900 ? -900 : e;
            const double x = fma(v[u], ldexp(1.0, QB - e), kMagic);
            const int lo = __double2loint(x) ^ 0x80808080, hi = __double2hiint(x) ^ 0x80;
            unsigned char* dst = plane + row * H + 16 * (gq ^ slot_swz(row)) + jb;
            dst[0 * SLICE_BYTES] = (unsigned char)lo;
            dst[1 * SLICE_BYTES] = (unsigned char)(lo >> 8);
            dst[2 * SLICE_BYTES] = (unsigned char)(lo >> 16);
            dst[3 * SLICE_BYTES] = (unsigned char)(lo >> 24);
            dst[4 * SLICE_BYTES] = (unsigned char)hi;
            dst[5 * SLICE_BYTES] = (unsigned char)(hi >> 8);
            if (lane == 0) {
                sb[2 * row] = ldexp(1.0, e - 2 * QB + 8 * LMIN);       // integer sum (in units of 256^LMIN) -> W_j . a
                sb[2 * row + 1] = nb ? chk(Wg[H * H + row]) : 0.0;
            }
        }
    }
    return bad;
}

// The wave's 32 rows in plain float64 from the ORIGINAL weights (rare: non-finite / huge weights or inputs).  Lane j
// owns feature j of one row at a time; the previous layer's activations are broadcast through `scr` (128 doubles).
template <int DP>
__device__ __noinline__ double slow_tile(const FusedArgs& a, const double* __restrict__ lds, double* __restrict__ scr,
                                         const double* __restrict__ Wb, const double* __restrict__ X,
                                         const double* __restrict__ Y, const int32_t* __restrict__ row_idx, int nbase, int b,
                                         double* __restrict__ pred_out) {
    const int lane = threadIdx.x & 63;
    const int nb = a.has_bias ? 1 : 0, o = a.o, d = a.d;
    const int64_t gHH = (int64_t)H * d + nb * H;
    const int lb0 = H * DP, lWl = lb0 + H, lbl = lWl + OMAX * H;
    double sse = 0.0;
    for (int n = nbase; n < nbase + 16 * G && n < a.Nb; ++n) {
        const int64_t rr = row_idx ? (int64_t)row_idx[(int64_t)b * a.Nb + n] : (int64_t)n;
        double z = lds[lb0 + lane];
        for (int k = 0; k < d; ++k) z = fma(lds[lane * DP + k], X[rr * d + k], z);
        double act = qn_tanh_f64(z);
        for (int layer = 1; layer < a.nhid; ++layer) {
            double* cur = scr + 64 * (layer & 1);
            cur[lane] = act;
            __builtin_amdgcn_s_waitcnt(0xC07F);                     // lgkmcnt(0): the wave's own LDS writes have landed
            const double* Wg = Wb + gHH + (int64_t)(layer - 1) * (H * H + nb * H);
            z = nb ? Wg[H * H + lane] : 0.0;
            for (int i = 0; i < H; ++i) z = fma(Wg[lane * H + i], cur[i], z);
            act = qn_tanh_f64(z);
        }
        for (int qo = 0; qo < o; ++qo) {
            const double pr = wave_sum(lds[lWl + qo * H + lane] * act) + lds[lbl + qo];      // lane 0 holds the sum
            const double res = pr - Y[rr * o + qo];
            if (lane == 0) {
                sse += res * res;
                if (pred_out) pred_out[((int64_t)b * a.Nb + n) * o + qo] = pr;
            }
        }
    }
    return sse;
}

template <int DP, int LMIN, int OM>
__global__ __launch_bounds__(WGT, 2) void k_fused_fwd_i8(FusedArgs a, const double* __restrict__ W,
                                                        const double* __restrict__ X, const double* __restrict__ Y,
                                                        const int32_t* __restrict__ row_idx, double* __restrict__ pred_out,
                                                        double* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NLEV = 2 * (NS - 1) - LMIN + 1;       // levels LMIN .. 10
    double* lds = reinterpret_cast<double*>(smem);
    const int NH = a.nhid, d = a.d, o = a.o;
    const int b = blockIdx.y, split = blockIdx.x;
    const int offb0 = H * DP, offWl = offb0 + H, offbl = offWl + OMAX * H, offred = offbl + OMAX, offsb = thin_doubles(DP);
    double* tanh_tab = lds + ((offsb + (NH - 1) * 2 * H + 1) & ~1);
    double* scratch = tanh_tab + ((TANH_TAB + 1) & ~1);
    unsigned char* wq = reinterpret_cast<unsigned char*>(lds + head_doubles(DP, NH));
    double* red = lds + offred;
    const double* Wb = W + (int64_t)b * a.p;
    qn_tanh_table_stage(tanh_tab, threadIdx.x, WGT);
    const bool w_bad = block_or(stage<DP, LMIN>(lds, wq, Wb, a), red + 6);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, c = lane & 15;
    const int lofs = c * H + 16 * (q ^ slot_swz(c));          // this lane's 16 bytes inside a 16-row tile of a digit plane
    double sse = 0.0;

    double xn[G][DP], yn[G][OM];
    int nrow_n[G];
    bool valid_n[G];
    int xbad_n = 0;
    auto fetch = [&](int it) {
        xbad_n = 0;
        const int nbase = split * a.rows_per_split + (it * (WGT / 64) + wave) * 16 * G;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int n = nbase + 16 * g + c;
            valid_n[g] = n < a.Nb;
            nrow_n[g] = n;
            const int nn = valid_n[g] ? n : 0;
            const int64_t rr = row_idx ? (int64_t)row_idx[(int64_t)b * a.Nb + nn] : (int64_t)nn;
#pragma unroll
            for (int k = 0; k < DP; ++k) {
                xn[g][k] = k < d ? X[rr * d + k] : 0.0;
                xbad_n |= !qn_bounded(xn[g][k]);
            }
#pragma unroll
            for (int qo = 0; qo < OM; ++qo) yn[g][qo] = qo < o ? Y[rr * o + qo] : 0.0;
        }
    };
    fetch(0);
    for (int it = 0; it < a.iters; ++it) {
        double xk[G][DP], yk[G][OM];
        int nrow[G];
        bool valid[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            valid[g] = valid_n[g];
            nrow[g] = nrow_n[g];
#pragma unroll
            for (int k = 0; k < DP; ++k) xk[g][k] = xn[g][k];
#pragma unroll
            for (int qo = 0; qo < OM; ++qo) yk[g][qo] = yn[g][qo];
        }
        const bool exceptional = w_bad || __any(xbad_n);
        if (it + 1 < a.iters) fetch(it + 1);
        if (exceptional) {                                              // wave-uniform
            sse += slow_tile<DP>(a, lds, scratch + 128 * wave, Wb, X, Y, row_idx,
                                 split * a.rows_per_split + (it * (WGT / 64) + wave) * 16 * G, b, pred_out);
            continue;
        }
        // ---- first layer (VALU): a_1 = tanh(W0 x + b0), sliced straight into the B operand
        v4i Bc[G][NS];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int t = 0; t < T; ++t) {
                double av[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * t + 4 * q + r;
                    double z = lds[offb0 + j];
#pragma unroll
                    for (int k = 0; k < DP; ++k) z = fma(lds[j * DP + k], xk[g][k], z);
                    av[r] = qn_tanh_f64_tab<false>(z, tanh_tab);
                }
                int S[NS];
                slice4(av, S);
#pragma unroll
                for (int k = 0; k < NS; ++k) Bc[g][k][t] = S[k];
            }
        // ---- hidden -> hidden layers: digit products on the int8 matrix pipe
        double part[G][OM];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int qo = 0; qo < OM; ++qo) part[g][qo] = 0.0;
        for (int layer = 1; layer < NH; ++layer) {
            const unsigned char* plane = wq + (layer - 1) * LAYER_BYTES + lofs;
            const double* sb = lds + offsb + (layer - 1) * 2 * H;
            const bool last = layer == NH - 1;
            // one 16-row group at a time (its 7 level accumulators, the weight digits of a tile and the operand being
            // built stay within the register budget); the other group's MFMAs / the other wave of the SIMD fill the pipe
#pragma unroll
            for (int g = 0; g < G; ++g) {
                v4i Bn[NS];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    v4i acc[NLEV];
#pragma unroll
                    for (int l = 0; l < NLEV; ++l) acc[l] = (v4i){0, 0, 0, 0};
#pragma unroll
                    for (int wi = 0; wi < NS; ++wi) {
                        const v4i Af = *reinterpret_cast<const v4i*>(plane + wi * SLICE_BYTES + t * 16 * H);
#pragma unroll
                        for (int aj = 0; aj < NS; ++aj) {
                            if (wi + aj < LMIN) continue;
                            acc[wi + aj - LMIN] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Af, Bc[g][aj], acc[wi + aj - LMIN], 0, 0, 0);
                        }
                    }
                    // epilogue of the tile: recombine the levels, scale + bias, tanh, then digits (or the last layer's dot)
                    double av[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // pairs of levels in int32 (bounds in the header), then float64: exact up to the last 2^16 step
                        double tsum;
                        if constexpr ((NLEV & 1) == 1) {
                            tsum = (double)acc[NLEV - 1][r];
#pragma unroll
                            for (int l = NLEV - 3; l >= 0; l -= 2)
                                tsum = fma(tsum, 65536.0, (double)(acc[l][r] + (acc[l + 1][r] << 8)));
                        } else {
                            tsum = (double)(acc[NLEV - 2][r] + (acc[NLEV - 1][r] << 8));
#pragma unroll
                            for (int l = NLEV - 4; l >= 0; l -= 2)
                                tsum = fma(tsum, 65536.0, (double)(acc[l][r] + (acc[l + 1][r] << 8)));
                        }
                        const int j = 16 * t + 4 * q + r;
                        const double2 sc = *reinterpret_cast<const double2*>(sb + 2 * j);
                        av[r] = qn_tanh_f64_tab<false>(fma(tsum, sc.x, sc.y), tanh_tab);
                    }
                    if (!last) {
                        int S[NS];
                        slice4(av, S);
#pragma unroll
                        for (int k = 0; k < NS; ++k) Bn[k][t] = S[k];
                    } else {
#pragma unroll
                        for (int qo = 0; qo < OM; ++qo)
                            if (qo < o) {
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    part[g][qo] = fma(lds[offWl + qo * H + 16 * t + 4 * q + r], av[r], part[g][qo]);
                            }
                    }
                }
                if (!last) {
#pragma unroll
                    for (int k = 0; k < NS; ++k) Bc[g][k] = Bn[k];
                }
            }
        }
        // ---- last layer: finish the dot over the four lane groups, residual, SSE
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int qo = 0; qo < OM; ++qo) {
                if (qo >= o) break;
                double p = part[g][qo];
                p += __shfl_xor(p, 16, 64);
                p += __shfl_xor(p, 32, 64);
                const double pr = p + lds[offbl + qo];
                const double res = pr - yk[g][qo];
                if (valid[g] && q == 0) {
                    sse += res * res;
                    if (pred_out) pred_out[((int64_t)b * a.Nb + nrow[g]) * o + qo] = pr;
                }
            }
    }
    sse = wave_sum(sse);
    if (lane == 0) red[wave] = sse;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)b * a.nsplit + split] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace

// ---- what qn_fused.hip needs to dispatch to this kernel
bool qn_fused_i8_applies(int Hh, int nhid, int act, int d, int o) {
    if (Hh != H || act != QN_ACT_TANH || nhid < 2 || d > 4 || o > OMAX) return false;
    return qn_fused_i8_lds_bytes(d, nhid) <= 160 * 1024;
}
size_t qn_fused_i8_lds_bytes(int d, int nhid) {
    const int dp = d <= 2 ? 2 : 4;
    return sizeof(double) * (size_t)head_doubles(dp, nhid) + (size_t)(nhid - 1) * LAYER_BYTES;
}
qn_fwd_fn qn_fused_i8_kernel(int d, int o) {
#ifndef QN_I8_LMIN
#define QN_I8_LMIN 4
#endif
    if (o == 1) return d <= 2 ? k_fused_fwd_i8<2, QN_I8_LMIN, 1> : k_fused_fwd_i8<4, QN_I8_LMIN, 1>;
    return d <= 2 ? k_fused_fwd_i8<2, QN_I8_LMIN, OMAX> : k_fused_fwd_i8<4, QN_I8_LMIN, OMAX>;
}
