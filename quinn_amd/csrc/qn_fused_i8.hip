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
#include <type_traits>
#include <utility>

#ifndef QN_I8_G
#define QN_I8_G 1              // 16-row groups per wave iteration (register budget of the pipelined epilogue: one)
#endif
#ifndef QN_I8_VPM
#define QN_I8_VPM 7            // vector instructions scheduled behind each MFMA of the pipelined epilogue
#endif

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int H = 64, T = 4, NS = 6, QB = 46, WGT = 256, G = QN_I8_G;
constexpr int OMAX = 4;
constexpr int SLICE_BYTES = H * H;                  // one digit plane of a layer: [64 rows][64 bytes]
constexpr int LAYER_BYTES = NS * SLICE_BYTES;       // 24 KB
constexpr double kMagic = 6755399441055744.0 + 551911719040.0;     // 1.5 * 2^52 + 0x8080808080 (exact)
constexpr int TANH_TAB = QN_TANH64_LDS_DOUBLES;       // tanh(n / 64): the absolute-accuracy activation (qn_math.h)

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
// maximum of an unsigned integer over each row of 16 lanes (DPP, no LDS traffic); every lane of the row gets it
__device__ __forceinline__ unsigned row16_max_u32(unsigned x) {
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false));      // quad_perm [1,0,3,2]
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, false));      // quad_perm [2,3,0,1]
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, false));     // row_half_mirror
    x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, false));     // row_mirror
    return x;
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

// the kept digit products of a tile in issue order (weight digit major): product k is (wi, aj) with wi + aj >= LMIN
__host__ __device__ constexpr int nprod(int lmin) {
    int n = 0;
    for (int wi = 0; wi < NS; ++wi)
        for (int aj = 0; aj < NS; ++aj) n += (wi + aj >= lmin) ? 1 : 0;
    return n;
}
__host__ __device__ constexpr int prod_wi(int lmin, int k) {
    int n = 0;
    for (int wi = 0; wi < NS; ++wi)
        for (int aj = 0; aj < NS; ++aj)
            if (wi + aj >= lmin) { if (n == k) return wi; ++n; }
    return 0;
}
__host__ __device__ constexpr int prod_aj(int lmin, int k) {
    int n = 0;
    for (int wi = 0; wi < NS; ++wi)
        for (int aj = 0; aj < NS; ++aj)
            if (wi + aj >= lmin) { if (n == k) return aj; ++n; }
    return 0;
}
// cumulative number of the next tile's products issued up to and including epilogue stage `st` (20 stages): weights ~
// the stage's vector cycles / 16 (recombination 45, clamp 36, table 40, ..., v_rcp_f64 64, Newton 18 each, digits 12)
// (plain conditional arithmetic: it has to fold while the stage loop is unrolled, or every register index turns dynamic)
__host__ __device__ constexpr int stage_quota(int st, int np) {
    const int c = st < 2 ? 0 : st == 2 ? 3 : st == 3 ? 6 : st == 4 ? 8 : st == 5 ? 10 : st == 6 ? 11 : st == 7 ? 13 : st == 8 ? 14 :
                  st == 9 ? 16 : st == 10 ? 19 : st == 11 ? 20 : st == 12 ? 21 : st == 13 ? 22 : st == 14 ? 23 : st == 15 ? 24 :
                  st == 16 ? 25 : 26;
    const int v = (c * np + 25) / 26;                             // fewer products (LMIN = 5): same shape, scaled
    return st >= 17 ? np : (v > np ? np : v);
}
// is product k the first one of its level (then the accumulator input is the constant 0)
__host__ __device__ constexpr bool prod_first(int lmin, int k) {
    const int l = prod_wi(lmin, k) + prod_aj(lmin, k);
    for (int j = 0; j < k; ++j)
        if (prod_wi(lmin, j) + prod_aj(lmin, j) == l) return false;
    return true;
}
template <class F, int... I>
__device__ __forceinline__ void for_each_stage(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int LMIN, int NLEV>
__device__ __forceinline__ void issue_product(int k, v4i (&acc)[NLEV], const v4i (&Af)[NS], const v4i (&B)[NS]) {
    // (k is a compile-time constant after unrolling; the switch makes the register indices static)
#define QN_PRODUCT(KK)                                                                                               \
    case KK:                                                                                                         \
        if constexpr (KK < nprod(LMIN)) {                                                                            \
            constexpr int wi = prod_wi(LMIN, KK), aj = prod_aj(LMIN, KK), l = wi + aj - LMIN;                          \
            acc[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Af[wi], B[aj], prod_first(LMIN, KK) ? (v4i){0, 0, 0, 0} : acc[l], 0, 0, 0); \
        }                                                                                                            \
        break;
    switch (k) {
        QN_PRODUCT(0) QN_PRODUCT(1) QN_PRODUCT(2) QN_PRODUCT(3) QN_PRODUCT(4) QN_PRODUCT(5) QN_PRODUCT(6) QN_PRODUCT(7)
        QN_PRODUCT(8) QN_PRODUCT(9) QN_PRODUCT(10) QN_PRODUCT(11) QN_PRODUCT(12) QN_PRODUCT(13) QN_PRODUCT(14) QN_PRODUCT(15)
        QN_PRODUCT(16) QN_PRODUCT(17) QN_PRODUCT(18) QN_PRODUCT(19) QN_PRODUCT(20) QN_PRODUCT(21) QN_PRODUCT(22) QN_PRODUCT(23)
        QN_PRODUCT(24) QN_PRODUCT(25) QN_PRODUCT(26) QN_PRODUCT(27) QN_PRODUCT(28) QN_PRODUCT(29)
    default: break;
    }
#undef QN_PRODUCT
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
    // hidden matrices: a thread takes 4 consecutive input features of one row = the 4 bytes of one dword of every digit
    // plane (k-slot map of the header), the 16 lanes of a DPP row take one matrix row: the row's largest exponent is 4
    // DPP steps, the digits come out of slice4 already packed, 6 ds_write_b32 per item (byte stores of single digits
    // were 4-way bank conflicts and, with a ds_bpermute row maximum, 19 % of the kernel)
    const int q16 = lane & 15, m4 = q16 >> 2, g4 = q16 & 3;                // quad (m, g): features 16 m + 4 g + {0..3}
#ifdef QN_I8_SKIP_STAGE
    if (a.nhid > 100)             // timing-only build: what the digit staging costs
#endif
    for (int layer = 1; layer < a.nhid; ++layer) {
        const double* Wg = Wb + gHH + (int64_t)(layer - 1) * (H * H + nb * H);
        unsigned char* plane = wq + (layer - 1) * LAYER_BYTES;
        double* sb = lds + lsb + (layer - 1) * 2 * H;
        double v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = 16 * u + 4 * wave + (lane >> 4);
            const double2* src = reinterpret_cast<const double2*>(Wg + row * H + 16 * m4 + 4 * g4);
            const double2 v01 = src[0], v23 = src[1];
            v[u][0] = chk(v01.x); v[u][1] = chk(v01.y); v[u][2] = chk(v23.x); v[u][3] = chk(v23.y);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = 16 * u + 4 * wave + (lane >> 4);
            unsigned ex = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) ex = max(ex, ((unsigned)__double2hiint(v[u][r]) & 0x7fffffffu) >> 20);
            // 2^e > every |W_ji| of the row, from the largest biased exponent field E: |w| < 2^(E - 1022)
            int e = (int)row16_max_u32(ex) - 1022;
            e = e < -900 ? -900 : e;                                   // (all-zero / denormal rows: any scale will do)
            double an[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) an[r] = ldexp(v[u][r], -e);    // exact, |an| < 1
            int S[NS];
            slice4(an, S);
            unsigned char* dst = plane + row * H + 16 * (g4 ^ slot_swz(row)) + 4 * m4;
#pragma unroll
            for (int k = 0; k < NS; ++k) *reinterpret_cast<int*>(dst + k * SLICE_BYTES) = S[k];
            if (q16 == 0) {
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
// (the arguments it needs by value: a reference to the kernel's argument struct would force a copy of the struct into
// scratch memory at every kernel entry -- 8 MB of writes per launch in the first version)
__device__ __noinline__ double slow_tile(int Nb, int d, int o, int nhid, int has_bias, const double* __restrict__ lds,
                                         double* __restrict__ scr, const double* __restrict__ Wb,
                                         const double* __restrict__ X, const double* __restrict__ Y,
                                         const int32_t* __restrict__ row_idx, int nbase, int b,
                                         double* __restrict__ pred_out) {
    const int lane = threadIdx.x & 63;
    const int nb = has_bias ? 1 : 0;
    const int64_t gHH = (int64_t)H * d + nb * H;
    const int lb0 = H * DP, lWl = lb0 + H, lbl = lWl + OMAX * H;
    double sse = 0.0;
    for (int n = nbase; n < nbase + 16 * G && n < Nb; ++n) {
        const int64_t rr = row_idx ? (int64_t)row_idx[(int64_t)b * Nb + n] : (int64_t)n;
        double z = lds[lb0 + lane];
        for (int k = 0; k < d; ++k) z = fma(lds[lane * DP + k], X[rr * d + k], z);
        double act = qn_tanh_f64(z);
        for (int layer = 1; layer < nhid; ++layer) {
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
                if (pred_out) pred_out[((int64_t)b * Nb + n) * o + qo] = pr;
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
    qn_tanh_table64_stage(tanh_tab, threadIdx.x, WGT);
    const bool w_bad = block_or(stage<DP, LMIN>(lds, wq, Wb, a), red + 6);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, c = lane & 15;
#ifdef QN_I8_STAGGER
    // experiment: break the lockstep of the two waves of a SIMD (same program, same start) by delaying the odd wave slot
    if (__builtin_amdgcn_s_getreg(0x1804) & 1) __builtin_amdgcn_s_sleep(QN_I8_STAGGER);
#endif
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
            sse += slow_tile<DP>(a.Nb, a.d, a.o, a.nhid, a.has_bias, lds, scratch + 128 * wave, Wb, X, Y, row_idx,
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
                    av[r] = qn_tanh_f64_tab64(z, tanh_tab);
                }
                int S[NS];
                slice4(av, S);
#pragma unroll
                for (int k = 0; k < NS; ++k) Bc[g][k][t] = S[k];
            }
        // ---- hidden -> hidden layers: digit products on the int8 matrix pipe.
        // Software pipeline over the 8 (row group, output tile) items of a layer: the MFMAs of item i + 1 are issued
        // BETWEEN the vector instructions of item i's epilogue (recombine, tanh, digits).  An i8 MFMA costs a
        // VALU-bound wave ~5 issue cycles (tools/ubench_i8.hip) while its 16 cycles run on the other pipe; issued as a
        // burst ahead of the epilogue (first version of this kernel) the two phases simply added up: rocprofv3 showed
        // VALU busy 65 % + MFMA busy 24 % of the SIMD time, no overlap, also not between the two waves of a SIMD.
        double part[G][OM];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int qo = 0; qo < OM; ++qo) part[g][qo] = 0.0;
        constexpr int NPROD = nprod(LMIN);                              // digit products per tile
        // all of a tile's MFMAs back to back (the first tile of a layer: nothing to hide them behind)
        auto load_frags = [&](v4i (&Af)[NS], const unsigned char* tile) {
#pragma unroll
            for (int wi = 0; wi < NS; ++wi) Af[wi] = *reinterpret_cast<const v4i*>(tile + wi * SLICE_BYTES);
        };
        // Epilogue of one tile in STAGES of a few vector instructions for each of its 4 elements, with the MFMAs of the
        // NEXT tile dealt out between the stages (scheduling fences keep them there): recombine the levels, scale + bias,
        // tanh (qn_tanh_f64_tab64 written out), then digits (or the last layer's dot product).
        auto epilogue = [&](auto last_tag, auto next_tag, const v4i (&acc)[NLEV], v4i (&accn)[NLEV], const unsigned char* tile_next,
                            const v4i (&B)[NS], const double* sbt, const double* wlt, int (&S)[NS], double (&prt)[OM]) {
            constexpr bool LAST = decltype(last_tag)::value, NEXT = decltype(next_tag)::value;
            constexpr int NSTAGE = 20;
            v4i Af[NS];
            double2 sc[4];
            double ts[4], z[4], ax[4], zm[4], Tt[4], bb[4], b2[4], pp[4], tb[4], num[4], den[4], y0[4], e0[4], av[4];
            int lo[4], hi[4], p01, q01, p23, q23, r01, r23;
            auto stage = [&](auto st_tag) {
                constexpr int st = decltype(st_tag)::value;                // (a compile-time stage index: every register index below is static)
                if constexpr (NEXT) {
                    // the next tile's products, dealt out in proportion to the vector work of the stages (16 cycles each)
                    constexpr int from = st ? stage_quota(st - 1, NPROD) : 0, upto = stage_quota(st, NPROD);
#pragma unroll
                    for (int k = 0; k < NPROD; ++k)
                        if (k >= from && k < upto) issue_product<LMIN>(k, accn, Af, B);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    switch (st) {
                    case 0:
                        sc[r] = *reinterpret_cast<const double2*>(sbt + 2 * r);
                        ts[r] = (NLEV & 1) ? (double)acc[NLEV - 1][r] : (double)(acc[NLEV - 2][r] + (acc[NLEV - 1][r] << 8));
                        break;
                    case 1:
                        if constexpr (NLEV >= 5) { constexpr int l = ((NLEV & 1) ? NLEV - 3 : NLEV - 4); ts[r] = fma(ts[r], 65536.0, (double)(acc[l][r] + (acc[l + 1][r] << 8))); }
                        break;
                    case 2:
                        if constexpr (NLEV >= 5) { constexpr int l = ((NLEV & 1) ? NLEV - 5 : NLEV - 6); ts[r] = fma(ts[r], 65536.0, (double)(acc[l][r] + (acc[l + 1][r] << 8))); }
                        break;
                    case 3:
                        if constexpr (NLEV == 7) ts[r] = fma(ts[r], 65536.0, (double)(acc[0][r] + (acc[1][r] << 8)));
                        z[r] = fma(ts[r], sc[r].x, sc[r].y);
                        break;
                    case 4:
                        asm("v_min_f64 %0, |%1|, %2" : "=v"(ax[r]) : "v"(z[r]), "s"(20.0));
                        zm[r] = fma(ax[r], 64.0, 6755399441055744.0);
                        break;
                    case 5:
                        Tt[r] = tanh_tab[__double2loint(zm[r])];
                        bb[r] = fma(zm[r] - 6755399441055744.0, -0.015625, ax[r]);
                        break;
                    case 6:
                        b2[r] = bb[r] * bb[r];
                        break;
                    case 7:
                        pp[r] = fma(b2[r], 1.33333333333333333e-01, -3.33333333333333333e-01);
                        b2[r] = bb[r] * b2[r];
                        break;
                    case 8:
                        tb[r] = fma(b2[r], pp[r], bb[r]);
                        break;
                    case 9:
                        num[r] = Tt[r] + tb[r];
                        den[r] = fma(Tt[r], tb[r], 1.0);
                        break;
                    case 10:
                        y0[r] = __builtin_amdgcn_rcp(den[r]);
                        break;
                    case 11:
                        e0[r] = fma(-den[r], y0[r], 1.0);
                        break;
                    case 12:
                        e0[r] = fma(e0[r], e0[r], e0[r]);
                        break;
                    case 13:
                        y0[r] = fma(y0[r], e0[r], y0[r]);
                        break;
                    case 14:
                        av[r] = __builtin_copysign(num[r] * y0[r], z[r]);
                        break;
                    case 15:
                        if constexpr (LAST) {
#pragma unroll
                            for (int qo = 0; qo < OM; ++qo)
                                if (qo < o) prt[qo] = fma(wlt[qo * H + r], av[r], prt[qo]);
                        } else {
                            const double x = fma(av[r], 0x1p46, kMagic);
                            lo[r] = __double2loint(x);
                            hi[r] = __double2hiint(x);
                        }
                        break;
                    default: break;
                    }
                }
                if (st == 0 && NEXT) load_frags(Af, tile_next);          // (no product before stage 2: the reads are in flight)
                if constexpr (!LAST) {
                    if (st == 16) {
                        p01 = __builtin_amdgcn_perm(lo[1], lo[0], 0x05010400); q01 = __builtin_amdgcn_perm(lo[1], lo[0], 0x07030602);
                        p23 = __builtin_amdgcn_perm(lo[3], lo[2], 0x05010400); q23 = __builtin_amdgcn_perm(lo[3], lo[2], 0x07030602);
                    }
                    if (st == 17) {
                        r01 = __builtin_amdgcn_perm(hi[1], hi[0], 0x05010400); r23 = __builtin_amdgcn_perm(hi[3], hi[2], 0x05010400);
                        S[0] = __builtin_amdgcn_perm(p23, p01, 0x05040100) ^ 0x80808080;
                        S[1] = __builtin_amdgcn_perm(p23, p01, 0x07060302) ^ 0x80808080;
                    }
                    if (st == 18) {
                        S[2] = __builtin_amdgcn_perm(q23, q01, 0x05040100) ^ 0x80808080;
                        S[3] = __builtin_amdgcn_perm(q23, q01, 0x07060302) ^ 0x80808080;
                    }
                    if (st == 19) {
                        S[4] = __builtin_amdgcn_perm(r23, r01, 0x05040100) ^ 0x80808080;
                        S[5] = __builtin_amdgcn_perm(r23, r01, 0x07060302);
                    }
                }
#ifdef QN_I8_FENCE              // (scheduling fences between the stages: A/B-tested 1 % slower than the compiler's own order)
                __builtin_amdgcn_sched_barrier(0);
#endif
            };
            for_each_stage(stage, std::make_integer_sequence<int, NSTAGE>{});
        };
        auto hidden_layer = [&](auto last_tag, int layer) {
            constexpr bool LAST = decltype(last_tag)::value;
            const unsigned char* plane = wq + (layer - 1) * LAYER_BYTES + lofs;
            const double* sb = lds + offsb + (layer - 1) * 2 * H + 2 * 4 * q;      // this lane group's features 16 t + 4 q + r
            const double* wl = lds + offWl + 4 * q;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                v4i Bn[NS];
                v4i accA[NLEV], accB[NLEV];
                {
                    v4i Af0[NS];
                    load_frags(Af0, plane);
#pragma unroll
                    for (int k = 0; k < NPROD; ++k) issue_product<LMIN>(k, accA, Af0, Bc[g]);     // tile 0: nothing to hide it behind
                }
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    int S[NS];
                    const unsigned char* nxt = plane + (t + 1) * 16 * H;
                    if (t == T - 1) {
                        if (t & 1) epilogue(last_tag, std::false_type{}, accB, accA, nxt, Bc[g], sb + 32 * t, wl + 16 * t, S, part[g]);
                        else epilogue(last_tag, std::false_type{}, accA, accB, nxt, Bc[g], sb + 32 * t, wl + 16 * t, S, part[g]);
                    } else {
                        if (t & 1) epilogue(last_tag, std::true_type{}, accB, accA, nxt, Bc[g], sb + 32 * t, wl + 16 * t, S, part[g]);
                        else epilogue(last_tag, std::true_type{}, accA, accB, nxt, Bc[g], sb + 32 * t, wl + 16 * t, S, part[g]);
                    }
                    if constexpr (!LAST) {
#pragma unroll
                        for (int k = 0; k < NS; ++k) Bn[k][t] = S[k];
                    }
                }
                if constexpr (!LAST) {
#pragma unroll
                    for (int k = 0; k < NS; ++k) Bc[g][k] = Bn[k];
                }
            }
        };
        for (int layer = 1; layer < NH - 1; ++layer) hidden_layer(std::false_type{}, layer);
        hidden_layer(std::true_type{}, NH - 1);
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
int qn_fused_i8_rows_per_iteration() { return (WGT / 64) * 16 * G; }
bool qn_fused_i8_applies(int Hh, int nhid, int act, int d, int o) {
    // (several outputs: the staged epilogue with a 4-output dot spills ~170 registers; those networks keep the f64 kernel)
    if (Hh != H || act != QN_ACT_TANH || nhid < 2 || d > 4 || o != 1) return false;
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
    (void)o;
    return d <= 2 ? k_fused_fwd_i8<2, QN_I8_LMIN, 1> : k_fused_fwd_i8<4, QN_I8_LMIN, 1>;
}
