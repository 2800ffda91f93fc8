// Shared pieces of the int8-slice kernels (qn_fused_i8.hip, qn_wide_i8.hip): digit slicing, the kept digit products
// and their issue order, the slot swizzle of the weight digit planes, and the once-per-call weight slicing kernel.
// Everything sits in an anonymous namespace: each translation unit gets its own copy (not part of the C ABI).
#pragma once
#include "qn_common.h"
#include "qn_math.h"
#include <type_traits>
#include <utility>

#ifndef QN_I8_G
#define QN_I8_G 1              // 16-row groups per wave iteration (register budget of the pipelined epilogue: one)
#endif
#ifndef QN_I8_LMIN
#define QN_I8_LMIN 4
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

// Top digits (word 5 of slice4: four values' digit 5 as signed bytes) -> nonzero if one of them is >= 4 or <= -5 (give or
// take one: a carry may cross into the neighbouring byte), i.e. if one of the four values is at least ~2^-4.7 in
// magnitude.  The fused forward kernels OR this over a layer's activations of a wave's rows: zero = ALL of them are tiny,
// and the fixed activation scale 2^-46 (absolute error 2^-47) no longer gives relative accuracy -- those rows are redone in
// plain float64.  Otherwise the slicing error stays <= 2^-47 / 2^-4.7 = 1.8e-13 of the rows' largest activation.
__device__ __forceinline__ int top_digits_large(int s5) { return (s5 + 0x04040404) & (int)0xF8F8F8F8; }
// Largest exponent a sliced matrix may have (2^e > every entry of the row): the digits keep 2^-47 of the ROW MAXIMUM, so an
// outlier weight takes the small entries of its row their precision (norm-wise error bound 2^-47 max|W_j.| sum|a|) --
// harmless while the outlier's term dominates the sum, wrong when it is switched off exactly (a saturated unit behind it has
// derivative 0: tests/fuzz_all.py, weight 1e30).  Chains with |w| >= 2^20 in a sliced matrix take the plain-float64 paths.
constexpr int I8_MAX_WEIGHT_EXP = 20;
constexpr unsigned TINY_ACT_HI = 0x3FA00000u;       // high word of 2^-5: the same bound where float64 values are at hand

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

// slice4 for N groups of four values at once, STEP-MAJOR (every step for all groups before the next step): a wave that is
// (almost) alone on its SIMD then always has N independent instructions between a result and its use
template <int N>
__device__ __forceinline__ void slice4_n(const double (&a)[N][4], int (&S)[N][NS]) {
    int lo[N][4], hi[N][4], p01[N], q01[N], p23[N], q23[N], r01[N], r23[N];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double x = fma(a[i][r], 0x1p46, kMagic);
            lo[i][r] = __double2loint(x);
            hi[i][r] = __double2hiint(x);
        }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        p01[i] = __builtin_amdgcn_perm(lo[i][1], lo[i][0], 0x05010400); q01[i] = __builtin_amdgcn_perm(lo[i][1], lo[i][0], 0x07030602);
        p23[i] = __builtin_amdgcn_perm(lo[i][3], lo[i][2], 0x05010400); q23[i] = __builtin_amdgcn_perm(lo[i][3], lo[i][2], 0x07030602);
        r01[i] = __builtin_amdgcn_perm(hi[i][1], hi[i][0], 0x05010400); r23[i] = __builtin_amdgcn_perm(hi[i][3], hi[i][2], 0x05010400);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        S[i][0] = __builtin_amdgcn_perm(p23[i], p01[i], 0x05040100) ^ 0x80808080;
        S[i][1] = __builtin_amdgcn_perm(p23[i], p01[i], 0x07060302) ^ 0x80808080;
        S[i][2] = __builtin_amdgcn_perm(q23[i], q01[i], 0x05040100) ^ 0x80808080;
        S[i][3] = __builtin_amdgcn_perm(q23[i], q01[i], 0x07060302) ^ 0x80808080;
        S[i][4] = __builtin_amdgcn_perm(r23[i], r01[i], 0x05040100) ^ 0x80808080;
        S[i][5] = __builtin_amdgcn_perm(r23[i], r01[i], 0x07060302);
    }
}
// maximum over each row of 16 lanes for N values at once, step-major (a DPP operand needs two wait states behind its producer)
template <int N>
__device__ __forceinline__ void row16_max_u32_n(unsigned (&x)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = max(x[i], (unsigned)__builtin_amdgcn_update_dpp(0, (int)x[i], 0xB1, 0xF, 0xF, false));
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = max(x[i], (unsigned)__builtin_amdgcn_update_dpp(0, (int)x[i], 0x4E, 0xF, 0xF, false));
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = max(x[i], (unsigned)__builtin_amdgcn_update_dpp(0, (int)x[i], 0x141, 0xF, 0xF, false));
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = max(x[i], (unsigned)__builtin_amdgcn_update_dpp(0, (int)x[i], 0x140, 0xF, 0xF, false));
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
template <int LMIN, int NLEV, bool FRESH = true>
__device__ __forceinline__ void issue_product(int k, v4i (&acc)[NLEV], const v4i (&Af)[NS], const v4i (&B)[NS]) {
    // (k is a compile-time constant after unrolling; the switch makes the register indices static)
#define QN_PRODUCT(KK)                                                                                               \
    case KK:                                                                                                         \
        if constexpr (KK < nprod(LMIN)) {                                                                            \
            constexpr int wi = prod_wi(LMIN, KK), aj = prod_aj(LMIN, KK), l = wi + aj - LMIN;                          \
            acc[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Af[wi], B[aj], (FRESH && prod_first(LMIN, KK)) ? (v4i){0, 0, 0, 0} : acc[l], 0, 0, 0); \
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
// the same with a compile-time product index (no switch to fold: long unrolled sequences stay cheap to compile)
template <int LMIN, int NLEV, bool FRESH, int KK>
__device__ __forceinline__ void issue_product_c(v4i (&acc)[NLEV], const v4i (&Af)[NS], const v4i (&B)[NS]) {
    constexpr int wi = prod_wi(LMIN, KK), aj = prod_aj(LMIN, KK), l = wi + aj - LMIN;
    acc[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Af[wi], B[aj], (FRESH && prod_first(LMIN, KK)) ? (v4i){0, 0, 0, 0} : acc[l], 0, 0, 0);
}

struct I8Net {
    int nl;                                   // hidden->hidden layers handled (layer li maps dims[li+1] -> dims[li+2])
    int h[QN_MAX_LAYERS + 1];                 // h[0] = first hidden width, h[li+1] = output width of layer li
    int64_t offW[QN_MAX_LAYERS], offB[QN_MAX_LAYERS];   // offsets into a flat weight vector
    int64_t offD[QN_MAX_LAYERS];              // byte offset of layer li's digit planes inside a chain's block
    int64_t offS[QN_MAX_LAYERS];              // double offset of layer li's {scale, bias} pairs inside a chain's block
    int64_t p, dbytes, sdoubles;
    int has_bias;
};

__device__ __forceinline__ unsigned wave_max_u32(unsigned x) {
    x = row16_max_u32(x);
    const unsigned a = __builtin_amdgcn_readlane((int)x, 0), b = __builtin_amdgcn_readlane((int)x, 16);
    const unsigned c = __builtin_amdgcn_readlane((int)x, 32), d = __builtin_amdgcn_readlane((int)x, 48);
    return max(max(a, b), max(c, d));
}

// grid (B, layers, row parts): a workgroup slices rows [part * h_out / parts, ...) of one matrix of one chain
template <int LMIN>
__global__ __launch_bounds__(256) void k_i8_slice_w(I8Net net, const double* __restrict__ W, unsigned char* __restrict__ Wd,
                                                   double* __restrict__ sb, int* __restrict__ flags) {
    const int b = blockIdx.x, li = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bad = 0;
    const double* Wb = W + (int64_t)b * net.p;
    const int h_in = net.h[li], h_out = net.h[li + 1], nq = h_in / 4;
    const double* Wg = Wb + net.offW[li];
    unsigned char* planes = Wd + (int64_t)b * net.dbytes + net.offD[li];
    double* sbl = sb + (int64_t)b * net.sdoubles + net.offS[li];
    const int64_t plane = (int64_t)h_out * h_in;
    const int rows_per = (h_out + gridDim.z - 1) / gridDim.z;
    const int r0 = blockIdx.z * rows_per, r1 = r0 + rows_per < h_out ? r0 + rows_per : h_out;
    for (int row = r0 + wave; row < r1; row += 4) {
        unsigned ex = 0;
        for (int qd = lane; qd < nq; qd += 64) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = Wg[(int64_t)row * h_in + 4 * qd + r];
                bad |= !qn_bounded(v);
                ex = max(ex, ((unsigned)__double2hiint(v) & 0x7fffffffu) >> 20);
            }
        }
        int e = (int)wave_max_u32(ex) - 1022;
        bad |= e > I8_MAX_WEIGHT_EXP;
        e = e < -900 ? -900 : e;
        for (int qd = lane; qd < nq; qd += 64) {
            double an[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) an[r] = ldexp(Wg[(int64_t)row * h_in + 4 * qd + r], -e);
            int S[NS];
            slice4(an, S);
            const int i0 = 4 * qd, kc = i0 >> 6, m = (i0 & 63) >> 4, g = (i0 & 15) >> 2;
            unsigned char* dst = planes + (int64_t)row * h_in + 64 * kc + 16 * (g ^ slot_swz(row)) + 4 * m;
#pragma unroll
            for (int k = 0; k < NS; ++k) *reinterpret_cast<int*>(dst + k * plane) = S[k];
        }
        if (lane == 0) {
            const double bias = net.has_bias ? Wb[net.offB[li] + row] : 0.0;
            bad |= !qn_bounded(bias);
            sbl[2 * row] = ldexp(1.0, e - 2 * QB + 8 * LMIN);
            sbl[2 * row + 1] = bias;
        }
    }
    if (__any(bad) && lane == 0) atomicOr(&flags[b], 1);           // (flags are zeroed by a memset node ahead of this kernel)
}

}  // namespace
