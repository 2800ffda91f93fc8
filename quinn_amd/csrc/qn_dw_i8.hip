// Weight gradient of a hidden->hidden layer as SLICED EXACT PRODUCTS on the int8 matrix pipe:
//   dW[j][i] = sum_n dZ[j][n] a[i][n],  db[j] = sum_n dZ[j][n]        (float64 operands [B][h][Nb], n = data row)
// -- what k_gemm64<DW> of qn_generic.hip computes with float64 MFMAs (the reference's autograd through
// quinn/nns/mlp.py:92-101, quinn/nns/nnwrap.py:128-150), for the 128 / 256-wide tanh networks whose forward and
// activation-gradient passes run in qn_wide_i8.hip.
//
// The contraction runs over data rows, so BOTH operands are sliced on the fly, 64 rows (one K-chunk) at a time:
// a[i][n] in [-1, 1] (tanh outputs) with the fixed scale 2^-46, dZ[j][n] with one scale per feature j AND chunk
// (2^e > the largest |dZ[j][n]| of the chunk's 64 rows: 4 DPP steps over the 16 lanes that hold the row), six balanced
// base-256 digits each (qn_i8_slice.h).  A chunk's 26 digit products per 16 x 16 tile are exact int32 sums (pair sums of
// adjacent levels fit int32 at K = 64); they are recombined and added into float64 accumulators chunk by chunk, because
// the scales change from chunk to chunk.
//
// Workgroup = 8 waves = one 64 x 64 output tile of one chain over one K-slab, two ROLES (one wave of each per SIMD):
//   waves 0..3  slicers: load the chunk's 64 x 64 float64 blocks of dZ and a (a 16-lane group = 64 consecutive rows of
//               one feature: 512 contiguous bytes), scale, slice, write the digit planes [6][64 features][64 B] of both
//               operands (slot-swizzled: conflict-free ds_read_b128 fragments) + the chunk's 64 scales into one of TWO
//               LDS buffers; they also carry the bias row sums;
//   waves 4..7  matrix waves: wave m owns output rows 16 m .. 16 m + 15 x all 64 columns = 4 tiles: 6 + 4 x 6 fragment
//               reads and 4 x 26 MFMAs per chunk; tile t's int32 levels are recombined into its float64 accumulators
//               BETWEEN tile t + 1's MFMAs (pinned micro-steps; the last tile's beside the next chunk's first tile).
// One barrier per chunk.  The int8 pipe runs beside the slicers' vector work (tools/ubench_i8.hip: an MFMA wave keeps
// its full rate next to a VALU wave of the same SIMD).  Per chunk and SIMD: 104 MFMAs (1700 cycles) against ~310 + 180
// vector instructions.
// A chunk with a value that is not finite (or >= 2^900), or an a block whose values are all tiny (< 2^-5: the fixed scale
// would lose relative accuracy), makes the workgroup redo its tile in plain float64 at the end.
#include "qn_common.h"
#include "qn_fused_args.h"
#include "qn_math.h"
#include "qn_i8_slice.h"
#include <mutex>

namespace {

constexpr int DWT = 512;                                  // threads per workgroup
constexpr int DW_OPER = NS * 64 * 64;                     // one operand's digit planes of a chunk: 24 KB
constexpr int DW_BUF = 2 * DW_OPER + 64 * (int)sizeof(double);      // + the chunk's scales

struct DwArgs {
    int64_t out_stride_b, out_stride_k;
    int h_in, h_out, Nb, has_bias, kchunk;                // kchunk: rows per K-slab
    int Ns;                                               // row stride of both operands [B][h][Ns] (>= Nb)
    int nmain;                                            // k_i8_dw_g: rows it covers = Nb rounded down to whole 64-row chunks (the rest: k_dw_tail)
    int inner, outer_total, per_b;                        // XCD-aware 1-D grid as gemm_grid() of qn_generic.hip
};

template <int LMIN>
__global__ __launch_bounds__(DWT, 1) void k_i8_dw(DwArgs g, const double* __restrict__ dz, const double* __restrict__ ap,
                                                 double* __restrict__ out, const double* /*rowsc: k_i8_dw_g only*/) {
    constexpr int NLEV = 2 * (NS - 1) - LMIN + 1, NPROD = nprod(LMIN);
    static_assert(NLEV == 7 || NLEV == 6, "level recombination below is written for LMIN = 4 / 5");
    extern __shared__ __attribute__((aligned(16))) char smemd[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seq = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int inner_i = seq % g.inner, outer = (seq / g.inner) * 8 + xcd;
    if (outer >= g.outer_total) return;
    const int b = outer / g.per_b, slab = outer % g.per_b;
    const int tiles_i = g.h_in / 64;
    const int j0 = (inner_i / tiles_i) * 64, i0 = (inner_i % tiles_i) * 64;
    const int Nb = g.Nb, kbeg = slab * g.kchunk, kend = kbeg + g.kchunk < Nb ? kbeg + g.kchunk : Nb;
    const int nchunks = (kend - kbeg + 63) / 64;
    const int64_t Ns = g.Ns;
    const double* Z = dz + ((int64_t)b * g.h_out + j0) * Ns;
    const double* A = ap + ((int64_t)b * g.h_in + i0) * Ns;
    double* O = out + (int64_t)b * g.out_stride_b + (int64_t)slab * g.out_stride_k;
    int* badflag = reinterpret_cast<int*>(smemd + 2 * DW_BUF);
    if (tid == 0) *badflag = 0;
    const bool want_rowsum = g.has_bias && i0 == 0;

    // the 4 x 4 values of a lane's items of one operand block (items: feature 16 u + fl_; rows {2 q, 2 q + 1, 32 + 2 q, 33 + 2 q}
    // of the chunk: two 16-byte loads per item, each instruction 256 contiguous bytes per feature)
    auto load_block = [&](const double* base, int ch, int q16_, int fl_, double (&v)[4][4]) {
        const int c0 = kbeg + 64 * ch;
        if (c0 + 64 <= kend) {                                           // wave-uniform: a whole chunk
            // 16-byte loads at 8-byte aligned addresses: with an ODD row count (an ensemble member's 80 % subset: 13107 rows at
            // cfg4) every other feature's rows start 8 bytes off a 16-byte boundary; the hardware's unaligned mode serves them.
            // (Round 3 sent odd row counts through the scalar branch below: 4.3 ms per matrix and 128 members against 3.x here.)
            typedef double d2u_ __attribute__((ext_vector_type(2), aligned(8)));
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const d2u_* sp = reinterpret_cast<const d2u_*>(base + (int64_t)(16 * u + fl_) * Ns + c0 + 2 * q16_);
                const d2u_ v01 = sp[0], v23 = sp[16];
                v[u][0] = v01.x; v[u][1] = v01.y; v[u][2] = v23.x; v[u][3] = v23.y;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = c0 + 32 * (r >> 1) + 2 * q16_ + (r & 1);
                    v[u][r] = n < kend ? base[(int64_t)(16 * u + fl_) * Ns + n] : 0.0;
                }
        }
    };
    if (wave < 4) {
        // ------------------------------------------------------------------ slicers
        // item (u, wave, lane): feature f = 16 u + 4 wave + (lane >> 4), 4 of the chunk's rows chosen by q16 = lane & 15
        const int q16 = lane & 15, m4 = q16 >> 2, g4 = q16 & 3, fl = 4 * wave + (lane >> 4);
        double rsum[4] = {0.0, 0.0, 0.0, 0.0};
        int bad = 0;
        // a chunk's 4 x 4 + 4 x 4 values of this lane: loaded ONE CHUNK AHEAD of their slicing (two register sets; the
        // first version loaded and sliced in the same step and spent a memory round trip per chunk: 5800 cycles per chunk
        // against 1700 of MFMA work)
        // Of the chunk's 64 rows the lane takes rows {2 q16, 2 q16 + 1, 32 + 2 q16, 33 + 2 q16}: two 16-byte loads per
        // operand and feature, each instruction 256 contiguous bytes per feature (whole cache lines; 4 consecutive rows
        // per lane made every instruction touch twice the lines for the same bytes).  Both operands use the same
        // row -> K-slot map, so any map is as good as another.
        // (the slicer waves take the dZ block: exponents, scales, bias sums; the a block, cheaper per item, is sliced by
        // the matrix waves between their barrier and their MFMAs -- with all slicing on these four waves they were the
        // critical path and the matrix waves idle a third of the time)
        auto load_chunk = [&](int ch, double (&vz)[4][4]) { load_block(Z, ch, q16, fl, vz); };
        auto slice_chunk = [&](int ch, const double (&vz)[4][4]) {
            char* buf = smemd + (ch & 1) * DW_BUF;
            unsigned char* pa = reinterpret_cast<unsigned char*>(buf);
            double* scl = reinterpret_cast<double*>(buf + 2 * DW_OPER);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = 16 * u + fl;
                unsigned ex = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) ex = max(ex, ((unsigned)__double2hiint(vz[u][r]) & 0x7fffffffu) >> 20);
                bad |= ex >= 1923u;                                      // |dZ| >= 2^900 or not finite
                int e = (int)row16_max_u32(ex) - 1022;                   // 2^e > every |dZ| of the feature's 64 rows
                e = e < -900 ? -900 : e;
                const double dn = __hiloint2double((1023 - e) << 20, 0);               // 2^-e
                double an[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) an[r] = vz[u][r] * dn;       // exact, |an| < 1
                int S[NS];
                slice4(an, S);
                const int ofs = f * 64 + 16 * (g4 ^ slot_swz(f)) + 4 * m4;
#pragma unroll
                for (int k = 0; k < NS; ++k) *reinterpret_cast<int*>(pa + k * 4096 + ofs) = S[k];
                if (q16 == 0) scl[f] = __hiloint2double((1023 + e - 2 * QB + 8 * LMIN) << 20, 0);   // integer sum -> dZ . a
                if (want_rowsum) rsum[u] += (vz[u][0] + vz[u][1]) + (vz[u][2] + vz[u][3]);
            }
        };
#ifdef QN_DW_STAMPS
        long long ts_slice = 0, ts_bar = 0, ts_0 = __builtin_amdgcn_s_memtime();
#define QN_DST(var, code) { const long long t_ = __builtin_amdgcn_s_memtime(); code; var += __builtin_amdgcn_s_memtime() - t_; }
#else
#define QN_DST(var, code) { code; }
#endif
        double z0[4][4], z1[4][4];
        if (nchunks > 0) load_chunk(0, z0);
        for (int ch = 0; ch < nchunks; ch += 2) {
            if (ch + 1 < nchunks) load_chunk(ch + 1, z1);
            QN_DST(ts_slice, slice_chunk(ch, z0))
            // chunk ch is written; the matrix waves are done with chunk ch - 1 (whose buffer chunk ch + 1 overwrites)
            QN_DST(ts_bar, __syncthreads())
            if (ch + 1 < nchunks) {
                if (ch + 2 < nchunks) load_chunk(ch + 2, z0);
                QN_DST(ts_slice, slice_chunk(ch + 1, z1))
                QN_DST(ts_bar, __syncthreads())
            }
        }
#ifdef QN_DW_STAMPS
        if (blockIdx.x == 9 && tid == 0)
            printf("dw slicer: chunks %d total %lld slice(+load wait) %lld barrier %lld\n", nchunks, (long long)(__builtin_amdgcn_s_memtime() - ts_0), ts_slice, ts_bar);
#endif
        __syncthreads();                                                 // (matches the matrix waves' drain step)
        if (want_rowsum) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                double s = rsum[u];
                s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
                if (q16 == 0) O[(int64_t)g.h_in * g.h_out + j0 + 16 * u + fl] = s;
            }
        }
        if (bad) *badflag = 1;
    } else {
        // ------------------------------------------------------------------ matrix waves
        const int m = wave - 4, q = lane >> 4, c = lane & 15;
        const int lofs = c * 64 + 16 * (q ^ slot_swz(c));               // this lane's 16 bytes inside a [16 rows][64 B] block
        // these waves also slice the a block (tanh outputs: fixed scale, no exponent work) of the NEXT chunk, before their
        // MFMAs on the current one; the block after that is requested right behind it
        const int q16b = lane & 15, m4b = q16b >> 2, g4b = q16b & 3, flb = 4 * m + (lane >> 4);
        int badb = 0;
        unsigned amaxb = 0;                                              // largest |a| (high word) this wave has sliced
        double vb[4][4];
        auto slice_b = [&](int ch) {
            unsigned char* pbn = reinterpret_cast<unsigned char*>(smemd + (ch & 1) * DW_BUF) + DW_OPER;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = 16 * u + flb;
                unsigned exa = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) exa = max(exa, (unsigned)__double2hiint(vb[u][r]) & 0x7fffffffu);
                badb |= exa >= 0x40000000u;                              // |a| >= 2 or not finite
                amaxb = max(amaxb, exa);
                int S[NS];
                slice4(vb[u], S);
                const int ofs = f * 64 + 16 * (g4b ^ slot_swz(f)) + 4 * m4b;
#pragma unroll
                for (int k = 0; k < NS; ++k) *reinterpret_cast<int*>(pbn + k * 4096 + ofs) = S[k];
            }
        };
        if (nchunks > 0) {
            load_block(A, 0, q16b, flb, vb);
            slice_b(0);
            load_block(A, nchunks > 1 ? 1 : 0, q16b, flb, vb);
        }
        double facc[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) facc[t][r] = 0.0;
        v4i acc[2][NLEV];                                                // the tile in flight / the tile being recombined
        double sc[4], scp[4];                                            // this / the previous chunk's scales of rows 16 m + 4 q + r
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[r] = scp[r] = 0.0;
        // One tile step: the 26 MFMAs of tile T (into acc[T & 1]) dealt out between the micro-steps that recombine the
        // PREVIOUS tile's levels (acc[(T - 1) & 1]: pairs in int32, then float64) into its accumulators -- pinned, as in
        // qn_wide_i8.hip: left to itself a wave runs fragment reads, MFMAs and the recombination one after the other
        // (measured 3500 cycles per chunk against 1700 of MFMA work).  The next tile's B fragments are fetched meanwhile.
        auto tile_step = [&](auto t_tag, auto conv_tag, const v4i (&Af)[NS], v4i (&Bf)[2][NS], const unsigned char* pb, const double (&s)[4]) {
            constexpr int T_ = decltype(t_tag)::value, TP = (T_ + 3) & 3;          // TP: the tile recombined here
            constexpr bool CONV = decltype(conv_tag)::value;
            constexpr int NMICRO = 20;
            double ts[4];
            if constexpr (T_ < 3) {
#pragma unroll
                for (int k = 0; k < NS; ++k) Bf[(T_ + 1) & 1][k] = *reinterpret_cast<const v4i*>(pb + k * 4096 + (T_ + 1) * 1024 + lofs);
            }
            for_each_stage([&](auto id_tag) {
                constexpr int id = decltype(id_tag)::value, st = id >> 2, r = id & 3;
                constexpr int from = (id * NPROD + NMICRO - 1) / NMICRO, upto = ((id + 1) * NPROD + NMICRO - 1) / NMICRO;
                for_each_stage([&](auto k_tag) {
                    constexpr int k = from + decltype(k_tag)::value;
                    issue_product_c<LMIN, NLEV, true, k>(acc[T_ & 1], Af, Bf[T_ & 1]);
                }, std::make_integer_sequence<int, upto - from>{});
                if constexpr (CONV) {
                    const v4i (&ap_)[NLEV] = acc[TP & 1];
                    if constexpr (st == 0) {
                        if constexpr (NLEV == 7) ts[r] = (double)ap_[6][r];
                        else ts[r] = (double)(ap_[4][r] + (ap_[5][r] << 8));
                    } else if constexpr (st < 4) {
                        constexpr int lo = NLEV == 7 ? 6 - 2 * st : 4 - 2 * st;
                        if constexpr (lo >= 0) ts[r] = fma(ts[r], 65536.0, (double)(ap_[lo][r] + (ap_[lo + 1][r] << 8)));
                    } else {
                        facc[TP][r] = fma(ts[r], s[r], facc[TP][r]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }, std::make_integer_sequence<int, NMICRO>{});
        };
#ifdef QN_DW_STAMPS
        long long tm_bar = 0, tm_0 = __builtin_amdgcn_s_memtime();
#endif
        for (int ch = 0; ch < nchunks; ++ch) {
#ifdef QN_DW_STAMPS
            { const long long t_ = __builtin_amdgcn_s_memtime(); __syncthreads(); tm_bar += __builtin_amdgcn_s_memtime() - t_; }
#else
            __syncthreads();                                             // chunk ch is in LDS
#endif
            const char* buf = smemd + (ch & 1) * DW_BUF;
            const unsigned char* pa = reinterpret_cast<const unsigned char*>(buf);
            const unsigned char* pb = pa + DW_OPER;
            const double* scl = reinterpret_cast<const double*>(buf + 2 * DW_OPER);
            v4i Af[NS], Bf[2][NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) Af[k] = *reinterpret_cast<const v4i*>(pa + k * 4096 + m * 1024 + lofs);
#pragma unroll
            for (int k = 0; k < NS; ++k) Bf[0][k] = *reinterpret_cast<const v4i*>(pb + k * 4096 + lofs);
#pragma unroll
            for (int r = 0; r < 4; ++r) { scp[r] = sc[r]; sc[r] = scl[16 * m + 4 * q + r]; }
            if (ch + 1 < nchunks) slice_b(ch + 1);                       // (into the other buffer: nobody reads it before the next barrier)
            load_block(A, ch + 2 < nchunks ? ch + 2 : ch, q16b, flb, vb);     // (unconditional: keeps the registers' live range short)
            // (tile 0 runs beside the previous chunk's last tile)
            if (ch > 0) tile_step(std::integral_constant<int, 0>{}, std::true_type{}, Af, Bf, pb, scp);
            else tile_step(std::integral_constant<int, 0>{}, std::false_type{}, Af, Bf, pb, scp);
            tile_step(std::integral_constant<int, 1>{}, std::true_type{}, Af, Bf, pb, sc);
            tile_step(std::integral_constant<int, 2>{}, std::true_type{}, Af, Bf, pb, sc);
            tile_step(std::integral_constant<int, 3>{}, std::true_type{}, Af, Bf, pb, sc);
        }
#ifdef QN_DW_STAMPS
        if (blockIdx.x == 9 && tid == 256)
            printf("dw matrix wave: chunks %d total %lld barrier %lld\n", nchunks, (long long)(__builtin_amdgcn_s_memtime() - tm_0), tm_bar);
#endif
        // the fixed scale of the a operand (absolute error 2^-47) is a relative accuracy only while the activations are not
        // ALL tiny: a wave whose 16 features stayed below 2^-5 over the whole slab (and are not exactly zero: padding)
        // sends the tile through the plain float64 loop as well
        amaxb = wave_max_u32(amaxb);
        if (badb || (amaxb != 0 && amaxb < TINY_ACT_HI)) *badflag = 1;
        __syncthreads();                                                 // drain step
        if (nchunks > 0) {                                               // the last chunk's last tile
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const v4i (&ap_)[NLEV] = acc[1];
                double ts;
                if constexpr (NLEV == 7) {
                    ts = (double)ap_[6][r];
                    ts = fma(ts, 65536.0, (double)(ap_[4][r] + (ap_[5][r] << 8)));
                } else {
                    ts = (double)(ap_[4][r] + (ap_[5][r] << 8));
                }
                ts = fma(ts, 65536.0, (double)(ap_[2][r] + (ap_[3][r] << 8)));
                ts = fma(ts, 65536.0, (double)(ap_[0][r] + (ap_[1][r] << 8)));
                facc[3][r] = fma(ts, sc[r], facc[3][r]);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) O[(int64_t)(j0 + 16 * m + 4 * q + r) * g.h_in + i0 + 16 * t + c] = facc[t][r];
    }
    __syncthreads();
    if (*badflag) {
        // exceptional values: the tile again in plain float64 (thread = 8 outputs of one row j)
        const int jl = tid >> 3, ib = (tid & 7) * 8;
        double s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sb = 0.0;
        for (int n = kbeg; n < kend; ++n) {
            const double zv = Z[(int64_t)jl * Ns + n];
            sb += zv;
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] = fma(zv, A[(int64_t)(ib + u) * Ns + n], s[u]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) O[(int64_t)(j0 + jl) * g.h_in + i0 + ib + u] = s[u];
        if (want_rowsum && ib == 0) O[(int64_t)g.h_in * g.h_out + j0 + jl] = sb;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// The same product with GROUP scales (the kernel that runs; k_i8_dw above is kept for A/B: -DQN_DW_GROUP=0).
// In k_i8_dw the matrix waves are the critical path (94 % busy, docs/experiments.md R3.7) and a third of their vector work
// is the recombination of every tile's int32 levels into float64 after EVERY chunk -- forced by dZ scales that change from
// chunk to chunk.  Here a feature's dZ scale is shared by GC consecutive chunks (64 GC rows):
//   * the slicer waves read the exponents of a group one group AHEAD (a second, early read of the same lines -- they come
//     from L2 / MALL when the slicing read follows -- folded into a running maximum: 2 instructions per value), and fold
//     the scale into the rounding fma (no separate scaling multiply);
//   * the matrix waves keep ALL FOUR tiles' int32 levels in registers across the group's chunks (|level sum| <= GC * 6 * 2^20)
//     and recombine once per group: tile 0 beside the group's last tile step, tiles 1..3 beside the first three tile steps
//     of the next group (whose MFMAs restart those accumulators from zero one step later).  The levels are then summed in
//     float64 (their pair sums fit int32 only for a single chunk);
//   * the float64 accumulators live in LDS (lane-private slots, touched once per group) and the group's scales in a
//     two-slot LDS ring: the registers they held carry the two extra tiles' levels.
// Accuracy: a feature's digits keep 2^-47 of its largest |dZ| over 64 GC rows instead of 64 rows; the bound on the sum
// (2^-47 max|dZ_j| sum|a|) is the one of the forward kernels' row scales.
#ifndef QN_DW_GROUP
#define QN_DW_GROUP 4
#endif
constexpr int DWG_BUF = 2 * DW_OPER;                                  // a chunk's digit planes of both operands: 48 KB
constexpr int DWG_RING = 2 * 64 * (int)sizeof(double);                // scales of the current / previous group
constexpr int DWG_FACC = 4 * 16 * 64 * (int)sizeof(double);           // float64 accumulators of the four matrix waves: 32 KB
constexpr int DWG_LDS = 2 * DWG_BUF + DWG_RING + DWG_FACC + 16;

// UNB (round 4: relu / identity networks, whose activations are not bounded by 1): `rowsc` [B][Nb] holds the scale 2^f_n > every
// |a[.][n]| of data row n that the forward sliced the row with (k_i8_wide_fwd_u).  The contraction runs over n, so the scale
// moves to the other operand: a[i][n] 2^-f_n (in (-1, 1): the fixed scale of the tanh case) against dZ[j][n] 2^f_n (whose feature /
// group scales are found from the scaled values); powers of two, the product is unchanged.  Bias sums use the raw dZ.
template <int LMIN, int GC, bool UNB = false>
__global__ __launch_bounds__(DWT, 1) void k_i8_dw_g(DwArgs g, const double* __restrict__ dz, const double* __restrict__ ap,
                                                   double* __restrict__ out, const double* __restrict__ rowsc) {
    constexpr int NLEV = 2 * (NS - 1) - LMIN + 1, NPROD = nprod(LMIN);
    static_assert(GC >= 2 && (GC & (GC - 1)) == 0 && GC <= 64, "chunks per scale group: a power of two (int32 level sums: GC * 6 * 2^20)");
    extern __shared__ __attribute__((aligned(16))) char smemd[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seq = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int inner_i = seq % g.inner, outer = (seq / g.inner) * 8 + xcd;
    if (outer >= g.outer_total) return;
    const int b = outer / g.per_b, slab = outer % g.per_b;
    const int tiles_i = g.h_in / 64;
    const int j0 = (inner_i / tiles_i) * 64, i0 = (inner_i % tiles_i) * 64;
    const int Nb = g.Nb, kbeg = slab * g.kchunk, kend = kbeg + g.kchunk < g.nmain ? kbeg + g.kchunk : g.nmain;     // (Nb: the row stride)
    const int nchunks = (kend - kbeg + 63) / 64, npad = (nchunks + GC - 1) / GC * GC;
    const int64_t Ns = g.Ns;
    const double* Z = dz + ((int64_t)b * g.h_out + j0) * Ns;
    const double* A = ap + ((int64_t)b * g.h_in + i0) * Ns;
    double* O = out + (int64_t)b * g.out_stride_b + (int64_t)slab * g.out_stride_k;
    double* ring = reinterpret_cast<double*>(smemd + 2 * DWG_BUF);
    double* faccs = reinterpret_cast<double*>(smemd + 2 * DWG_BUF + DWG_RING);
    int* badflag = reinterpret_cast<int*>(smemd + 2 * DWG_BUF + DWG_RING + DWG_FACC);
    if (tid == 0) *badflag = 0;
    const bool want_rowsum = g.has_bias && i0 == 0;
    if (nchunks <= 0) {                                               // an empty K-slab: zeros (the loops below assume a chunk)
        for (int e = tid; e < 64 * 64; e += DWT) O[(int64_t)(j0 + (e >> 6)) * g.h_in + i0 + (e & 63)] = 0.0;
        if (want_rowsum && tid < 64) O[(int64_t)g.h_in * g.h_out + j0 + tid] = 0.0;
        return;
    }
    double magic = kMagic;                                            // (opaque register pair: see slice4_scaled, qn_fused_bwd_i8.hip)
    asm volatile("" : "+v"(magic));
    const double* RS = UNB ? rowsc + (int64_t)b * Nb : nullptr;
    // the row scales of a lane's 4 rows of chunk ch (rows {2 q, 2 q + 1, 32 + 2 q, 33 + 2 q}: as load_block)
    // (16-byte loads at 8-byte aligned addresses, served by the hardware's unaligned mode: with an odd row count every other
    // feature's rows start 8 bytes off a 16-byte boundary)
    typedef double d2u_ __attribute__((ext_vector_type(2), aligned(8)));
    auto load_rs = [&](int ch, int q16_, double (&r)[4]) {
        const d2u_* sp = reinterpret_cast<const d2u_*>(RS + kbeg + 64 * ch + 2 * q16_);
        const d2u_ v01 = sp[0], v23 = sp[16];
        r[0] = v01.x; r[1] = v01.y; r[2] = v23.x; r[3] = v23.y;
    };

    // (items and rows of a lane: as in k_i8_dw; whole chunks only: the rows beyond nmain = Nb - Nb % 64 are k_dw_tail's)
    auto load_block = [&](const double* base, int ch, int q16_, int fl_, double (&v)[4][4]) {
        const int c0 = kbeg + 64 * ch;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const d2u_* sp = reinterpret_cast<const d2u_*>(base + (int64_t)(16 * u + fl_) * Ns + c0 + 2 * q16_);
            const d2u_ v01 = sp[0], v23 = sp[16];
            v[u][0] = v01.x; v[u][1] = v01.y; v[u][2] = v23.x; v[u][3] = v23.y;
        }
    };
    // four values -> six digit words, x = v * scale + magic in ONE rounding (what slice4 does to v * 2^-e, exactly)
    auto slice_item = [&](const double (&v)[4], double scale, int (&S)[NS]) {
        int lo[4], hi[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double x;
            asm("v_fma_f64 %0, %1, %2, %3" : "=v"(x) : "v"(v[r]), "v"(scale), "v"(magic));
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
        S[5] = __builtin_amdgcn_perm(r23, r01, 0x07060302);
    };
    if (wave < 4) {
        // ------------------------------------------------------------------ slicers: the dZ block
        const int q16 = lane & 15, m4 = q16 >> 2, g4 = q16 & 3, fl = 4 * wave + (lane >> 4);
        double rsum[4] = {0.0, 0.0, 0.0, 0.0};
        int bad = 0;
        unsigned pmax[4] = {0u, 0u, 0u, 0u};                             // largest |dZ| high word seen in the group being read ahead
        double dn[4] = {0.0, 0.0, 0.0, 0.0};                             // 2^(46 - e) of the group being sliced
        double rsc[4] = {1.0, 1.0, 1.0, 1.0}, rsp[4] = {1.0, 1.0, 1.0, 1.0};     // UNB: row scales of the chunk being sliced / read ahead
        auto fold = [&](const double (&pz)[4][4]) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (UNB) pmax[u] = max(pmax[u], (unsigned)__double2hiint(pz[u][r] * rsp[r]) & 0x7fffffffu);   // (0 x 2^f = 0)
                    else
                    pmax[u] = max(pmax[u], (unsigned)__double2hiint(pz[u][r]) & 0x7fffffffu);
                    // (the low words count as used until here: dead on arrival, the allocator hands them out as scratch
                    // registers while the load is still in flight, and every such write waits for ALL outstanding loads)
                    asm volatile("" :: "v"(__double2loint(pz[u][r])));
                }
#pragma unroll
            for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(pmax[u]));     // folded HERE (sunk to its use, it waits for the loads issued meanwhile)
        };
        auto group_scale = [&](int grp) {
            unsigned ex[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { ex[u] = pmax[u] >> 20; pmax[u] = 0u; }
            row16_max_u32_n<4>(ex);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                bad |= ex[u] >= 1923u;                                   // |dZ| >= 2^900 or not finite
                int e = (int)ex[u] - 1022;                               // 2^e > every |dZ| of the feature's 64 GC rows
                e = e < -900 ? -900 : e;
                dn[u] = __hiloint2double((1023 + QB - e) << 20, 0);
                if (q16 == 0) ring[(grp & 1) * 64 + 16 * u + fl] = __hiloint2double((1023 + e - 2 * QB + 8 * LMIN) << 20, 0);
            }
        };
        auto slice_chunk = [&](int ch, const double (&vz)[4][4]) {
            unsigned char* pa = reinterpret_cast<unsigned char*>(smemd + (ch & 1) * DWG_BUF);
            if (want_rowsum) {                                           // (a real branch: one workgroup in h_in / 64 carries the bias sums)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int u = 0; u < 4; ++u) rsum[u] += (vz[u][0] + vz[u][1]) + (vz[u][2] + vz[u][3]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = 16 * u + fl;
                int S[NS];
                if constexpr (UNB) {
                    // (dZ 2^f_n first: 0 stays 0 whatever the scales; the product with 2^(46 - e) is the one rounding)
                    double vs[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) vs[r] = vz[u][r] * rsc[r];
                    slice_item(vs, dn[u], S);
                } else {
                    slice_item(vz[u], dn[u], S);
                }
                const int ofs = f * 64 + 16 * (g4 ^ slot_swz(f)) + 4 * m4;
#pragma unroll
                for (int k = 0; k < NS; ++k) *reinterpret_cast<int*>(pa + k * 4096 + ofs) = S[k];
                __builtin_amdgcn_sched_barrier(0);                       // (item by item: interleaving all four costs more registers than the sets leave)
            }
        };
        // the a block (tanh outputs: fixed scale 2^-46, no exponent work) of the same chunk
        unsigned amaxa = 0;                                              // largest |a| (high word) this wave has sliced
        auto slice_chunk_a = [&](int ch, const double (&va)[4][4]) {
            unsigned char* pbn = reinterpret_cast<unsigned char*>(smemd + (ch & 1) * DWG_BUF) + DW_OPER;
            const double p46 = 0x1p46;
            double ainv[4];                                              // UNB: 2^-f_n of the lane's 4 rows (exponent arithmetic)
            if constexpr (UNB) {
#pragma unroll
                for (int r = 0; r < 4; ++r) ainv[r] = __hiloint2double((2046 << 20) - __double2hiint(rsc[r]), 0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = 16 * u + fl;
                unsigned exa = 0;
                double vn[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    vn[r] = UNB ? va[u][r] * ainv[r] : va[u][r];         // (exact: a power of two; |vn| < 1 by the forward's choice of f_n)
                    exa = max(exa, (unsigned)__double2hiint(vn[r]) & 0x7fffffffu);
                }
                bad |= exa >= 0x40000000u;                               // |a| >= 2 or not finite
                amaxa = max(amaxa, exa);
                int S[NS];
                slice_item(vn, p46, S);
                const int ofs = f * 64 + 16 * (g4 ^ slot_swz(f)) + 4 * m4;
#pragma unroll
                for (int k = 0; k < NS; ++k) *reinterpret_cast<int*>(pbn + k * 4096 + ofs) = S[k];
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // A slab's last group is filled up with empty chunks (zero digits): every group is GC chunk steps, no special cases
        // on the matrix waves.  Every load below is UNCONDITIONAL, with the chunk index clamped to the slab's last one: a load
        // inside a branch meets the other path's registers at the join, and the compiler then waits for it on the spot.
        double zc[4][4], a0[4][4], a1[4][4], pz[4][4];
        const int lastc = nchunks - 1;
        auto zero_planes = [&](int ch, int oper) {
            unsigned char* pa = reinterpret_cast<unsigned char*>(smemd + (ch & 1) * DWG_BUF) + oper;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = 16 * u + fl, ofs = f * 64 + 16 * (g4 ^ slot_swz(f)) + 4 * m4;
#pragma unroll
                for (int k = 0; k < NS; ++k) *reinterpret_cast<int*>(pa + k * 4096 + ofs) = 0;
            }
        };
        if (q16 == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) ring[64 + 16 * u + fl] = 0.0;     // (slot 1 is read beside group 0's first chunk: 0 x 0)
        }
#pragma unroll
        for (int c = 0; c < GC; ++c) {                                                            // group 0's exponents
            if constexpr (UNB) load_rs(min(c, lastc), q16, rsp);
            load_block(Z, min(c, lastc), q16, fl, pz);
            fold(pz);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (UNB) { load_rs(0, q16, rsc); load_rs(min(GC, lastc), q16, rsp); }
        __builtin_amdgcn_sched_barrier(0);
        // (in the order and number the loop leaves them outstanding at its head: the compiler's wait counts at the head are
        // the worst case over the two ways in, and with another order there it waited for EVERYTHING in every other step)
        load_block(A, 0, q16, fl, a0);
        __builtin_amdgcn_sched_barrier(0);
        load_block(Z, 0, q16, fl, zc);
        __builtin_amdgcn_sched_barrier(0);
        load_block(A, min(1, lastc), q16, fl, a1);
        __builtin_amdgcn_sched_barrier(0);
        load_block(Z, min(GC, lastc), q16, fl, pz);
        __builtin_amdgcn_sched_barrier(0);
        // an a block is requested TWO chunk steps before it is sliced (into the register set the chunk two steps back has
        // just left: it comes from HBM for the first workgroup of an XCD that touches it, and one chunk step is shorter than
        // that round trip); a dZ block one step before (its lines were pulled into L2 by the exponent read a group earlier:
        // the chunk GC ahead belongs to the next group -- past the slab's end the last chunk is read again, which belongs to
        // the last group anyway)
#ifdef QN_DW_STAMPS
        long long tsg_bar = 0, tsg_wz = 0, tsg_a = 0, tsg_f = 0;
#endif
        auto step = [&](int ch, double (&va)[4][4]) {
            if ((ch & (GC - 1)) == 0) group_scale(ch / GC);
#ifdef QN_DW_STAMPS
            { const long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); tsg_wz += __builtin_amdgcn_s_memtime() - t_; }
#endif
            if (ch < nchunks) slice_chunk(ch, zc); else zero_planes(ch, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_block(Z, min(ch + 1, lastc), q16, fl, zc);
            __builtin_amdgcn_sched_barrier(0);
#ifdef QN_DW_STAMPS
            const long long ta_ = __builtin_amdgcn_s_memtime();
#endif
            if (ch < nchunks) slice_chunk_a(ch, va); else zero_planes(ch, DW_OPER);
#ifdef QN_DW_STAMPS
            tsg_a += __builtin_amdgcn_s_memtime() - ta_;
#endif
            __builtin_amdgcn_sched_barrier(0);
            load_block(A, min(ch + 2, lastc), q16, fl, va);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (UNB) { load_rs(min(ch + 1, lastc), q16, rsc); __builtin_amdgcn_sched_barrier(0); }   // (both slicings of chunk ch are done)
#ifdef QN_DW_STAMPS
            { const long long t_ = __builtin_amdgcn_s_memtime(); __syncthreads(); tsg_bar += __builtin_amdgcn_s_memtime() - t_; }
#else
            __syncthreads();          // chunk ch is written; the matrix waves are done with chunk ch - 1 (whose buffer chunk ch + 1 overwrites)
#endif
#ifdef QN_DW_STAMPS
            { const long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); tsg_f += __builtin_amdgcn_s_memtime() - t_; }
#endif
            fold(pz);                 // (exponents of chunk ch + GC, requested a step ago)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (UNB) load_rs(min(ch + 1 + GC, lastc), q16, rsp);
            load_block(Z, min(ch + 1 + GC, lastc), q16, fl, pz);
            __builtin_amdgcn_sched_barrier(0);
        };
#ifdef QN_DW_STAMPS
        const long long tsg_0 = __builtin_amdgcn_s_memtime();
#endif
        for (int ch = 0; ch < npad; ch += 2) {                            // (npad is even: GC is)
            step(ch, a0);
            step(ch + 1, a1);
        }
#ifdef QN_DW_STAMPS
        if (blockIdx.x == 9 && tid == 0)
            printf("dwg slicer: chunks %d total %lld barrier %lld wait-z %lld slice-a %lld wait-peek %lld\n", npad, (long long)(__builtin_amdgcn_s_memtime() - tsg_0), tsg_bar, tsg_wz, tsg_a, tsg_f);
#endif
        // the fixed scale of the a operand (absolute error 2^-47) is a relative accuracy only while the activations are not
        // ALL tiny: a wave whose features stayed below 2^-5 over the whole slab (and are not exactly zero: padding) sends
        // the tile through the plain float64 loop as well
        amaxa = wave_max_u32(amaxa);
        if (amaxa != 0 && amaxa < TINY_ACT_HI) bad = 1;
        __syncthreads();                                                 // (matches the matrix waves' drain step)
        if (want_rowsum) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                double s = rsum[u];
                s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
                if (q16 == 0) O[(int64_t)g.h_in * g.h_out + j0 + 16 * u + fl] = s;
            }
        }
        if (bad) *badflag = 1;
    } else {
        // ------------------------------------------------------------------ matrix waves
        const int m = wave - 4, q = lane >> 4, c = lane & 15;
        const int lofs = c * 64 + 16 * (q ^ slot_swz(c));
        double* fl_ = faccs + m * 16 * 64 + lane;                        // element (tile t, register r): fl_[(4 t + r) * 64]
#pragma unroll
        for (int e = 0; e < 16; ++e) fl_[e * 64] = 0.0;
        v4i acc[4][NLEV];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int l = 0; l < NLEV; ++l) acc[t][l] = (v4i){0, 0, 0, 0};
        // One tile step: the 26 MFMAs of tile T_ and, when TP >= 0, the recombination of tile TP's
        // levels (a finished group) into its float64 accumulators with the scales of ring slot `par`, dealt out between them
        // in pinned micro-steps.  The next tile's B fragments are fetched meanwhile.
        auto tile_step = [&](auto t_tag, auto tp_tag, const v4i (&Af)[NS], v4i (&Bf)[NS], const unsigned char* pb, int par) {
            constexpr int T_ = decltype(t_tag)::value, TP = decltype(tp_tag)::value;

            // product k in B-DIGIT-MAJOR order (the kept set and its level sequence are symmetric in the two digit indices): a
            // B digit's fragment is dead after its block of products and is refilled at once with the next tile's (one
            // fragment set instead of two: the registers carry the extra tiles' levels)
            auto product = [&](auto k_tag) {
                constexpr int k = decltype(k_tag)::value, aj = prod_wi(LMIN, k), wi = prod_aj(LMIN, k), l = wi + aj - LMIN;
                acc[T_][l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Af[wi], Bf[aj], acc[T_][l], 0, 0, 0);
                if constexpr (k == NPROD - 1 || prod_wi(LMIN, k + 1) != aj) {
                    if constexpr (T_ < 3) Bf[aj] = *reinterpret_cast<const v4i*>(pb + aj * 4096 + (T_ + 1) * 1024 + lofs);
                    __builtin_amdgcn_sched_barrier(0);                   // (left free, the scheduler hoists every refill to the top of the chunk: six more fragment sets alive)
                }
            };
            if constexpr (TP < 0) {
                for_each_stage(product, std::make_integer_sequence<int, NPROD>{});
            } else {
                constexpr int NST = NLEV + 1, NMICRO = 4 * NST;
                double ts[4];
                for_each_stage([&](auto id_tag) {
                    constexpr int id = decltype(id_tag)::value, st = id >> 2, r = id & 3;
                    constexpr int from = (id * NPROD + NMICRO - 1) / NMICRO, upto = ((id + 1) * NPROD + NMICRO - 1) / NMICRO;
                    for_each_stage([&](auto k_tag) { product(std::integral_constant<int, from + decltype(k_tag)::value>{}); },
                                   std::make_integer_sequence<int, upto - from>{});
                    const v4i (&ap_)[NLEV] = acc[TP];
                    if constexpr (st == 0) ts[r] = (double)ap_[NLEV - 1][r];
                    else if constexpr (st < NLEV) ts[r] = fma(ts[r], 256.0, (double)ap_[NLEV - 1 - st][r]);
                    else {
                        fl_[(4 * TP + r) * 64] = fma(ts[r], ring[par * 64 + 16 * m + 4 * q + r], fl_[(4 * TP + r) * 64]);
                        if constexpr (r == 3) {                          // the tile's levels restart from zero for the next group
#pragma unroll
                            for (int l = 0; l < NLEV; ++l) acc[TP][l] = (v4i){0, 0, 0, 0};
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }, std::make_integer_sequence<int, NMICRO>{});
            }
        };
        using std::integral_constant;
        using no_tile = integral_constant<int, -1>;
        // (one straight-line body per group: with run-time choices between recombining and plain steps the accumulators met
        // at every join and the register allocator spilled them)
#ifdef QN_DW_STAMPS
        long long tmg_bar = 0;
        const long long tmg_0 = __builtin_amdgcn_s_memtime();
#endif
        for (int grp = 0; grp * GC < npad; ++grp) {
            const int pp = (grp - 1) & 1;                                // ring slot of the previous group (group 0: zeros)
            for_each_stage([&](auto i_tag) {
                constexpr int i = decltype(i_tag)::value;
#ifdef QN_DW_STAMPS
                { const long long t_ = __builtin_amdgcn_s_memtime(); __syncthreads(); tmg_bar += __builtin_amdgcn_s_memtime() - t_; }
#else
                __syncthreads();                                         // chunk grp * GC + i is in LDS
#endif
                const unsigned char* pa = reinterpret_cast<const unsigned char*>(smemd + (i & 1) * DWG_BUF);
                const unsigned char* pb = pa + DW_OPER;
                v4i Af[NS], Bf[NS];
#pragma unroll
                for (int k = 0; k < NS; ++k) Af[k] = *reinterpret_cast<const v4i*>(pa + k * 4096 + m * 1024 + lofs);
#pragma unroll
                for (int k = 0; k < NS; ++k) Bf[k] = *reinterpret_cast<const v4i*>(pb + k * 4096 + lofs);
                if constexpr (i == 0) {                                  // beside the previous group's tiles 1..3
                    tile_step(integral_constant<int, 0>{}, integral_constant<int, 1>{}, Af, Bf, pb, pp);
                    tile_step(integral_constant<int, 1>{}, integral_constant<int, 2>{}, Af, Bf, pb, pp);
                    tile_step(integral_constant<int, 2>{}, integral_constant<int, 3>{}, Af, Bf, pb, pp);
                } else {
                    tile_step(integral_constant<int, 0>{}, no_tile{}, Af, Bf, pb, 0);
                    tile_step(integral_constant<int, 1>{}, no_tile{}, Af, Bf, pb, 0);
                    tile_step(integral_constant<int, 2>{}, no_tile{}, Af, Bf, pb, 0);
                }
                if constexpr (i == GC - 1) tile_step(integral_constant<int, 3>{}, integral_constant<int, 0>{}, Af, Bf, pb, grp & 1);
                else tile_step(integral_constant<int, 3>{}, no_tile{}, Af, Bf, pb, 0);
            }, std::make_integer_sequence<int, GC>{});
        }
#ifdef QN_DW_STAMPS
        if (blockIdx.x == 9 && tid == 256)
            printf("dwg matrix wave: chunks %d total %lld barrier %lld\n", npad, (long long)(__builtin_amdgcn_s_memtime() - tmg_0), tmg_bar);
#endif
        __syncthreads();                                                 // drain step
        if (nchunks > 0) {                                               // the last group's tiles 1..3
            const int pp = ((npad - 1) / GC) & 1;
#pragma unroll
            for (int t = 1; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double ts = (double)acc[t][NLEV - 1][r];
#pragma unroll
                    for (int l = NLEV - 2; l >= 0; --l) ts = fma(ts, 256.0, (double)acc[t][l][r]);
                    fl_[(4 * t + r) * 64] = fma(ts, ring[pp * 64 + 16 * m + 4 * q + r], fl_[(4 * t + r) * 64]);
                }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) O[(int64_t)(j0 + 16 * m + 4 * q + r) * g.h_in + i0 + 16 * t + c] = fl_[(4 * t + r) * 64];
    }
    __syncthreads();
    if (*badflag) {
        // exceptional values: the tile again in plain float64 (thread = 8 outputs of one row j)
        const int jl = tid >> 3, ib = (tid & 7) * 8;
        double s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sb = 0.0;
        for (int n = kbeg; n < kend; ++n) {
            const double zv = Z[(int64_t)jl * Ns + n];
            sb += zv;
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] = fma(zv, A[(int64_t)(ib + u) * Ns + n], s[u]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) O[(int64_t)(j0 + jl) * g.h_in + i0 + ib + u] = s[u];
        if (want_rowsum && ib == 0) O[(int64_t)g.h_in * g.h_out + j0 + jl] = sb;
    }
}

// The last Nb % 64 rows, which k_i8_dw_g leaves out, in plain float64, ADDED to the finished gradient block of the layer
// (G + b * p: [h_out x h_in] weights, h_out bias sums behind them): workgroup = one 64 x 64 tile of one chain, both operands' tail
// rows through LDS, thread (jl, ig) = outputs (jl, 16 ig .. 16 ig + 15).  r <= 63 rows: ~10 us at the cfg3 shape.
__global__ __launch_bounds__(256) void k_dw_tail(int h_in, int h_out, int Nb, int Ns, int n0, int has_bias, const double* __restrict__ dz,
                                                 const double* __restrict__ ap, double* __restrict__ G, int64_t p) {
    __shared__ double zt[64][65], at[64][65];
    const int b = blockIdx.z, j0 = blockIdx.y * 64, i0 = blockIdx.x * 64, r = Nb - n0, tid = threadIdx.x;
    for (int e = tid; e < 64 * r; e += 256) {
        const int f = e / r, n = e - f * r;
        zt[f][n] = dz[((int64_t)b * h_out + j0 + f) * Ns + n0 + n];
        at[f][n] = ap[((int64_t)b * h_in + i0 + f) * Ns + n0 + n];
    }
    __syncthreads();
    const int jl = tid >> 2, ig = tid & 3;
    double acc[16], sb = 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u] = 0.0;
    for (int n = 0; n < r; ++n) {
        const double zv = zt[jl][n];
        sb += zv;
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u] = fma(zv, at[16 * ig + u][n], acc[u]);
    }
    double* O = G + (int64_t)b * p;
#pragma unroll
    for (int u = 0; u < 16; ++u) O[(int64_t)(j0 + jl) * h_in + i0 + 16 * ig + u] += acc[u];
    if (has_bias && i0 == 0 && ig == 0) O[(int64_t)h_in * h_out + j0 + jl] += sb;
}

}  // namespace

// The rows k_i8_dw_g left out of a layer's weight gradient (Nb % 64 of them), added to the finished block G [B][p] + offset:
// call it AFTER the split-K reduction of the layer.  Returns QN_OK (also when there is nothing to do).
int qn_i8_dw_tail(int h_in, int h_out, int has_bias, const double* dz, const double* a_prev, int B, int Nb, int Ns, double* G, int64_t p,
                  hipStream_t st) {
    if (h_in % 64 || h_out % 64 || Ns < Nb) return QN_EUNSUPPORTED;
    const int n0 = Nb - Nb % 64;
    if (Nb < 64 || n0 == Nb) return QN_OK;                   // (fewer than 64 rows: k_i8_dw took them all)
#ifdef QN_DW_RAGGED_OLD                                      // (A/B: ragged row counts on the per-chunk kernel k_i8_dw, as until round 4)
    return QN_OK;
#endif
    hipLaunchKernelGGL(k_dw_tail, dim3(h_in / 64, h_out / 64, B), dim3(256), 0, st, h_in, h_out, Nb, Ns, n0, has_bias, dz, a_prev, G, p);
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}

// dst + b * out_stride_b + slab * out_stride_k receives [h_out x h_in] weights (+ h_out bias sums behind them): the
// conventions of k_gemm64<DW> (qn_generic.hip), whose split-K slabs and reduction kernel the caller keeps
int qn_i8_dw(int h_in, int h_out, int has_bias, const double* dz, const double* a_prev, int B, int Nb, int Ns, double* dst,
             int64_t out_stride_b, int64_t out_stride_k, int ksplit, int kchunk, const double* rowsc, hipStream_t st) {
    if (h_in % 64 || h_out % 64 || Ns < Nb) return QN_EUNSUPPORTED;
    if (rowsc && (QN_DW_GROUP < 2 || Nb < 64 || kchunk % 64)) return QN_EUNSUPPORTED;      // (row scales: the group-scale kernel only)
    DwArgs g;
    g.out_stride_b = out_stride_b; g.out_stride_k = out_stride_k; g.h_in = h_in; g.h_out = h_out; g.Nb = Nb;
    g.has_bias = has_bias; g.kchunk = kchunk; g.nmain = Nb - Nb % 64; g.Ns = Ns;
    g.inner = (h_in / 64) * (h_out / 64); g.per_b = ksplit; g.outer_total = ksplit * B;
    const unsigned grid = (unsigned)(((g.outer_total + 7) / 8) * 8 * g.inner);
    // group scales (k_i8_dw_g) for row counts in whole chunks; the per-chunk kernel (partial chunks, odd row counts) otherwise
    size_t lds = 2 * (size_t)DW_BUF + 16;
    void (*kern)(DwArgs, const double*, const double*, double*, const double*) = k_i8_dw<QN_I8_LMIN>;
    int which = 0;
#if QN_DW_GROUP >= 2
    // (whole 64-row chunks; the caller adds the last Nb % 64 rows with qn_i8_dw_tail behind its split-K reduction)
#ifdef QN_DW_RAGGED_OLD
    if (rowsc && Nb % 64) return QN_EUNSUPPORTED;
    if (Nb % 64 == 0 && kchunk % 64 == 0) {
#else
    if (Nb >= 64 && kchunk % 64 == 0) {
#endif
        lds = (size_t)DWG_LDS;
        if (rowsc) { kern = k_i8_dw_g<QN_I8_LMIN, QN_DW_GROUP, true>; which = 2; }
        else { kern = k_i8_dw_g<QN_I8_LMIN, QN_DW_GROUP, false>; which = 1; }
    }
#endif
    {   // raise the dynamic-LDS limit once per device (not per launch: the launch path stays capturable into a HIP graph)
        static std::mutex mu;
        static unsigned long long armed[3] = {0, 0, 0};                  // per kernel: bit = device ordinal
        int dev = 0;
        QN_HIP_CHECK(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lock(mu);
        if (dev < 0 || dev >= 64) return QN_EUNSUPPORTED;
        if (!(armed[which] >> dev & 1ull)) {
            QN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            armed[which] |= 1ull << dev;
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(DWT), lds, st, g, dz, a_prev, dst, rowsc);
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}
