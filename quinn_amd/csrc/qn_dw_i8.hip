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
    int inner, outer_total, per_b;                        // XCD-aware 1-D grid as gemm_grid() of qn_generic.hip
};

template <int LMIN>
__global__ __launch_bounds__(DWT, 1) void k_i8_dw(DwArgs g, const double* __restrict__ dz, const double* __restrict__ ap,
                                                 double* __restrict__ out) {
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
    const double* Z = dz + ((int64_t)b * g.h_out + j0) * Nb;
    const double* A = ap + ((int64_t)b * g.h_in + i0) * Nb;
    double* O = out + (int64_t)b * g.out_stride_b + (int64_t)slab * g.out_stride_k;
    int* badflag = reinterpret_cast<int*>(smemd + 2 * DW_BUF);
    if (tid == 0) *badflag = 0;
    const bool want_rowsum = g.has_bias && i0 == 0;

    // the 4 x 4 values of a lane's items of one operand block (items: feature 16 u + fl_; rows {2 q, 2 q + 1, 32 + 2 q, 33 + 2 q}
    // of the chunk: two 16-byte loads per item, each instruction 256 contiguous bytes per feature)
    auto load_block = [&](const double* base, int ch, int q16_, int fl_, double (&v)[4][4]) {
        const int c0 = kbeg + 64 * ch;
        if (c0 + 64 <= kend && (Nb & 1) == 0) {                          // wave-uniform: a whole chunk of 16-byte aligned rows
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double2* sp = reinterpret_cast<const double2*>(base + (int64_t)(16 * u + fl_) * Nb + c0 + 2 * q16_);
                const double2 v01 = sp[0], v23 = sp[16];
                v[u][0] = v01.x; v[u][1] = v01.y; v[u][2] = v23.x; v[u][3] = v23.y;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = c0 + 32 * (r >> 1) + 2 * q16_ + (r & 1);
                    v[u][r] = n < kend ? base[(int64_t)(16 * u + fl_) * Nb + n] : 0.0;
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
            const double zv = Z[(int64_t)jl * Nb + n];
            sb += zv;
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] = fma(zv, A[(int64_t)(ib + u) * Nb + n], s[u]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) O[(int64_t)(j0 + jl) * g.h_in + i0 + ib + u] = s[u];
        if (want_rowsum && ib == 0) O[(int64_t)g.h_in * g.h_out + j0 + jl] = sb;
    }
}

}  // namespace

// dst + b * out_stride_b + slab * out_stride_k receives [h_out x h_in] weights (+ h_out bias sums behind them): the
// conventions of k_gemm64<DW> (qn_generic.hip), whose split-K slabs and reduction kernel the caller keeps
int qn_i8_dw(int h_in, int h_out, int has_bias, const double* dz, const double* a_prev, int B, int Nb, double* dst,
             int64_t out_stride_b, int64_t out_stride_k, int ksplit, int kchunk, hipStream_t st) {
    if (h_in % 64 || h_out % 64) return QN_EUNSUPPORTED;
    DwArgs g;
    g.out_stride_b = out_stride_b; g.out_stride_k = out_stride_k; g.h_in = h_in; g.h_out = h_out; g.Nb = Nb;
    g.has_bias = has_bias; g.kchunk = kchunk;
    g.inner = (h_in / 64) * (h_out / 64); g.per_b = ksplit; g.outer_total = ksplit * B;
    const unsigned grid = (unsigned)(((g.outer_total + 7) / 8) * 8 * g.inner);
    const size_t lds = 2 * (size_t)DW_BUF + 16;
    auto kern = k_i8_dw<QN_I8_LMIN>;
    {   // raise the dynamic-LDS limit once per process (not per launch: the launch path stays capturable into a HIP graph)
        static std::mutex mu;
        static bool armed = false;
        std::lock_guard<std::mutex> lock(mu);
        if (!armed) {
            QN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            armed = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(DWT), lds, st, g, dz, a_prev, dst);
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}
