// Fused float64 forward for 128- and 256-wide tanh networks (BASELINE configs 3..5) with every hidden->hidden GEMM as
// SLICED EXACT PRODUCTS on the int8 matrix pipe -- the arithmetic of qn_fused_i8.hip (see its header) for widths whose
// digit planes no longer fit LDS (6 x 128^2 = 96 KB, 6 x 256^2 = 384 KB per layer).
//
// Replaces, for these shapes, the reference's per-sample / per-member forward loops (quinn/vi/bnet.py:202-205,
// quinn/nns/nnfit.py:133-140, quinn/mcmc/hmc.py:48-60 through quinn/nns/mlp.py:92-101) in ONE launch: first layer,
// hidden layers, last layer, residual and SSE; with `act0` set it also writes every hidden activation as float64
// [B][h][Nb] (what the layer-wise backward kernels of qn_generic.hip read), so the gradient path needs no other forward.
//
// Organisation.  Workgroup = 4 waves = one chain x 64 data rows per iteration, ONE wave per SIMD (512 registers):
//   * activations never leave the register file between layers: a wave's 16 rows x h features are h/64 x 6 digit
//     operands (v4i each); the layer's INPUT digits live in AccVGPRs and feed the MFMAs' B operand directly, the OUTPUT
//     digits are collected 64 features at a time in VGPRs and parked in AccVGPRs (2 x 96 registers at h = 256);
//   * the weight digits of a layer are consumed as a stream of TILES (16 output features x whole K: h/64 x 6 KB) that
//     the four waves fetch together by LDS-DMA into a ring of 3 tile buffers, two tiles ahead; one s_barrier per tile
//     (raw s_barrier behind an explicit s_waitcnt: the activation stores of the previous epilogue stay in flight);
//   * per tile: h/64 x 26 exact digit products into 7 int32 level accumulators, then the staged epilogue of
//     qn_fused_i8.hip (recombine, scale + bias, tanh(n/64)-table activation, digits) with the NEXT tile's MFMAs dealt
//     out between its stages.  At h = 256 a tile is 104 MFMAs = 1664 cycles of the int8 pipe against ~700 cycles of
//     vector work: the kernel is bound by the int8 matrix pipe (a float64-MFMA tile of this size: 4096 cycles + tanh).
// The digit planes come from k_i8_slice_w (once per call, qn_i8_slice.h).  A chain with an unbounded weight, or 64 rows
// with an unbounded input, take a plain float64 loop (IEEE semantics of the layer-wise kernels).
#include "qn_common.h"
#include "qn_fused_args.h"
#include "qn_math.h"
#include "qn_i8_slice.h"
#include <cstring>
#include <mutex>
#include <unordered_set>

// This file is compiled as TWO objects so that its ~40 kernel instances build in parallel: part 0 (this file) holds everything but the
// relu / identity forward instances, part 1 (qn_wide_u_i8.hip: `#define QN_WIDE_PART 1` + `#include` of this file) holds those and
// their launcher qn_i8_wide_launch_u.  -1: one object with everything.
#ifndef QN_WIDE_PART
#define QN_WIDE_PART 0
#endif
#ifndef QN_WIDE_FOLD_ALL
#define QN_WIDE_FOLD_ALL 0            // 1: the output layer's weight gradient rides on the backward kernel at h = 256 too (A/B)
#endif

namespace {

constexpr int WNBUF = 3;                    // ring of weight-tile buffers
constexpr int WWG = 256;                    // threads per workgroup

struct WideArgs {
    int64_t p, act_stride;                  // act_stride: doubles between the stashed activations of consecutive layers
    int B, Nb, d, nhid, has_bias;
    int Ns;                                 // row stride of the stash [B][h][Ns] (>= Nb: a multiple of 16 rows keeps every feature's rows on whole 128-byte lines)
    int nsplit, rows_per_split, iters;
};

// LDS, doubles first: W0 [h][DP] | b0 [h] | Wl [h] | bl, pad | red [8] | {scale, bias} (nhid-1) x [h][2] | tanh table |
// slow-path scratch 4 waves x 2 x h | then bytes: ring of WNBUF tiles [h/64][6][16 rows][64 B]
__host__ __device__ constexpr int wide_thin(int hid, int dp) { return hid * dp + hid + hid + 2 + 8; }
__host__ __device__ constexpr int wide_head(int hid, int dp, int nhid) {
    return ((wide_thin(hid, dp) + (nhid - 1) * 2 * hid + 1) & ~1) + ((TANH_TAB + 1) & ~1) + 4 * 2 * hid;
}
__host__ __device__ constexpr size_t wide_lds_bytes(int kc, int dp, int nhid) {
    return sizeof(double) * (size_t)wide_head(64 * kc, dp, nhid) + (size_t)WNBUF * kc * NS * 1024;
}

// LDS-DMA of 16 bytes per lane, issued as inline assembly ON PURPOSE: for the builtin the compiler's wait-count pass
// puts an s_waitcnt vmcnt(0) in front of the next LDS read that may alias the destination (it cannot tell the ring's
// buffers apart) -- i.e. right behind the prefetch, which then is no prefetch.  The kernel synchronises the ring itself
// (sync_tile).  `lds_addr`: byte address in LDS of the wave's 1 KB destination (wave-uniform).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"         // (m0 on the clobber list: that is the point)
// 16 bytes per lane from global memory straight into LDS (m0 = LDS address of the wave's 1 KB piece): the address is a uniform
// base (scalar register pair) + a 32-bit per-lane offset -- with a full 64-bit vector address per instruction the tile stream
// cost a dozen v_lshl_add_u64 per weight tile and wave
__device__ __forceinline__ void wglds16s(unsigned voff, const unsigned char* sbase, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
#pragma clang diagnostic pop
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
// a 128-bit value into AccVGPRs (the compiler makes the copy: v_accvgpr_write); MFMA operands read it from there
__device__ __forceinline__ v4i to_acc(v4i v) {
    v4i r;
    asm volatile("" : "=a"(r) : "0"(v));
    return r;
}

// The workgroup's rows of one iteration in plain float64 from the ORIGINAL weights (rare: unbounded weights / inputs).
// A wave takes its 16 rows one at a time, lane j owns features j, j + 64, ...; the previous layer's activations are
// broadcast through `scr` (2 x h doubles per wave).  Returns the wave's SSE share in lane 0.
__device__ __forceinline__ double wide_actf(double z, int act) {      // (torch semantics: relu(NaN) = NaN)
    return act == QN_ACT_TANH ? qn_tanh_f64(z) : act == QN_ACT_RELU ? qn_relu<double>(z) : z;
}
// (x times the activation's derivative from the activation's VALUE: qn_act_bwd of qn_math.h -- relu is the SELECT of torch's
// threshold_backward, a <= 0 ? 0 : x, so that Inf x 0 never forms and a NaN output lets the gradient pass)
__device__ __forceinline__ double wide_dmul(double x, double av, int act) { return qn_act_bwd<double>(x, av, act); }
template <int KC>
__device__ __noinline__ double wide_slow_rows(int Nb, int Ns, int d, int nhid, int has_bias, int64_t act_stride,
                                              const double* __restrict__ Wb, const double* __restrict__ X,
                                              const double* __restrict__ Y, const int32_t* __restrict__ row_idx, int nbase,
                                              int b, double* __restrict__ scr, double* __restrict__ act0,
                                              double* __restrict__ dz_last, double* __restrict__ pred_out,
                                              int actk = QN_ACT_TANH, double* __restrict__ rowsc = nullptr, int64_t rs_stride = 0) {
    constexpr int HID = 64 * KC;
    const int lane = threadIdx.x & 63, nb = has_bias ? 1 : 0;
    const int64_t gb0 = (int64_t)HID * d, gHH = gb0 + nb * HID, blk = (int64_t)HID * HID + nb * HID;
    const int64_t gWl = gHH + (int64_t)(nhid - 1) * blk, gbl = gWl + HID;
    double sse = 0.0;
    for (int n = nbase; n < nbase + 16 && n < Nb; ++n) {
        const int64_t rr = row_idx ? (int64_t)row_idx[(int64_t)b * Nb + n] : (int64_t)n;
        double act[KC];
#pragma unroll
        for (int m = 0; m < KC; ++m) {
            const int f = lane + 64 * m;
            double z = nb ? Wb[gb0 + f] : 0.0;
            for (int k = 0; k < d; ++k) z = fma(Wb[(int64_t)f * d + k], X[rr * d + k], z);
            act[m] = wide_actf(z, actk);
            if (act0) act0[((int64_t)b * HID + f) * Ns + n] = act[m];
        }
        // (relu / identity gradient calls: the row's scale 2^f > every |a| of the layer, what the fast rows leave for the
        // weight-gradient kernel; a row that is not finite gets exponent 2046 and is caught there)
        auto put_scale = [&](int layer) {
            unsigned ex = 0;
#pragma unroll
            for (int m = 0; m < KC; ++m) ex = max(ex, ((unsigned)__double2hiint(act[m]) & 0x7fffffffu) >> 20);
            ex = wave_max_u32(ex);
            ex = ex < 122u ? 122u : (ex > 2045u ? 2045u : ex);
            if (lane == 0) rowsc[layer * rs_stride + (int64_t)b * Nb + n] = __hiloint2double((int)((ex + 1) << 20), 0);
        };
        if (rowsc && nhid > 1) put_scale(0);
        for (int layer = 1; layer < nhid; ++layer) {
            double* cur = scr + HID * (layer & 1);
#pragma unroll
            for (int m = 0; m < KC; ++m) cur[lane + 64 * m] = act[m];
            __builtin_amdgcn_s_waitcnt(0xC07F);                     // lgkmcnt(0): the wave's own LDS writes have landed
            const double* Wg = Wb + gHH + (int64_t)(layer - 1) * blk;
#pragma unroll
            for (int m = 0; m < KC; ++m) {
                const int f = lane + 64 * m;
                double z = nb ? Wg[(int64_t)HID * HID + f] : 0.0;
                for (int i = 0; i < HID; ++i) z = fma(Wg[(int64_t)f * HID + i], cur[i], z);
                act[m] = wide_actf(z, actk);
                if (act0) act0[layer * act_stride + ((int64_t)b * HID + f) * Ns + n] = act[m];
            }
            if (rowsc && layer + 1 < nhid) put_scale(layer);
        }
        double pp = 0.0;
#pragma unroll
        for (int m = 0; m < KC; ++m) pp = fma(Wb[gWl + lane + 64 * m], act[m], pp);
        const double pr = wave_sum(pp) + (nb ? Wb[gbl] : 0.0);      // lane 0 holds the sum
        const double res = pr - Y[rr];
        if (lane == 0) {
            sse += res * res;
            if (pred_out) pred_out[(int64_t)b * Nb + n] = pr;
            if (dz_last) dz_last[(int64_t)b * Nb + n] = 2.0 * res;
        }
    }
    return sse;
}

template <int KC, int DP, int LMIN, bool STASH>
__global__ __launch_bounds__(WWG, 1) void k_i8_wide_fwd(WideArgs a, const double* __restrict__ W, const double* __restrict__ X,
                                                       const double* __restrict__ Y, const int32_t* __restrict__ row_idx,
                                                       const unsigned char* __restrict__ Wd, const double* __restrict__ sbg,
                                                       const int* __restrict__ flags, double* __restrict__ act0,
                                                       double* __restrict__ dz_last, double* __restrict__ pred_out,
                                                       double* __restrict__ partial, double* __restrict__ dump) {
    constexpr int HID = 64 * KC, TL = 4 * KC, TILE_B = KC * NS * 1024, PLANE = HID * HID, LAYERB = NS * PLANE;
    constexpr int NLEV = 2 * (NS - 1) - LMIN + 1, NPROD = nprod(LMIN), NPT = NPROD * KC;
    extern __shared__ __attribute__((aligned(16))) char smemw[];
    double* lds = reinterpret_cast<double*>(smemw);
    int b, split;
    if (!qn_fused_wg(a.nsplit, a.B, &b, &split)) return;
    const int NH = a.nhid, NHH = NH - 1, d = a.d, nb = a.has_bias ? 1 : 0;
    const int offb0 = HID * DP, offWl = offb0 + HID, offbl = offWl + HID, offred = offbl + 2, offsb = wide_thin(HID, DP);
    double* tanh_tab = lds + ((offsb + NHH * 2 * HID + 1) & ~1);
    double* scratch = tanh_tab + ((TANH_TAB + 1) & ~1);
    unsigned char* ring = reinterpret_cast<unsigned char*>(lds + wide_head(HID, DP, NH));
    double* red = lds + offred;
    const double* Wb = W + (int64_t)b * a.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, c = lane & 15;

    // ---- the weight-tile stream: tile (li, T) = output features 16 T .. 16 T + 15 of hidden->hidden layer li, whole K
    const unsigned char* wbase = Wd + (int64_t)b * NHH * LAYERB;       // (uniform)
    const unsigned ring_addr = lds_addr_of(ring);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned dma_off[KC * NS / 4];                                      // this lane's byte offsets inside a tile, per DMA instruction
#pragma unroll
    for (int u = 0; u < KC * NS / 4; ++u) {
        const int i = wave_u + 4 * u, kc = i / NS, wi = i - kc * NS;
        dma_off[u] = (unsigned)((lane >> 2) * HID + 16 * (lane & 3) + wi * PLANE + 64 * kc);
    }
    int pf_li = 0, pf_T = 0, pf_slot = 0;
    auto dma_next = [&]() {
        const unsigned char* src = wbase + (int64_t)pf_li * LAYERB + (int64_t)(16 * pf_T) * HID;     // (uniform)
        const unsigned dst = ring_addr + pf_slot * TILE_B;
#pragma unroll
        for (int u = 0; u < KC * NS / 4; ++u) wglds16s(dma_off[u], src, dst + (wave_u + 4 * u) * 1024);
        if (++pf_T == TL) {
            pf_T = 0;
            if (++pf_li == NHH) pf_li = 0;
        }
        pf_slot = pf_slot + 1 == WNBUF ? 0 : pf_slot + 1;
    };
    dma_next();
    dma_next();
    int rd_slot = 0;
    // The next tile has landed for every wave and every wave is done reading the tile before the current one, whose buffer
    // receives the tile after next.  s_waitcnt vmcnt(N) waits for all but the wave's N youngest vector-memory operations
    // (loads, stores and LDS-DMA count together, in issue order): N = the DMA instructions of ONE tile (the tile fetched
    // at the previous sync_tile stays in flight: a two-tile lead; vmcnt(0) here cut the lead to one tile and every tile
    // waited for its successor's DMA) + the 4 activation stores of the epilogue in between.
#ifdef QN_WIDE_STAMPS
    long long st_sync = 0, st_burst = 0, st_epi = 0, st_first = 0, st_t0 = __builtin_amdgcn_s_memtime();
#define QN_ST(var, code) { const long long t_ = __builtin_amdgcn_s_memtime(); code; var += __builtin_amdgcn_s_memtime() - t_; }
#else
#define QN_ST(var, code) { code; }
#endif
    auto sync_tile = [&]() {
        if constexpr (STASH) {
            if constexpr (KC == 4) asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(7)\n\ts_barrier" ::: "memory");
        } else {
            if constexpr (KC == 4) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory");
        }
        dma_next();
    };

    // ---- resident pieces
    qn_tanh_table64_stage(tanh_tab, tid, WWG);
    int bad = flags[b];
    {
        auto chk = [&](double v) { bad |= !qn_bounded(v); return v; };
        const int64_t gb0 = (int64_t)HID * d, gHH = gb0 + nb * HID, blk = (int64_t)HID * HID + nb * HID;
        const int64_t gWl = gHH + (int64_t)NHH * blk, gbl = gWl + HID;
        for (int e = tid; e < HID * DP; e += WWG) {
            const int j = e / DP, k = e % DP;
            lds[e] = k < d ? chk(Wb[(int64_t)j * d + k]) : 0.0;
        }
        for (int e = tid; e < HID; e += WWG) {
            lds[offb0 + e] = nb ? chk(Wb[gb0 + e]) : 0.0;
            lds[offWl + e] = chk(Wb[gWl + e]);
        }
        if (tid == 0) lds[offbl] = nb ? chk(Wb[gbl]) : 0.0;
        const double* sbs = sbg + (int64_t)b * NHH * 2 * HID;
        for (int e = tid; e < NHH * 2 * HID; e += WWG) lds[offsb + e] = sbs[e];
    }
    const bool w_bad = block_or(bad, red + 6);

    const int lofs = c * 64 + 16 * (q ^ slot_swz(c));          // this lane's 16 bytes inside one [16 rows][64 B] block
    // rounding constants as opaque register pairs, small constants through scalar registers (qn_fused_i8.hip: a known 64-bit
    // constant is re-materialised by a v_mov in front of every v_fmac_f64 that adds it; 32-bit literals make an instruction
    // 8 bytes and ~2 issue cycles dearer)
    double magic52 = 6755399441055744.0, magicS = kMagic;
    asm volatile("" : "+v"(magic52), "+v"(magicS));
    double sse = 0.0;
    double xn[DP], yn;
    int nrow_n, xbad_n;
    auto fetch = [&](int it) {
        xbad_n = 0;
        const int n = split * a.rows_per_split + (it * 4 + wave) * 16 + c;
        nrow_n = n;
        const int nn = n < a.Nb ? n : 0;
        const int64_t rr = row_idx ? (int64_t)row_idx[(int64_t)b * a.Nb + nn] : (int64_t)nn;
#pragma unroll
        for (int k = 0; k < DP; ++k) {
            xn[k] = k < d ? X[rr * d + k] : 0.0;
            xbad_n |= !qn_bounded(xn[k]);
        }
        yn = Y[rr];
    };
    fetch(0);
    for (int it = 0; it < a.iters; ++it) {
        double xk[DP];
#pragma unroll
        for (int k = 0; k < DP; ++k) xk[k] = xn[k];
        const double yk = yn;
        const int nrow = nrow_n;
        const bool live = nrow < a.Nb;
        // (workgroup-uniform: the tile barriers below need all four waves)
        const bool exceptional = block_or(w_bad | xbad_n, red + 6);
        if (it + 1 < a.iters) fetch(it + 1);
        if (exceptional) {
            sse += wide_slow_rows<KC>(a.Nb, a.Ns, d, NH, a.has_bias, a.act_stride, Wb, X, Y, row_idx,
                                      split * a.rows_per_split + (it * 4 + wave) * 16, b, scratch + 2 * HID * wave,
                                      STASH ? act0 : nullptr, dz_last, pred_out);
            // (the plain-float64 rows use flat loads / stores, which complete out of order: drain them before the counted
            // vmcnt waits of the tile stream resume)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            continue;
        }
        // stash: feature 4 q of this lane's row; rows beyond Nb write to a dump area (no branch in the epilogue: it has
        // to stay one scheduling region, and every epilogue really issues its 4 stores: sync_tile counts on them)
        const int64_t srow = ((int64_t)b * HID + 4 * q) * a.Ns + (live ? nrow : 0);
        double* const dmp = dump + lane;

        // ---- first layer (VALU): a_1 = tanh(W0 x + b0), sliced into the B operand of the first hidden layer
        v4i Bin[KC][NS];
        // Activations are sliced with the FIXED scale 2^-46 (absolute error 2^-47): a layer whose activations are ALL
        // tiny for the wave's 16 rows (top_digits_large: |a| < ~2^-4.7, e.g. very small weights without bias) would lose
        // relative accuracy -- those rows are redone in plain float64 at the end of the iteration (`redo`); otherwise
        // the error stays <= 1.8e-13 of the rows' largest activation.  Costs two instructions per tile.
        int top = 0;
#ifdef QN_WIDE_STAMPS
        const long long tf0_ = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            v4i Bcur[NS];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                double av[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 64 * kc + 16 * t + 4 * q + r;
                    double z = lds[offb0 + j];
#pragma unroll
                    for (int k = 0; k < DP; ++k) z = fma(lds[j * DP + k], xk[k], z);
                    av[r] = qn_tanh_f64_tab64(z, tanh_tab);
                    if constexpr (STASH) (live ? act0 + srow + (int64_t)(64 * kc + 16 * t) * a.Ns : dmp)[(int64_t)r * a.Ns] = av[r];
                }
                int S[NS];
                slice4(av, S);
#pragma unroll
                for (int k = 0; k < NS; ++k) Bcur[k][t] = S[k];
                top |= top_digits_large(S[NS - 1]);
            }
#pragma unroll
            for (int k = 0; k < NS; ++k) Bin[kc][k] = to_acc(Bcur[k]);
        }
        bool redo = !__any(top != 0);

#ifdef QN_WIDE_STAMPS
        st_first += __builtin_amdgcn_s_memtime() - tf0_;
#endif
        // ---- hidden -> hidden layers
        double prt = 0.0;
        auto load_frags = [&](v4i (&Af)[NS], const unsigned char* blk) {
#pragma unroll
            for (int wi = 0; wi < NS; ++wi) Af[wi] = *reinterpret_cast<const v4i*>(blk + wi * 1024);
        };
        // all of a tile's MFMAs back to back (the first tile of a layer: nothing to hide them behind)
        auto burst = [&](v4i (&acc)[NLEV], const unsigned char* tile) {
            v4i Af[2][NS];
            load_frags(Af[0], tile);
            for_each_stage([&](auto k_tag) {
                constexpr int k = decltype(k_tag)::value, kc = k / NPROD, kk = k - kc * NPROD;
                if constexpr (kk == 0 && kc + 1 < KC) load_frags(Af[(kc + 1) & 1], tile + (kc + 1) * NS * 1024);
                issue_product_c<LMIN, NLEV, kc == 0, kk>(acc, Af[kc & 1], Bin[kc]);
            }, std::make_integer_sequence<int, NPT>{});
        };
        // Epilogue of one tile as a PINNED sequence of micro-steps (one or two vector instructions of one element each,
        // stage-major over the tile's 4 elements), with the NEXT tile's MFMAs dealt out evenly between them.  With one wave
        // per SIMD nothing else fills the pipes: a burst of MFMAs stalls the wave's in-order issue for 16 cycles each
        // (vector pipe idle), a run of vector instructions leaves the matrix pipe idle -- the first version of this
        // kernel (MFMAs in bursts ahead of each stage) measured MFMA busy 40 % + VALU 34 % + waits 25 %, i.e. serial.  One
        // MFMA every <= 3 vector instructions keeps the int8 pipe busy back to back (16 cycles per MFMA >= 4 issue cycles
        // + 2..3 x 4); scheduling fences keep the compiler from regrouping.  K > 64: levels are recombined one by one
        // in float64 (pair sums would not fit int32).
        auto epilogue = [&](auto last_tag, auto next_tag, const v4i (&acc)[NLEV], v4i (&accn)[NLEV], const unsigned char* tile_next,
                            const double* sbt, const double* wlt, double* (&sp)[4], int64_t sstride, int (&S)[NS]) {
            constexpr bool LAST = decltype(last_tag)::value, NEXT = decltype(next_tag)::value;
            constexpr int NST = 11 + NLEV;                        // per-element stages (NLEV recombination stages first)
            constexpr int NMICRO = NST * 4 + 4;                   // + 4 steps: digit transposition, or the last layer's dot
            constexpr int LEAD = 3;                               // micro-steps before the first MFMA (its fragments are in flight)
            v4i Af[2][NS];
            double2 sc[4];
            double ts[4], z[4], ax[4], zm[4], Tt[4], bb[4], b2[4], pp[4], tb[4], num[4], den[4], y0[4], e0[4], av[4];
            int lo[4], hi[4], p01, q01, p23, q23, r01, r23;
            auto micro = [&](auto id_tag) {
                constexpr int id = decltype(id_tag)::value;
                if constexpr (id == 0 && NEXT) load_frags(Af[0], tile_next);
                if constexpr (NEXT && id >= LEAD) {
                    constexpr int from = ((id - LEAD) * NPT + (NMICRO - LEAD) - 1) / (NMICRO - LEAD);
                    constexpr int upto = ((id - LEAD + 1) * NPT + (NMICRO - LEAD) - 1) / (NMICRO - LEAD);
                    for_each_stage([&](auto k_tag) {
                        constexpr int k = from + decltype(k_tag)::value, kc = k / NPROD, kk = k - kc * NPROD;
                        if constexpr (kk == 0 && kc + 1 < KC) load_frags(Af[(kc + 1) & 1], tile_next + (kc + 1) * NS * 1024);
                        issue_product_c<LMIN, NLEV, kc == 0, kk>(accn, Af[kc & 1], Bin[kc]);
                    }, std::make_integer_sequence<int, upto - from>{});
                }
                if constexpr (id < NST * 4) {
                    constexpr int st = id >> 2, r = id & 3;
                    if constexpr (st == 0) {
                        sc[r] = *reinterpret_cast<const double2*>(sbt + 2 * r);
                        ts[r] = (double)acc[NLEV - 1][r];
                    } else if constexpr (st < NLEV) {
                        { const double cv_ = (double)acc[NLEV - 1 - st][r]; asm("v_fma_f64 %0, %1, %2, %3" : "=v"(ts[r]) : "v"(ts[r]), "s"(256.0), "v"(cv_)); }
                    } else if constexpr (st == NLEV) {
                        z[r] = fma(ts[r], sc[r].x, sc[r].y);
                        asm("v_min_f64 %0, |%1|, %2" : "=v"(ax[r]) : "v"(z[r]), "s"(20.0));
                    } else if constexpr (st == NLEV + 1) {
                        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(zm[r]) : "v"(ax[r]), "s"(64.0), "v"(magic52));
                        Tt[r] = tanh_tab[__double2loint(zm[r])];
                    } else if constexpr (st == NLEV + 2) {
                        { const double nf_ = zm[r] - magic52; asm("v_fma_f64 %0, %1, %2, %3" : "=v"(bb[r]) : "v"(nf_), "s"(-0.015625), "v"(ax[r])); }
                    } else if constexpr (st == NLEV + 3) {
                        b2[r] = bb[r] * bb[r];
                        pp[r] = fma(b2[r], 1.33333333333333333e-01, -3.33333333333333333e-01);
                    } else if constexpr (st == NLEV + 4) {
                        b2[r] = bb[r] * b2[r];
                        tb[r] = fma(b2[r], pp[r], bb[r]);
                    } else if constexpr (st == NLEV + 5) {
                        // (reciprocal-free tail of qn_tanh_f64_tab64: T + (1 - T^2) tb (1 - e)(1 + e^2 + e^4), e = T tb)
                        num[r] = Tt[r] * tb[r];
                        den[r] = fma(-Tt[r], Tt[r], 1.0);
                    } else if constexpr (st == NLEV + 6) {
                        e0[r] = num[r] * num[r];
                        y0[r] = fma(-tb[r], num[r], tb[r]);
                    } else if constexpr (st == NLEV + 7) {
                        e0[r] = fma(e0[r], e0[r], e0[r]);
                    } else if constexpr (st == NLEV + 8) {
                        y0[r] = fma(y0[r], e0[r], y0[r]);
                        num[r] = fma(den[r], y0[r], Tt[r]);
                    } else if constexpr (st == NLEV + 9) {
                        av[r] = __builtin_copysign(num[r], z[r]);
                        if constexpr (STASH) { *sp[r] = av[r]; sp[r] += sstride; }     // (running pointers: one 64-bit add per element)
                    } else {
                        if constexpr (!LAST) {
                            double x;
                            asm("v_fma_f64 %0, %1, %2, %3" : "=v"(x) : "v"(av[r]), "s"(0x1p46), "v"(magicS));
                            lo[r] = __double2loint(x);
                            hi[r] = __double2hiint(x);
                        }
                    }
                } else {
                    constexpr int ps = id - NST * 4;
                    if constexpr (LAST) {
                        prt = fma(wlt[ps], av[ps], prt);
                    } else if constexpr (ps == 0) {
                        p01 = __builtin_amdgcn_perm(lo[1], lo[0], 0x05010400); q01 = __builtin_amdgcn_perm(lo[1], lo[0], 0x07030602);
                        p23 = __builtin_amdgcn_perm(lo[3], lo[2], 0x05010400); q23 = __builtin_amdgcn_perm(lo[3], lo[2], 0x07030602);
                    } else if constexpr (ps == 1) {
                        r01 = __builtin_amdgcn_perm(hi[1], hi[0], 0x05010400); r23 = __builtin_amdgcn_perm(hi[3], hi[2], 0x05010400);
                        S[0] = __builtin_amdgcn_perm(p23, p01, 0x05040100) ^ 0x80808080;
                        S[1] = __builtin_amdgcn_perm(p23, p01, 0x07060302) ^ 0x80808080;
                    } else if constexpr (ps == 2) {
                        S[2] = __builtin_amdgcn_perm(q23, q01, 0x05040100) ^ 0x80808080;
                        S[3] = __builtin_amdgcn_perm(q23, q01, 0x07060302) ^ 0x80808080;
                    } else {
                        S[4] = __builtin_amdgcn_perm(r23, r01, 0x05040100) ^ 0x80808080;
                        S[5] = __builtin_amdgcn_perm(r23, r01, 0x07060302);
                    }
                }
#ifndef QN_WIDE_NOFENCE
                __builtin_amdgcn_sched_barrier(0);
#endif
            };
            for_each_stage(micro, std::make_integer_sequence<int, NMICRO>{});
        };
        auto hidden_layer = [&](auto last_tag, int li) {
            constexpr bool LAST = decltype(last_tag)::value;
            const double* sb = lds + offsb + li * 2 * HID + 2 * 4 * q;       // this lane group's features 16 T + 4 q + r
            const double* wl = lds + offWl + 4 * q;
            double* stl = STASH ? act0 + (int64_t)(li + 1) * a.act_stride + srow : nullptr;
            // the stash pointers of a tile's four elements advance by 16 features per tile (rows beyond Nb: the dump area, no
            // advance): recomputed from the layer's base for every store they cost four 64-bit vector adds per element
            double* sp[4];
            int64_t sstride = live ? (int64_t)16 * a.Ns : 0;
            asm volatile("" : "+v"(sstride));
#pragma unroll
            for (int r = 0; r < 4; ++r) sp[r] = STASH ? (live ? stl : dmp) + (int64_t)r * a.Ns : nullptr;
            v4i Bout[KC][NS];
            v4i Bcur[NS];
            v4i accA[NLEV], accB[NLEV];
            int topl = 0;
            QN_ST(st_sync, sync_tile())
            QN_ST(st_burst, burst(accA, ring + rd_slot * TILE_B + lofs))
            auto tile = [&](auto t_tag) {
                constexpr int Tt_ = decltype(t_tag)::value;
                int S[NS];
                rd_slot = rd_slot + 1 == WNBUF ? 0 : rd_slot + 1;            // (now the slot of tile Tt_ + 1)
                const unsigned char* nxt = ring + rd_slot * TILE_B + lofs;
                if constexpr (Tt_ + 1 < TL) {
                    QN_ST(st_sync, sync_tile())
                    if constexpr (Tt_ & 1) QN_ST(st_epi, epilogue(last_tag, std::true_type{}, accB, accA, nxt, sb + 32 * Tt_, wl + 16 * Tt_, sp, sstride, S))
                    else QN_ST(st_epi, epilogue(last_tag, std::true_type{}, accA, accB, nxt, sb + 32 * Tt_, wl + 16 * Tt_, sp, sstride, S))
                } else {
                    if constexpr (Tt_ & 1) QN_ST(st_epi, epilogue(last_tag, std::false_type{}, accB, accA, nxt, sb + 32 * Tt_, wl + 16 * Tt_, sp, sstride, S))
                    else QN_ST(st_epi, epilogue(last_tag, std::false_type{}, accA, accB, nxt, sb + 32 * Tt_, wl + 16 * Tt_, sp, sstride, S))
                }
                if constexpr (!LAST) {
#pragma unroll
                    for (int k = 0; k < NS; ++k) Bcur[k][Tt_ & 3] = S[k];
                    topl |= top_digits_large(S[NS - 1]);
                    if constexpr ((Tt_ & 3) == 3) {
#pragma unroll
                        for (int k = 0; k < NS; ++k) Bout[Tt_ >> 2][k] = to_acc(Bcur[k]);
                    }
                }
            };
            for_each_stage(tile, std::make_integer_sequence<int, TL>{});
            if constexpr (!LAST) {
#pragma unroll
                for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                    for (int k = 0; k < NS; ++k) Bin[kc][k] = Bout[kc][k];
                redo |= !__any(topl != 0);
            }
        };
        for (int li = 0; li < NHH - 1; ++li) hidden_layer(std::false_type{}, li);
        hidden_layer(std::true_type{}, NHH - 1);

        // ---- last layer: finish the dot over the four lane groups, residual, SSE
        double pq = prt;
        pq += __shfl_xor(pq, 16, 64);
        pq += __shfl_xor(pq, 32, 64);
        const double pr = pq + lds[offbl];
        const double res = pr - yk;
        if (redo) {                                                  // wave-uniform, rare: the wave's 16 rows again, exactly
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (the stash stores above land before their rewrites)
            sse += wide_slow_rows<KC>(a.Nb, a.Ns, d, NH, a.has_bias, a.act_stride, Wb, X, Y, row_idx,
                                      split * a.rows_per_split + (it * 4 + wave) * 16, b, scratch + 2 * HID * wave,
                                      STASH ? act0 : nullptr, dz_last, pred_out);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        } else if (live && q == 0) {
            sse += res * res;
            if (pred_out) pred_out[(int64_t)b * a.Nb + nrow] = pr;
            if (dz_last) dz_last[(int64_t)b * a.Nb + nrow] = 2.0 * res;
        }
    }
#ifdef QN_WIDE_STAMPS
    if (blockIdx.x == 17 && tid == 64)
        printf("wide fwd KC=%d stash=%d iters=%d: total %lld  sync %lld  burst %lld  epilogue %lld  first layer %lld\n", KC, (int)STASH,
               a.iters, (long long)(__builtin_amdgcn_s_memtime() - st_t0), st_sync, st_burst, st_epi, st_first);
#endif
    sse = wave_sum(sse);
    if (lane == 0) red[wave] = sse;
    __syncthreads();                                     // (vmcnt(0): the two tiles fetched ahead have landed too)
    if (tid == 0) partial[(int64_t)b * a.nsplit + split] = (red[0] + red[1]) + (red[2] + red[3]);
}

// =====================================================================================================================
// BACKWARD through the hidden layers as sliced int8 products: dZ_l = (W_l^T dZ_{l+1}) . (1 - a_l^2), l = nhid-1 .. 0,
// written as float64 [B][h][Nb] per layer (what the weight-gradient kernels read).  Replaces the layer-wise float64
// GEMMs `dzp = act'(a) . W^T dz` (k_gemm64<DA>, k_bwd_dA of qn_generic.hip), i.e. the reference's autograd through
// quinn/nns/mlp.py:92-101 (quinn/nns/nnwrap.py:128-150).
//
// Same organisation as k_i8_wide_fwd (one wave per SIMD, weight tiles streamed through the LDS ring, B operand in
// AccVGPRs), with the transposed matrices' digit planes (k_i8_slice_wT: per INPUT feature i a scale 2^e_i over the
// column W[:, i]).  The B operand are the gradients dZ[j, n], which are not bounded like tanh outputs: every data row n
// gets its own scale 2^f_n > max_j |dZ[j, n]| (exponent of the row maximum over all h features: in-lane over the 16
// tiles, then across the 4 lane groups), so a layer's outputs are kept as float64 in AccVGPRs (h/4 values per lane) until
// the last tile is done and are sliced then.  Error of an element: ~2^-47 of (column scale x row scale) per term, i.e.
// ~1e-14 of the largest contribution to the sum -- gradients agree with the float64 GEMMs to ~1e-13 of max |g| (tests).
// Chains with a weight >= 2^100 (or not finite) and 64-row iterations with an unbounded input or |dz_last| >= 2^100 take
// a plain float64 loop (no overflow can occur below those bounds for up to 8 hidden layers).
struct WideBwdArgs {
    int64_t p, act_stride, dz_stride;
    int B, Nb, d, nhid, has_bias;
    int Ns;                                 // row stride of the activation / dZ stashes [B][h][Ns]
    int nsplit, rows_per_split, iters;
};
__host__ __device__ constexpr int wideb_head(int hid, int nhid) {       // Wl [h] | red [8] | scales (nhid-1) x [h] | scratch 4 x 2 h | dWl of plain-float64 rows 4 x (h + 8)
    return ((hid + 8 + (nhid - 1) * hid + 1) & ~1) + 4 * 2 * hid + 4 * (hid + 8);
}
__host__ __device__ constexpr size_t wideb_lds_bytes(int kc, int nhid) {
    return sizeof(double) * (size_t)wideb_head(64 * kc, nhid) + (size_t)WNBUF * kc * NS * 1024;
}
__device__ __forceinline__ double to_acc_d(double v) {
    double r;
    asm volatile("" : "=a"(r) : "0"(v));
    return r;
}
__device__ __forceinline__ bool qn_bounded100(double v) {              // |v| < 2^100 (and not NaN)
    return (unsigned)(__double2hiint(v) & 0x7fffffff) < 0x46300000u;
}
__device__ __forceinline__ bool qn_bounded20(double v) {               // |v| < 2^20 (and not NaN)
    return (unsigned)(__double2hiint(v) & 0x7fffffff) < 0x41300000u;
}
// four float64 values times `scale` (a power of two that brings them into (-1, 1) x 2^46) -> six digit words
__device__ __forceinline__ void slice4s(const double (&a)[4], double scale, int (&S)[NS]) {
    int lo[4], hi[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double x = fma(a[r], scale, kMagic);
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
}

// digit planes of the TRANSPOSED hidden matrices: "row" i of plane = input feature i, K = output feature j; scale per i.
// grid (B, layers, h / 16): a workgroup takes 16 rows i (= 16 columns of W: 128-byte segments per matrix row) and all of K;
// thread (i = tid & 15, jg = tid >> 4) holds the row's entries j = 64 kc + 4 jg + r (a K-slot word per kc) in registers for
// both passes (row maximum through LDS, then digits), and the 16 rows' planes leave through LDS as whole 16-byte pieces --
// one thread per row reading a column of W serially and storing single words h bytes apart took 94 us at the cfg3 shape
// (33 MB).  flags: bit 0 = a weight that is not finite and < 2^100.
template <int LMIN>
__global__ __launch_bounds__(256) void k_i8_slice_wT(I8Net net, const double* __restrict__ W, unsigned char* __restrict__ Wd,
                                                    double* __restrict__ sc, int* __restrict__ flags) {
    constexpr int KCM = 4;                                       // h <= 256
    __shared__ unsigned exs[16][17];
    __shared__ __attribute__((aligned(16))) unsigned char stage[NS][16][64 * KCM];
    const int b = blockIdx.x, li = blockIdx.y, tid = threadIdx.x;
    const int h = net.h[li], nkc = h >> 6;                       // square hidden matrices, h in {128, 256}
    const double* Wg = W + (int64_t)b * net.p + net.offW[li];
    unsigned char* planes = Wd + (int64_t)b * net.dbytes + net.offD[li];
    double* scl = sc + (int64_t)b * net.sdoubles + net.offS[li];
    const int64_t plane = (int64_t)h * h;
    const int il = tid & 15, jg = tid >> 4, i = blockIdx.z * 16 + il;
    int bad = 0;
    double v[KCM][4];
    unsigned ex = 0;
#pragma unroll
    for (int kc = 0; kc < KCM; ++kc)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[kc][r] = kc < nkc ? Wg[(int64_t)(64 * kc + 4 * jg + r) * h + i] : 0.0;
            bad |= !qn_bounded100(v[kc][r]);
            ex = max(ex, ((unsigned)__double2hiint(v[kc][r]) & 0x7fffffffu) >> 20);
        }
    exs[il][jg] = ex;
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 16; ++g) ex = max(ex, exs[il][g]);
    int e = (int)ex - 1022;
    bad |= e > I8_MAX_WEIGHT_EXP;
    e = e < -900 ? -900 : e;
    const double dn = ldexp(1.0, -e);
    const int m = jg >> 2, g4 = jg & 3;
#pragma unroll
    for (int kc = 0; kc < KCM; ++kc) {
        if (kc < nkc) {
            double an[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) an[r] = v[kc][r] * dn;      // exact
            int S[NS];
            slice4(an, S);
#pragma unroll
            for (int k = 0; k < NS; ++k)
                *reinterpret_cast<int*>(&stage[k][il][64 * kc + 16 * (g4 ^ slot_swz(i)) + 4 * m]) = S[k];
        }
    }
    if (jg == 0) scl[i] = ldexp(1.0, e - 2 * QB + 8 * LMIN);
    __syncthreads();
    // plane k, rows 16 z .. 16 z + 15: 16 h contiguous bytes
    const int per_plane = 16 * h / 16;                           // 16-byte pieces
    for (int pc = tid; pc < NS * per_plane; pc += 256) {
        const int k = pc / per_plane, o = (pc - k * per_plane) * 16, row = o / h, col = o - row * h;
        *reinterpret_cast<uint4*>(planes + k * plane + (int64_t)(blockIdx.z * 16 + row) * h + col) =
            *reinterpret_cast<const uint4*>(&stage[k][row][col]);
    }
    if (net.has_bias && blockIdx.z == 0)
        for (int t = tid; t < h; t += 256) bad |= !qn_bounded100(W[(int64_t)b * net.p + net.offB[li] + t]);
    if (__any(bad) && (tid & 63) == 0) atomicOr(&flags[b], 1);
}

// the workgroup's rows of one iteration in plain float64 (rare); lane j owns features j, j + 64, ...
template <int KC>
__device__ __noinline__ void wide_slow_bwd_rows(int Nb, int Ns, int d, int nhid, int has_bias, int64_t act_stride, int64_t dz_stride,
                                                const double* __restrict__ Wb, int nbase, int b, double* __restrict__ scr,
                                                const double* __restrict__ act0, const double* __restrict__ dz_last,
                                                double* __restrict__ dz0, double* __restrict__ dwl_acc, int actk = QN_ACT_TANH) {
    constexpr int HID = 64 * KC;
    const int lane = threadIdx.x & 63, nb = has_bias ? 1 : 0;
    const int64_t gHH = (int64_t)HID * d + nb * HID, blk = (int64_t)HID * HID + nb * HID, gWl = gHH + (int64_t)(nhid - 1) * blk;
    for (int n = nbase; n < nbase + 16 && n < Nb; ++n) {
        const double dzl = dz_last[(int64_t)b * Nb + n];
        double g[KC];
        if (dwl_acc && lane == 0) dwl_acc[HID] += dzl;                 // (this wave's LDS slots: lane j owns entries j, j + 64, ...)
#pragma unroll
        for (int m = 0; m < KC; ++m) {
            const int64_t idx = ((int64_t)b * HID + lane + 64 * m) * Ns + n;
            const double av = act0[(nhid - 1) * act_stride + idx];
            g[m] = wide_dmul(Wb[gWl + lane + 64 * m] * dzl, av, actk);
            dz0[(nhid - 1) * dz_stride + idx] = g[m];
            if (dwl_acc) dwl_acc[lane + 64 * m] = fma(dzl, av, dwl_acc[lane + 64 * m]);
        }
        for (int li = nhid - 2; li >= 0; --li) {
            double* cur = scr + HID * (li & 1);
#pragma unroll
            for (int m = 0; m < KC; ++m) cur[lane + 64 * m] = g[m];
            __builtin_amdgcn_s_waitcnt(0xC07F);
            const double* Wg = Wb + gHH + (int64_t)li * blk;
#pragma unroll
            for (int m = 0; m < KC; ++m) {
                const int i = lane + 64 * m;
                double acc = 0.0;
                for (int j = 0; j < HID; ++j) acc = fma(Wg[(int64_t)j * HID + i], cur[j], acc);
                const int64_t idx = ((int64_t)b * HID + i) * Ns + n;
                const double av = act0[li * act_stride + idx];
                g[m] = wide_dmul(acc, av, actk);
                dz0[li * dz_stride + idx] = g[m];
            }
        }
    }
}

// TANH = false (round 4): relu / identity networks (`actk`, uniform) -- the derivative is a select on the stashed activation.
template <int KC, int DP, int LMIN, bool TANH = true>
__global__ __launch_bounds__(WWG, 1) void k_i8_wide_bwd(WideBwdArgs a, const double* __restrict__ W, const double* __restrict__ X,
                                                       const int32_t* __restrict__ row_idx, const unsigned char* __restrict__ WdT,
                                                       const double* __restrict__ scT, const int* __restrict__ flags,
                                                       const double* __restrict__ act0, const double* __restrict__ dz_last,
                                                       double* __restrict__ dz0, double* __restrict__ dump, double* __restrict__ dwl_out,
                                                       int actk) {
    // derivative of the activation from its value: tanh 1 - a^2; relu (a > 0); identity 1 (dsel: 0 for relu, -inf for identity)
    const double dsel = actk == QN_ACT_RELU ? 0.0 : -__builtin_inf();
    auto dact = [&](double av) { return TANH ? fma(-av, av, 1.0) : (av > dsel ? 1.0 : 0.0); };
    constexpr int HID = 64 * KC, TL = 4 * KC, TILE_B = KC * NS * 1024, PLANE = HID * HID, LAYERB = NS * PLANE;
    constexpr int NLEV = 2 * (NS - 1) - LMIN + 1, NPROD = nprod(LMIN), NPT = NPROD * KC;
    extern __shared__ __attribute__((aligned(16))) char smemb[];
    double* lds = reinterpret_cast<double*>(smemb);
    int b, split;
    if (!qn_fused_wg(a.nsplit, a.B, &b, &split)) return;
    const int NH = a.nhid, NHH = NH - 1, d = a.d, nb = a.has_bias ? 1 : 0;
    const int offred = HID, offsc = HID + 8;
    double* scratch = lds + ((offsc + NHH * HID + 1) & ~1);
    // The output layer's weight gradient dWl[f] = sum_n dz_last[n] a[f][n], dbl = sum_n dz_last[n] rides on the top phase,
    // which has a[f][n] and dz_last[n] in registers and waits for memory (the separate kernel read the activations -- 1 GB at
    // the cfg3 shape -- once more): at h = 128, where 32 more accumulators per lane fit.  dwl_out: [B][nsplit][HID + 1].
    constexpr bool FOLD = QN_WIDE_FOLD_ALL || KC == 2;
    double* slow_acc = scratch + 4 * 2 * HID;                           // [4 waves][HID + 8]: rows that took the plain-float64 path
    unsigned char* ring = reinterpret_cast<unsigned char*>(lds + wideb_head(HID, NH));
    double* red = lds + offred;
    const double* Wb = W + (int64_t)b * a.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, c = lane & 15;

    // ---- weight-tile stream: layers from the top down (li = NHH-1 .. 0), tiles T = 16 input features of layer li
    const unsigned char* wbase = WdT + (int64_t)b * NHH * LAYERB;       // (uniform)
    const unsigned ring_addr = lds_addr_of(ring);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned dma_off[KC * NS / 4];                                      // this lane's byte offsets inside a tile, per DMA instruction
#pragma unroll
    for (int u = 0; u < KC * NS / 4; ++u) {
        const int i = wave_u + 4 * u, kc = i / NS, wi = i - kc * NS;
        dma_off[u] = (unsigned)((lane >> 2) * HID + 16 * (lane & 3) + wi * PLANE + 64 * kc);
    }
    int pf_li = NHH - 1, pf_T = 0, pf_slot = 0;
    auto dma_next = [&]() {
        const unsigned char* src = wbase + (int64_t)pf_li * LAYERB + (int64_t)(16 * pf_T) * HID;     // (uniform)
        const unsigned dst = ring_addr + pf_slot * TILE_B;
#pragma unroll
        for (int u = 0; u < KC * NS / 4; ++u) wglds16s(dma_off[u], src, dst + (wave_u + 4 * u) * 1024);
        if (++pf_T == TL) {
            pf_T = 0;
            if (--pf_li < 0) pf_li = NHH - 1;
        }
        pf_slot = pf_slot + 1 == WNBUF ? 0 : pf_slot + 1;
    };
    dma_next();
    dma_next();
    int rd_slot = 0;
#ifdef QN_WIDE_STAMPS
    long long st_sync = 0, st_burst = 0, st_epi = 0, st_top = 0, st_slice = 0, st_t0 = __builtin_amdgcn_s_memtime();
#endif
    // (as in k_i8_wide_fwd: the youngest tile DMA + one epilogue's 4 loads and 4 stores may stay in flight)
    auto sync_tile = [&]() {
        if constexpr (KC == 4) asm volatile("s_waitcnt vmcnt(14)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(11)\n\ts_barrier" ::: "memory");
        dma_next();
    };

    int bad = flags[b];
    {
        const int64_t gHH = (int64_t)HID * d + nb * HID, blk = (int64_t)HID * HID + nb * HID, gWl = gHH + (int64_t)NHH * blk;
        for (int e = tid; e < HID; e += WWG) {
            const double v = Wb[gWl + e];
            bad |= !qn_bounded100(v);
            lds[e] = v;
        }
        for (int e = tid; e < HID * d; e += WWG) bad |= !qn_bounded100(Wb[e]);          // (first layer: part of the chain's bound)
        const double* scs = scT + (int64_t)b * NHH * HID;
        for (int e = tid; e < NHH * HID; e += WWG) lds[offsc + e] = scs[e];
    }
    double dwl[FOLD ? TL : 1][4];
    double dbl = 0.0;
    if constexpr (FOLD) {
#pragma unroll
        for (int t = 0; t < TL; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) dwl[t][r] = 0.0;
        for (int e = tid; e < 4 * (HID + 8); e += WWG) slow_acc[e] = 0.0;
    }
    const bool fold = FOLD && dwl_out != nullptr;
    const bool w_bad = block_or(bad, red + 6);

    const int lofs = c * 64 + 16 * (q ^ slot_swz(c));
    for (int it = 0; it < a.iters; ++it) {
        const int nrow = split * a.rows_per_split + (it * 4 + wave) * 16 + c;
        const bool live = nrow < a.Nb;
        const int nn = live ? nrow : 0;
        const int64_t rr = row_idx ? (int64_t)row_idx[(int64_t)b * a.Nb + nn] : (int64_t)nn;
        const double dzl = dz_last[(int64_t)b * a.Nb + nn];
        int xbad = !qn_bounded100(dzl);
        for (int k = 0; k < d; ++k) xbad |= !qn_bounded(X[rr * d + k]);      // (run-time d: the instances for 5..8 inputs are the DP = 4 ones)
        const bool exceptional = block_or(w_bad | xbad, red + 6);
        if (exceptional) {
            wide_slow_bwd_rows<KC>(a.Nb, a.Ns, d, NH, a.has_bias, a.act_stride, a.dz_stride, Wb,
                                   split * a.rows_per_split + (it * 4 + wave) * 16, b, scratch + 2 * HID * wave, act0, dz_last, dz0,
                                   fold ? slow_acc + (HID + 8) * wave : nullptr, TANH ? QN_ACT_TANH : actk);
            // (flat loads / stores in there complete out of order: drain them before the counted vmcnt waits resume)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            continue;
        }
        // element (feature 4 q [+ 16 T + r], this lane's row); rows beyond Nb read row 0 and write to the dump area
        const int64_t erow = ((int64_t)b * HID + 4 * q) * a.Ns + nn;
        double* const dmp = dump + lane;

        // ---- top: dZ_{nhid-1} = (1 - a^2) wl dz_last (VALU), kept as float64 until the row maximum is known
        double V[TL][4];
        double amax = 0.0;
#ifdef QN_WIDE_STAMPS
        const long long tt0_ = __builtin_amdgcn_s_memtime();
#endif
        {
            // (loads in batches of 16, the next batch issued ahead of the current one's arithmetic and stores: the stores
            // may alias the loads as far as the compiler knows, and one load -> store at a time is one HBM round trip each:
            // measured 33 % / 53 % of the kernel at h = 256 / 128 before)
            const double* ap = act0 + (int64_t)(NH - 1) * a.act_stride + erow;
            double* zp = dz0 + (int64_t)(NH - 1) * a.dz_stride + erow;
            double abuf[2][16];
            const double dze = live ? dzl : 0.0;                          // (rows beyond Nb read row 0: no contribution)
            if (q == 0) dbl += dze;
            auto load_batch = [&](int bi, double (&dst)[16]) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[4 * t + r] = ap[(int64_t)(16 * (4 * bi + t) + r) * a.Ns];
            };
            load_batch(0, abuf[0]);
#pragma unroll
            for (int bi = 0; bi < KC; ++bi) {
                if (bi + 1 < KC) load_batch(bi + 1, abuf[(bi + 1) & 1]);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int T_ = 4 * bi + t;
                        const double av = abuf[bi & 1][4 * t + r];
                        if constexpr (FOLD) dwl[T_][r] = fma(dze, av, dwl[T_][r]);
                        const double v = (lds[16 * T_ + 4 * q + r] * dzl) * dact(av);
                        (live ? zp + (int64_t)(16 * T_) * a.Ns : dmp)[(int64_t)r * a.Ns] = v;
                        amax = fmax(amax, fabs(v));
                        V[T_][r] = to_acc_d(v);
                    }
            }
        }
#ifdef QN_WIDE_STAMPS
        st_top += __builtin_amdgcn_s_memtime() - tt0_;
#endif
        v4i Bin[KC][NS];
        double rs = 0.0;                                           // 2^f_n: this row's scale of the current B operand
        // row maximum over the 4 lane groups -> exponent -> digits of the whole row set
        auto slice_rows = [&]() {
            double m = amax;
            m = fmax(m, __shfl_xor(m, 16, 64));
            m = fmax(m, __shfl_xor(m, 32, 64));
            int E = (__double2hiint(m) >> 20) & 0x7ff;             // |v| < 2^(E - 1022) for every v of the row
            E = E < 122 ? 122 : E;
            const double sl = __hiloint2double((2091 - E) << 20, 0);        // 2^(46 - f), f = E - 1022
            rs = __hiloint2double((E + 1) << 20, 0);                        // 2^f
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                v4i Bcur[NS];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    double vv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) vv[r] = V[4 * kc + t][r];
                    int S[NS];
                    slice4s(vv, sl, S);
#pragma unroll
                    for (int k = 0; k < NS; ++k) Bcur[k][t] = S[k];
                }
#pragma unroll
                for (int k = 0; k < NS; ++k) Bin[kc][k] = to_acc(Bcur[k]);
            }
            amax = 0.0;
        };
        QN_ST(st_slice, slice_rows())

        auto load_frags = [&](v4i (&Af)[NS], const unsigned char* blk) {
#pragma unroll
            for (int wi = 0; wi < NS; ++wi) Af[wi] = *reinterpret_cast<const v4i*>(blk + wi * 1024);
        };
        auto burst = [&](v4i (&acc)[NLEV], const unsigned char* tile) {
            v4i Af[2][NS];
            load_frags(Af[0], tile);
            for_each_stage([&](auto k_tag) {
                constexpr int k = decltype(k_tag)::value, kc = k / NPROD, kk = k - kc * NPROD;
                if constexpr (kk == 0 && kc + 1 < KC) load_frags(Af[(kc + 1) & 1], tile + (kc + 1) * NS * 1024);
                issue_product_c<LMIN, NLEV, kc == 0, kk>(acc, Af[kc & 1], Bin[kc]);
            }, std::make_integer_sequence<int, NPT>{});
        };
        for (int li = NHH - 1; li >= 0; --li) {
            const double* sct = lds + offsc + li * HID + 4 * q;
            const double* apl = act0 + (int64_t)li * a.act_stride + erow;
            double* zpl = dz0 + (int64_t)li * a.dz_stride + erow;
            // a_l of the tile's 4 elements, fetched TWO tiles ahead (an HBM round trip is longer than one tile's epilogue)
            double ab[3][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ab[0][r] = apl[(int64_t)r * a.Ns];
                ab[1][r] = apl[(int64_t)(16 + r) * a.Ns];
            }
            // running pointers of a tile's four elements (activation loads two tiles ahead, dZ stores): 16 features per
            // tile; recomputed from the layer's base for every access they cost four to five 64-bit vector adds per element.
            // Rows beyond Nb store to the dump area and do not advance.
            const double* lp[4];
            double* sp[4];
            int64_t lstride = (int64_t)16 * a.Ns, sstride = live ? (int64_t)16 * a.Ns : 0;
            asm volatile("" : "+v"(lstride), "+v"(sstride));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                lp[r] = apl + (int64_t)((TL > 2 ? 32 : 16 * (TL - 1)) + r) * a.Ns;
                sp[r] = (live ? zpl : dmp) + (int64_t)r * a.Ns;
            }
            v4i accA[NLEV], accB[NLEV];
            int topl = 0;
            QN_ST(st_sync, sync_tile())
            QN_ST(st_burst, burst(accA, ring + rd_slot * TILE_B + lofs))
            // epilogue of tile T (pinned micro-steps, the next tile's MFMAs dealt out between them)
            auto epilogue = [&](auto next_tag, auto t_tag, const v4i (&acc)[NLEV], v4i (&accn)[NLEV], const unsigned char* tile_next) {
                constexpr bool NEXT = decltype(next_tag)::value;
                constexpr int Tt_ = decltype(t_tag)::value;
                constexpr int NST = NLEV + 3, NMICRO = NST * 4, LEAD = 2;
                v4i Af[2][NS];
                double ts[4], g[4], v[4];
                const double (&ac)[4] = ab[Tt_ % 3];
                // (always 4 loads per epilogue: sync_tile counts on them; past the last tile they re-read the last one)
                {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ab[(Tt_ + 2) % 3][r] = *lp[r];
                        if constexpr (Tt_ + 3 < TL) lp[r] += lstride;
                    }
                }
                auto micro = [&](auto id_tag) {
                    constexpr int id = decltype(id_tag)::value;
                    if constexpr (id == 0 && NEXT) load_frags(Af[0], tile_next);
                    if constexpr (NEXT && id >= LEAD) {
                        constexpr int from = ((id - LEAD) * NPT + (NMICRO - LEAD) - 1) / (NMICRO - LEAD);
                        constexpr int upto = ((id - LEAD + 1) * NPT + (NMICRO - LEAD) - 1) / (NMICRO - LEAD);
                        for_each_stage([&](auto k_tag) {
                            constexpr int k = from + decltype(k_tag)::value, kc = k / NPROD, kk = k - kc * NPROD;
                            if constexpr (kk == 0 && kc + 1 < KC) load_frags(Af[(kc + 1) & 1], tile_next + (kc + 1) * NS * 1024);
                            issue_product_c<LMIN, NLEV, kc == 0, kk>(accn, Af[kc & 1], Bin[kc]);
                        }, std::make_integer_sequence<int, upto - from>{});
                    }
                    constexpr int st = id >> 2, r = id & 3;
                    if constexpr (st == 0) {
                        ts[r] = (double)acc[NLEV - 1][r];
                    } else if constexpr (st < NLEV) {
                        { const double cv_ = (double)acc[NLEV - 1 - st][r]; asm("v_fma_f64 %0, %1, %2, %3" : "=v"(ts[r]) : "v"(ts[r]), "s"(256.0), "v"(cv_)); }
                    } else if constexpr (st == NLEV) {
                        g[r] = dact(ac[r]) * rs;
                    } else if constexpr (st == NLEV + 1) {
                        v[r] = (ts[r] * sct[16 * Tt_ + r]) * g[r];
                    } else {
                        *sp[r] = v[r];
                        sp[r] += sstride;
                        amax = fmax(amax, fabs(v[r]));
                        V[Tt_][r] = to_acc_d(v[r]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                for_each_stage(micro, std::make_integer_sequence<int, NMICRO>{});
            };
            auto tile = [&](auto t_tag) {
                constexpr int Tt_ = decltype(t_tag)::value;
                rd_slot = rd_slot + 1 == WNBUF ? 0 : rd_slot + 1;
                const unsigned char* nxt = ring + rd_slot * TILE_B + lofs;
                if constexpr (Tt_ + 1 < TL) {
                    QN_ST(st_sync, sync_tile())
                    if constexpr (Tt_ & 1) QN_ST(st_epi, epilogue(std::true_type{}, t_tag, accB, accA, nxt))
                    else QN_ST(st_epi, epilogue(std::true_type{}, t_tag, accA, accB, nxt))
                } else {
                    if constexpr (Tt_ & 1) QN_ST(st_epi, epilogue(std::false_type{}, t_tag, accB, accA, nxt))
                    else QN_ST(st_epi, epilogue(std::false_type{}, t_tag, accA, accB, nxt))
                }
            };
            for_each_stage(tile, std::make_integer_sequence<int, TL>{});
            if (li > 0) QN_ST(st_slice, slice_rows())
        }
    }
#ifdef QN_WIDE_STAMPS
    if (blockIdx.x == 17 && tid == 64)
        printf("wide bwd KC=%d iters=%d: total %lld  sync %lld  burst %lld  epilogue %lld  top %lld  slice %lld\n", KC, a.iters,
               (long long)(__builtin_amdgcn_s_memtime() - st_t0), st_sync, st_burst, st_epi, st_top, st_slice);
#endif
    __syncthreads();                                     // (vmcnt(0): the tiles fetched ahead have landed before the LDS is released)
    if constexpr (FOLD) {
        if (fold) {
            // sum over the 16 lanes (rows) that hold the same features, one slot per wave in the (now idle) scratch area,
            // then the four waves and the plain-float64 rows in a fixed order
            double* part = scratch + 2 * HID * wave;
#pragma unroll
            for (int t = 0; t < TL; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double v = dwl[t][r];
                    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                    if (c == 0) part[16 * t + 4 * q + r] = v;
                }
            double vb = dbl;                                      // (nonzero on the lanes q = 0 only)
            vb += __shfl_xor(vb, 1, 64); vb += __shfl_xor(vb, 2, 64); vb += __shfl_xor(vb, 4, 64); vb += __shfl_xor(vb, 8, 64);
            if (lane == 0) part[HID] = vb;
            __syncthreads();
            for (int e = tid; e < HID + 1; e += WWG) {
                double sv = (scratch[e] + scratch[2 * HID + e]) + (scratch[4 * HID + e] + scratch[6 * HID + e]);
                sv += (slow_acc[e] + slow_acc[(HID + 8) + e]) + (slow_acc[2 * (HID + 8) + e] + slow_acc[3 * (HID + 8) + e]);
                dwl_out[((int64_t)b * a.nsplit + split) * (HID + 1) + e] = sv;
            }
        }
    }
}

// =====================================================================================================================
// FORWARD for relu / identity networks (round 4; the reference's DEFAULT activation is relu, quinn/nns/mlp.py:23).
// Their activations are not bounded by 1, so -- as for the gradients dZ of k_i8_wide_bwd, whose organisation this kernel
// shares -- every data row n gets its own scale 2^f_n > max_j |a_l[j, n]| per layer: a layer's outputs stay float64 in
// AccVGPRs (h/4 values per lane) until its last tile is done (row maximum in-lane over the tiles, then across the 4 lane
// groups), are sliced then, and the next layer's integer sums are multiplied by 2^f_n.  No tanh table, no tiny-activation
// rule (the row maximum itself sets the scale).  Chains with a weight or bias >= 2^20 and rows with an input >= 2^100 take the
// plain float64 loop: below those bounds no product or sum overflows for up to 15 layers (2^(100 + 15 x 28)).
template <int KC, int DP, int LMIN, bool STASH>
__global__ __launch_bounds__(WWG, 1) void k_i8_wide_fwd_u(WideArgs a, const double* __restrict__ W, const double* __restrict__ X,
                                                         const double* __restrict__ Y, const int32_t* __restrict__ row_idx,
                                                         const unsigned char* __restrict__ Wd, const double* __restrict__ sbg,
                                                         const int* __restrict__ flags, double* __restrict__ act0,
                                                         double* __restrict__ dz_last, double* __restrict__ pred_out,
                                                         double* __restrict__ partial, double* __restrict__ dump, int actk,
                                                         double* __restrict__ rowsc, int64_t rs_stride) {
    constexpr int HID = 64 * KC, TL = 4 * KC, TILE_B = KC * NS * 1024, PLANE = HID * HID, LAYERB = NS * PLANE;
    constexpr int NLEV = 2 * (NS - 1) - LMIN + 1, NPROD = nprod(LMIN), NPT = NPROD * KC;
    extern __shared__ __attribute__((aligned(16))) char smemu[];
    double* lds = reinterpret_cast<double*>(smemu);
    int b, split;
    if (!qn_fused_wg(a.nsplit, a.B, &b, &split)) return;
    const int NH = a.nhid, NHH = NH - 1, d = a.d, nb = a.has_bias ? 1 : 0;
    // (the LDS layout of k_i8_wide_fwd; its tanh-table area stays unused)
    const int offb0 = HID * DP, offWl = offb0 + HID, offbl = offWl + HID, offred = offbl + 2, offsb = wide_thin(HID, DP);
    double* scratch = lds + ((offsb + NHH * 2 * HID + 1) & ~1) + ((TANH_TAB + 1) & ~1);
    unsigned char* ring = reinterpret_cast<unsigned char*>(lds + wide_head(HID, DP, NH));
    double* red = lds + offred;
    const double* Wb = W + (int64_t)b * a.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, c = lane & 15;
    const double alo = actk == QN_ACT_RELU ? 0.0 : -__builtin_inf();    // a = max(z, alo): relu / identity (z is finite here)

    // ---- the weight-tile stream (as in k_i8_wide_fwd)
    const unsigned char* wbase = Wd + (int64_t)b * NHH * LAYERB;
    const unsigned ring_addr = lds_addr_of(ring);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned dma_off[KC * NS / 4];
#pragma unroll
    for (int u = 0; u < KC * NS / 4; ++u) {
        const int i = wave_u + 4 * u, kc = i / NS, wi = i - kc * NS;
        dma_off[u] = (unsigned)((lane >> 2) * HID + 16 * (lane & 3) + wi * PLANE + 64 * kc);
    }
    int pf_li = 0, pf_T = 0, pf_slot = 0;
    auto dma_next = [&]() {
        const unsigned char* src = wbase + (int64_t)pf_li * LAYERB + (int64_t)(16 * pf_T) * HID;
        const unsigned dst = ring_addr + pf_slot * TILE_B;
#pragma unroll
        for (int u = 0; u < KC * NS / 4; ++u) wglds16s(dma_off[u], src, dst + (wave_u + 4 * u) * 1024);
        if (++pf_T == TL) {
            pf_T = 0;
            if (++pf_li == NHH) pf_li = 0;
        }
        pf_slot = pf_slot + 1 == WNBUF ? 0 : pf_slot + 1;
    };
    dma_next();
    dma_next();
    int rd_slot = 0;
    // (the youngest tile's DMA instructions + one epilogue's 4 activation stores may stay in flight: see k_i8_wide_fwd)
    auto sync_tile = [&]() {
        if constexpr (STASH) {
            if constexpr (KC == 4) asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(7)\n\ts_barrier" ::: "memory");
        } else {
            if constexpr (KC == 4) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory");
        }
        dma_next();
    };

    // ---- resident pieces
    int bad = flags[b];
    {
        auto chk = [&](double v) { bad |= !qn_bounded20(v); return v; };
        const int64_t gb0 = (int64_t)HID * d, gHH = gb0 + nb * HID, blk = (int64_t)HID * HID + nb * HID;
        const int64_t gWl = gHH + (int64_t)NHH * blk, gbl = gWl + HID;
        for (int e = tid; e < HID * DP; e += WWG) {
            const int j = e / DP, k = e % DP;
            lds[e] = k < d ? chk(Wb[(int64_t)j * d + k]) : 0.0;
        }
        for (int e = tid; e < HID; e += WWG) {
            lds[offb0 + e] = nb ? chk(Wb[gb0 + e]) : 0.0;
            lds[offWl + e] = chk(Wb[gWl + e]);
        }
        if (tid == 0) lds[offbl] = nb ? chk(Wb[gbl]) : 0.0;
        const double* sbs = sbg + (int64_t)b * NHH * 2 * HID;
        for (int e = tid; e < NHH * 2 * HID; e += WWG) {
            const double v = sbs[e];
            if (e & 1) bad |= !qn_bounded20(v);                        // (the hidden layers' biases; their weights: k_i8_slice_w)
            lds[offsb + e] = v;
        }
    }
    const bool w_bad = block_or(bad, red + 6);

    const int lofs = c * 64 + 16 * (q ^ slot_swz(c));
    double sse = 0.0;
    double xn[DP], yn;
    int nrow_n, xbad_n;
    auto fetch = [&](int it) {
        xbad_n = 0;
        const int n = split * a.rows_per_split + (it * 4 + wave) * 16 + c;
        nrow_n = n;
        const int nn = n < a.Nb ? n : 0;
        const int64_t rr = row_idx ? (int64_t)row_idx[(int64_t)b * a.Nb + nn] : (int64_t)nn;
#pragma unroll
        for (int k = 0; k < DP; ++k) {
            xn[k] = k < d ? X[rr * d + k] : 0.0;
            xbad_n |= !qn_bounded100(xn[k]);
        }
        yn = Y[rr];
    };
    fetch(0);
    for (int it = 0; it < a.iters; ++it) {
        double xk[DP];
#pragma unroll
        for (int k = 0; k < DP; ++k) xk[k] = xn[k];
        const double yk = yn;
        const int nrow = nrow_n;
        const bool live = nrow < a.Nb;
        const bool exceptional = block_or(w_bad | xbad_n, red + 6);     // (workgroup-uniform: the tile barriers need all four waves)
        if (it + 1 < a.iters) fetch(it + 1);
        if (exceptional) {
            sse += wide_slow_rows<KC>(a.Nb, a.Ns, d, NH, a.has_bias, a.act_stride, Wb, X, Y, row_idx,
                                      split * a.rows_per_split + (it * 4 + wave) * 16, b, scratch + 2 * HID * wave,
                                      STASH ? act0 : nullptr, dz_last, pred_out, actk, STASH ? rowsc : nullptr, rs_stride);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            continue;
        }
        const int64_t srow = ((int64_t)b * HID + 4 * q) * a.Ns + (live ? nrow : 0);
        double* const dmp = dump + lane;

        // ---- first layer (VALU): a_1 = act(W0 x + b0), kept as float64 until the row maximum is known
        double V[TL][4];
        double amax = 0.0;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 64 * kc + 16 * t + 4 * q + r;
                    double z = lds[offb0 + j];
#pragma unroll
                    for (int k = 0; k < DP; ++k) z = fma(lds[j * DP + k], xk[k], z);
                    const double av = fmax(z, alo);
                    if constexpr (STASH) (live ? act0 + srow + (int64_t)(64 * kc + 16 * t) * a.Ns : dmp)[(int64_t)r * a.Ns] = av;
                    amax = fmax(amax, fabs(av));
                    V[4 * kc + t][r] = to_acc_d(av);
                }
        v4i Bin[KC][NS];
        double rs = 0.0;                                           // 2^f_n: this row's scale of the current B operand
        auto slice_rows = [&](int layer) {                         // (as in k_i8_wide_bwd)
            double m = amax;
            m = fmax(m, __shfl_xor(m, 16, 64));
            m = fmax(m, __shfl_xor(m, 32, 64));
            int E = (__double2hiint(m) >> 20) & 0x7ff;             // |v| < 2^(E - 1022) for every v of the row
            E = E < 122 ? 122 : E;
            const double sl = __hiloint2double((2091 - E) << 20, 0);        // 2^(46 - f), f = E - 1022
            rs = __hiloint2double((E + 1) << 20, 0);                        // 2^f
            // gradient calls: the weight-gradient kernel slices a 2^-f_n against dZ 2^f_n (one store per row and layer; rows
            // beyond Nb to the dump area: the counted waits of the tile stream want the same stores from every lane group 0)
            if constexpr (STASH) {
                if (rowsc != nullptr && q == 0) (live ? rowsc + layer * rs_stride + (int64_t)b * a.Nb + nrow : dmp)[0] = rs;
            }
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                v4i Bcur[NS];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    double vv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) vv[r] = V[4 * kc + t][r];
                    int S[NS];
                    slice4s(vv, sl, S);
#pragma unroll
                    for (int k = 0; k < NS; ++k) Bcur[k][t] = S[k];
                }
#pragma unroll
                for (int k = 0; k < NS; ++k) Bin[kc][k] = to_acc(Bcur[k]);
            }
            amax = 0.0;
        };
        slice_rows(0);

        // ---- hidden -> hidden layers
        double prt = 0.0;
        auto load_frags = [&](v4i (&Af)[NS], const unsigned char* blk) {
#pragma unroll
            for (int wi = 0; wi < NS; ++wi) Af[wi] = *reinterpret_cast<const v4i*>(blk + wi * 1024);
        };
        auto burst = [&](v4i (&acc)[NLEV], const unsigned char* tile) {
            v4i Af[2][NS];
            load_frags(Af[0], tile);
            for_each_stage([&](auto k_tag) {
                constexpr int k = decltype(k_tag)::value, kc = k / NPROD, kk = k - kc * NPROD;
                if constexpr (kk == 0 && kc + 1 < KC) load_frags(Af[(kc + 1) & 1], tile + (kc + 1) * NS * 1024);
                issue_product_c<LMIN, NLEV, kc == 0, kk>(acc, Af[kc & 1], Bin[kc]);
            }, std::make_integer_sequence<int, NPT>{});
        };
        // epilogue of tile T (pinned micro-steps, the next tile's MFMAs dealt out between them): recombine the levels,
        // z = (sum 2^f_n) x scale_j + bias_j, a = max(z, alo); the last hidden layer feeds the output layer's dot instead of V
        auto epilogue = [&](auto last_tag, auto next_tag, auto t_tag, const v4i (&acc)[NLEV], v4i (&accn)[NLEV],
                            const unsigned char* tile_next, const double* sbt, const double* wlt, double* (&sp)[4], int64_t sstride) {
            constexpr bool LAST = decltype(last_tag)::value, NEXT = decltype(next_tag)::value;
            constexpr int Tt_ = decltype(t_tag)::value;
            constexpr int NST = NLEV + 2, NMICRO = NST * 4, LEAD = 2;
            v4i Af[2][NS];
            double2 sc[4];
            double ts[4], z[4];
            auto micro = [&](auto id_tag) {
                constexpr int id = decltype(id_tag)::value;
                if constexpr (id == 0 && NEXT) load_frags(Af[0], tile_next);
                if constexpr (NEXT && id >= LEAD) {
                    constexpr int from = ((id - LEAD) * NPT + (NMICRO - LEAD) - 1) / (NMICRO - LEAD);
                    constexpr int upto = ((id - LEAD + 1) * NPT + (NMICRO - LEAD) - 1) / (NMICRO - LEAD);
                    for_each_stage([&](auto k_tag) {
                        constexpr int k = from + decltype(k_tag)::value, kc = k / NPROD, kk = k - kc * NPROD;
                        if constexpr (kk == 0 && kc + 1 < KC) load_frags(Af[(kc + 1) & 1], tile_next + (kc + 1) * NS * 1024);
                        issue_product_c<LMIN, NLEV, kc == 0, kk>(accn, Af[kc & 1], Bin[kc]);
                    }, std::make_integer_sequence<int, upto - from>{});
                }
                constexpr int st = id >> 2, r = id & 3;
                if constexpr (st == 0) {
                    sc[r] = *reinterpret_cast<const double2*>(sbt + 2 * r);
                    ts[r] = (double)acc[NLEV - 1][r];
                } else if constexpr (st < NLEV) {
                    { const double cv_ = (double)acc[NLEV - 1 - st][r]; asm("v_fma_f64 %0, %1, %2, %3" : "=v"(ts[r]) : "v"(ts[r]), "s"(256.0), "v"(cv_)); }
                } else if constexpr (st == NLEV) {
                    z[r] = fma(ts[r] * rs, sc[r].x, sc[r].y);
                } else {
                    const double av = fmax(z[r], alo);
                    if constexpr (STASH) { *sp[r] = av; sp[r] += sstride; }
                    if constexpr (LAST) {
                        prt = fma(wlt[r], av, prt);
                    } else {
                        amax = fmax(amax, fabs(av));
                        V[Tt_][r] = to_acc_d(av);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            for_each_stage(micro, std::make_integer_sequence<int, NMICRO>{});
        };
        auto hidden_layer = [&](auto last_tag, int li) {
            constexpr bool LAST = decltype(last_tag)::value;
            const double* sb = lds + offsb + li * 2 * HID + 2 * 4 * q;       // this lane group's features 16 T + 4 q + r
            const double* wl = lds + offWl + 4 * q;
            double* stl = STASH ? act0 + (int64_t)(li + 1) * a.act_stride + srow : nullptr;
            double* sp[4];
            int64_t sstride = live ? (int64_t)16 * a.Ns : 0;
            asm volatile("" : "+v"(sstride));
#pragma unroll
            for (int r = 0; r < 4; ++r) sp[r] = STASH ? (live ? stl : dmp) + (int64_t)r * a.Ns : nullptr;
            v4i accA[NLEV], accB[NLEV];
            sync_tile();
            burst(accA, ring + rd_slot * TILE_B + lofs);
            auto tile = [&](auto t_tag) {
                constexpr int Tt_ = decltype(t_tag)::value;
                rd_slot = rd_slot + 1 == WNBUF ? 0 : rd_slot + 1;            // (now the slot of tile Tt_ + 1)
                const unsigned char* nxt = ring + rd_slot * TILE_B + lofs;
                if constexpr (Tt_ + 1 < TL) {
                    sync_tile();
                    if constexpr (Tt_ & 1) epilogue(last_tag, std::true_type{}, t_tag, accB, accA, nxt, sb + 32 * Tt_, wl + 16 * Tt_, sp, sstride);
                    else epilogue(last_tag, std::true_type{}, t_tag, accA, accB, nxt, sb + 32 * Tt_, wl + 16 * Tt_, sp, sstride);
                } else {
                    if constexpr (Tt_ & 1) epilogue(last_tag, std::false_type{}, t_tag, accB, accA, nxt, sb + 32 * Tt_, wl + 16 * Tt_, sp, sstride);
                    else epilogue(last_tag, std::false_type{}, t_tag, accA, accB, nxt, sb + 32 * Tt_, wl + 16 * Tt_, sp, sstride);
                }
            };
            for_each_stage(tile, std::make_integer_sequence<int, TL>{});
            if constexpr (!LAST) slice_rows(li + 1);
        };
        for (int li = 0; li < NHH - 1; ++li) hidden_layer(std::false_type{}, li);
        hidden_layer(std::true_type{}, NHH - 1);

        // ---- last layer: finish the dot over the four lane groups, residual, SSE
        double pq = prt;
        pq += __shfl_xor(pq, 16, 64);
        pq += __shfl_xor(pq, 32, 64);
        const double pr = pq + lds[offbl];
        const double res = pr - yk;
        if (live && q == 0) {
            sse += res * res;
            if (pred_out) pred_out[(int64_t)b * a.Nb + nrow] = pr;
            if (dz_last) dz_last[(int64_t)b * a.Nb + nrow] = 2.0 * res;
        }
    }
    sse = wave_sum(sse);
    if (lane == 0) red[wave] = sse;
    __syncthreads();                                     // (vmcnt(0): the two tiles fetched ahead have landed too)
    if (tid == 0) partial[(int64_t)b * a.nsplit + split] = (red[0] + red[1]) + (red[2] + red[3]);
}

// gradW[b][offW + f] = sum over the row splits of the output layer's partial weight gradients (entry h: the bias)
__global__ void k_wide_dwl_sum(const double* __restrict__ slab, int nsplit, int h, int has_bias, int64_t p, int64_t offW, int64_t offB,
                               double* __restrict__ gradW) {
    const int b = blockIdx.x;
    for (int e = threadIdx.x; e < h + (has_bias ? 1 : 0); e += blockDim.x) {
        double sv = 0.0;
        for (int k = 0; k < nsplit; ++k) sv += slab[((int64_t)b * nsplit + k) * (h + 1) + e];
        gradW[(int64_t)b * p + (e < h ? offW + e : offB)] = sv;
    }
}

__global__ void k_wide_sum(const double* __restrict__ partial, int nsplit, int B, double* __restrict__ sse) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double s = 0.0;
    for (int i = 0; i < nsplit; ++i) s += partial[(int64_t)b * nsplit + i];
    sse[b] = s;
}

void wide_plan(int B, int Nb, WideArgs* a) {
    const int rows_it = 64, target = 256;                // one workgroup per CU
    const int max_split = (Nb + rows_it - 1) / rows_it;
    int nsplit = (target + B - 1) / B;
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    int rps = (Nb + nsplit - 1) / nsplit;
    rps = (rps + rows_it - 1) / rows_it * rows_it;
    a->nsplit = (Nb + rps - 1) / rps;
    a->rows_per_split = rps;
    a->iters = rps / rows_it;
}

// Raise a kernel's dynamic-LDS limit ONCE per (kernel, device) to the hardware maximum: the bytes a launch asks for grow with
// the number of hidden layers while the kernel templates do not depend on it (arming with the first caller's size left a
// later, deeper network above the limit), and the attribute is per device.
int wide_arm(const void* fn, size_t bytes) {
    static std::mutex mu;
    static std::unordered_set<uint64_t> armed;
    if (bytes > 160 * 1024) {
        qn_set_error("int8-slice kernel needs %zu bytes of LDS (limit 160 KB)", bytes);
        return QN_EUNSUPPORTED;
    }
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

// columns of the first layer's LDS image / registers of a row's inputs: 2, 4 or (5..8 inputs: round 4) 8
// launch of k_i8_wide_fwd_u (relu / identity); `args`: the WideArgs of the call (opaque here: the struct lives in each object's
// anonymous namespace)
int qn_i8_wide_launch_u(int h, int dp, int stash, size_t lds, unsigned grid, hipStream_t st, const void* args, const double* W,
                        const double* X, const double* Y, const int32_t* row_idx, const unsigned char* Wd, const double* sb,
                        const int* flags, double* act0, double* dz_last, double* pred, double* partial, double* dump, int act,
                        double* rowsc, int64_t rs_stride);
#if QN_WIDE_PART != 0
int qn_i8_wide_launch_u(int h, int dp, int stash, size_t lds, unsigned grid, hipStream_t st, const void* args, const double* W,
                        const double* X, const double* Y, const int32_t* row_idx, const unsigned char* Wd, const double* sb,
                        const int* flags, double* act0, double* dz_last, double* pred, double* partial, double* dump, int act,
                        double* rowsc, int64_t rs_stride) {
    WideArgs a;
    memcpy(&a, args, sizeof(a));
    using ufn = void (*)(WideArgs, const double*, const double*, const double*, const int32_t*, const unsigned char*,
                         const double*, const int*, double*, double*, double*, double*, double*, int, double*, int64_t);
    ufn ku;
    if (stash)
        ku = h == 128 ? (dp == 2 ? k_i8_wide_fwd_u<2, 2, QN_I8_LMIN, true> : dp == 4 ? k_i8_wide_fwd_u<2, 4, QN_I8_LMIN, true> : k_i8_wide_fwd_u<2, 8, QN_I8_LMIN, true>)
                      : (dp == 2 ? k_i8_wide_fwd_u<4, 2, QN_I8_LMIN, true> : dp == 4 ? k_i8_wide_fwd_u<4, 4, QN_I8_LMIN, true> : k_i8_wide_fwd_u<4, 8, QN_I8_LMIN, true>);
    else
        ku = h == 128 ? (dp == 2 ? k_i8_wide_fwd_u<2, 2, QN_I8_LMIN, false> : dp == 4 ? k_i8_wide_fwd_u<2, 4, QN_I8_LMIN, false> : k_i8_wide_fwd_u<2, 8, QN_I8_LMIN, false>)
                      : (dp == 2 ? k_i8_wide_fwd_u<4, 2, QN_I8_LMIN, false> : dp == 4 ? k_i8_wide_fwd_u<4, 4, QN_I8_LMIN, false> : k_i8_wide_fwd_u<4, 8, QN_I8_LMIN, false>);
    if (int rc = wide_arm(reinterpret_cast<const void*>(ku), lds)) return rc;
    hipLaunchKernelGGL(ku, dim3(grid), dim3(WWG), lds, st, a, W, X, Y, row_idx, Wd, sb, flags, act0, dz_last, pred, partial, dump, act,
                       rowsc, rs_stride);
    return QN_OK;
}
#endif

#if QN_WIDE_PART != 1
static int wide_dp(int d) { return d <= 2 ? 2 : (d <= 4 ? 4 : 8); }
bool qn_i8_wide_applies(const qn_desc* d) {
    const int L = d->nlayers;
    if (d->kind != QN_KIND_MLP || L < 3 || d->dims[0] > 8 || d->dims[L] != 1) return false;
#ifdef QN_WIDE_TANH_ONLY                                            // (A/B: relu / identity on the layer-wise float64 kernels, as until round 4)
    if (d->act != QN_ACT_TANH) return false;
#endif
    const int h = d->dims[1];
    if (h != 128 && h != 256) return false;
    for (int l = 1; l < L; ++l)
        if (d->dims[l] != h) return false;
    return wide_lds_bytes(h / 64, wide_dp(d->dims[0]), L - 1) <= 160 * 1024 && wideb_lds_bytes(h / 64, L - 1) <= 160 * 1024;
}
// bytes of: weight digit planes | {scale, bias} pairs | chain flags | SSE partials | dump area of the activation stash
size_t qn_i8_wide_workspace(const qn_desc* d, int B, int Nb, int want_grad) {
    if (!qn_i8_wide_applies(d)) return 0;
    const int h = d->dims[1], nhh = d->nlayers - 2;
    WideArgs a;
    wide_plan(B, Nb, &a);
    size_t tot = qn_align((size_t)B * nhh * NS * h * h) + qn_align((size_t)B * nhh * 2 * h * sizeof(double)) +
                 qn_align((size_t)B * sizeof(int)) + qn_align((size_t)B * a.nsplit * sizeof(double)) +
                 qn_align(((size_t)3 * Nb + 64) * sizeof(double));
    // backward: digit planes of the transposed matrices | their scales | chain flags
    // ... | (h = 128) per-split partial sums of the output layer's weight gradient
    if (want_grad)
        tot += qn_align((size_t)B * nhh * NS * h * h) + qn_align((size_t)B * nhh * h * sizeof(double)) + qn_align((size_t)B * sizeof(int)) +
               qn_align((size_t)B * a.nsplit * (h + 1) * sizeof(double));
    // relu / identity: ... | the forward's per-row activation scales [nhh][B][Nb] for the weight-gradient kernel
    if (want_grad && d->act != QN_ACT_TANH) tot += qn_align((size_t)nhh * B * Nb * sizeof(double));
    return tot;
}
double* qn_i8_wide_rowscale(const qn_desc* d, int B, int Nb, int want_grad, void* ws) {
    if (!ws || !want_grad || !qn_i8_wide_applies(d) || d->act == QN_ACT_TANH) return nullptr;
    const int nhh = d->nlayers - 2;
    return reinterpret_cast<double*>(static_cast<char*>(ws) + qn_i8_wide_workspace(d, B, Nb, 1) -
                                     qn_align((size_t)nhh * B * Nb * sizeof(double)));
}
// One launch: sse [B] (+ pred [B][Nb], dz_last [B][Nb] = 2 (pred - y), hidden activations act0 + l * act_stride
// [B][h][Nb] for l = 0 .. L-2, each optional)
int qn_i8_wide_forward(const qn_desc* d, const double* W, const double* X, const double* Y, const int32_t* row_idx, int B,
                       int Nb, double* act0, int64_t act_stride, int Ns, double* dz_last, double* pred, double* sse, void* ws,
                       hipStream_t st) {
    if (act0 && Ns < Nb) return QN_EINVAL;
    if (!qn_i8_wide_applies(d)) return QN_EUNSUPPORTED;
    const int h = d->dims[1], nhh = d->nlayers - 2;
    I8Net net;
    net.nl = nhh; net.p = d->p; net.has_bias = d->has_bias;
    for (int li = 0; li < nhh; ++li) {
        net.h[li] = h; net.h[li + 1] = h;
        net.offW[li] = d->offW[li + 1]; net.offB[li] = d->offB[li + 1];
        net.offD[li] = (int64_t)li * NS * h * h; net.offS[li] = (int64_t)li * 2 * h;
    }
    net.dbytes = (int64_t)nhh * NS * h * h;
    net.sdoubles = (int64_t)nhh * 2 * h;
    char* base = static_cast<char*>(ws);
    unsigned char* Wd = reinterpret_cast<unsigned char*>(base);
    base += qn_align((size_t)B * net.dbytes);
    double* sb = reinterpret_cast<double*>(base);
    base += qn_align((size_t)B * net.sdoubles * sizeof(double));
    int* flags = reinterpret_cast<int*>(base);
    base += qn_align((size_t)B * sizeof(int));
    double* partial = reinterpret_cast<double*>(base);
    WideArgs a;
    wide_plan(B, Nb, &a);
    base += qn_align((size_t)B * a.nsplit * sizeof(double));
    double* dump = reinterpret_cast<double*>(base);
    a.p = d->p; a.act_stride = act_stride; a.B = B; a.Nb = Nb; a.d = d->dims[0]; a.nhid = d->nlayers - 1;
    a.has_bias = d->has_bias; a.Ns = act0 ? Ns : Nb;
    QN_HIP_CHECK(hipMemsetAsync(flags, 0, (size_t)B * sizeof(int), st));
    hipLaunchKernelGGL((k_i8_slice_w<QN_I8_LMIN>), dim3(B, nhh, 8), dim3(256), 0, st, net, W, Wd, sb, flags);
    const int dp = wide_dp(a.d);
    const size_t lds = wide_lds_bytes(h / 64, dp, a.nhid);
    using kfn = void (*)(WideArgs, const double*, const double*, const double*, const int32_t*, const unsigned char*,
                         const double*, const int*, double*, double*, double*, double*, double*);
    kfn kern;
    if (act0)
        kern = h == 128 ? (dp == 2 ? k_i8_wide_fwd<2, 2, QN_I8_LMIN, true> : dp == 4 ? k_i8_wide_fwd<2, 4, QN_I8_LMIN, true> : k_i8_wide_fwd<2, 8, QN_I8_LMIN, true>)
                        : (dp == 2 ? k_i8_wide_fwd<4, 2, QN_I8_LMIN, true> : dp == 4 ? k_i8_wide_fwd<4, 4, QN_I8_LMIN, true> : k_i8_wide_fwd<4, 8, QN_I8_LMIN, true>);
    else
        kern = h == 128 ? (dp == 2 ? k_i8_wide_fwd<2, 2, QN_I8_LMIN, false> : dp == 4 ? k_i8_wide_fwd<2, 4, QN_I8_LMIN, false> : k_i8_wide_fwd<2, 8, QN_I8_LMIN, false>)
                        : (dp == 2 ? k_i8_wide_fwd<4, 2, QN_I8_LMIN, false> : dp == 4 ? k_i8_wide_fwd<4, 4, QN_I8_LMIN, false> : k_i8_wide_fwd<4, 8, QN_I8_LMIN, false>);
    if (d->act != QN_ACT_TANH) {                                    // relu / identity: per-row activation scales (qn_wide_u_i8.hip)
        if (int rc = qn_i8_wide_launch_u(h, dp, act0 != nullptr, lds, qn_fused_grid(a.nsplit, B), st, &a, W, X, Y, row_idx,
                                         (const unsigned char*)Wd, (const double*)sb, (const int*)flags, act0, dz_last, pred, partial,
                                         dump, d->act, act0 ? qn_i8_wide_rowscale(d, B, Nb, 1, ws) : (double*)nullptr, (int64_t)B * Nb))
            return rc;
    } else {
        if (int rc = wide_arm(reinterpret_cast<const void*>(kern), lds)) return rc;
        hipLaunchKernelGGL(kern, dim3(qn_fused_grid(a.nsplit, B)), dim3(WWG), lds, st, a, W, X, Y, row_idx,
                           (const unsigned char*)Wd, (const double*)sb, (const int*)flags, act0, dz_last, pred, partial, dump);
    }
    hipLaunchKernelGGL(k_wide_sum, dim3((B + 63) / 64), dim3(64), 0, st, (const double*)partial, a.nsplit, B, sse);
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}

// dZ_l = d sse / d (pre-activation of hidden layer l), l = 0 .. L-2, as float64 [B][h][Nb] at dz0 + l * dz_stride, from the
// forward's stashed activations (act0 + l * act_stride) and dz_last = 2 (pred - y).  `ws` is the SAME workspace the forward
// call of this evaluation used (its dump area is shared; the backward's own regions lie behind the forward's).
// gradW (may be null): the flat gradient [B][p]; at h = 128 the OUTPUT layer's weight / bias gradient is written there too
// (*last_done = 1: the caller skips its own kernel for that layer)
int qn_i8_wide_backward(const qn_desc* d, const double* W, const double* X, const int32_t* row_idx, int B, int Nb,
                        const double* act0, int64_t act_stride, int Ns, const double* dz_last, double* dz0, int64_t dz_stride, void* ws,
                        double* gradW, int* last_done, hipStream_t st) {
    if (last_done) *last_done = 0;
    if (Ns < Nb) return QN_EINVAL;
    if (!qn_i8_wide_applies(d)) return QN_EUNSUPPORTED;
    const int h = d->dims[1], nhh = d->nlayers - 2;
    I8Net net;
    net.nl = nhh; net.p = d->p; net.has_bias = d->has_bias;
    for (int li = 0; li < nhh; ++li) {
        net.h[li] = h; net.h[li + 1] = h;
        net.offW[li] = d->offW[li + 1]; net.offB[li] = d->offB[li + 1];
        net.offD[li] = (int64_t)li * NS * h * h; net.offS[li] = (int64_t)li * h;
    }
    net.dbytes = (int64_t)nhh * NS * h * h;
    net.sdoubles = (int64_t)nhh * h;
    WideArgs fa;
    wide_plan(B, Nb, &fa);
    char* base = static_cast<char*>(ws);
    base += qn_align((size_t)B * net.dbytes) + qn_align((size_t)B * nhh * 2 * h * sizeof(double)) + qn_align((size_t)B * sizeof(int)) +
            qn_align((size_t)B * fa.nsplit * sizeof(double));
    double* dump = reinterpret_cast<double*>(base);
    base += qn_align(((size_t)3 * Nb + 64) * sizeof(double));
    unsigned char* WdT = reinterpret_cast<unsigned char*>(base);
    base += qn_align((size_t)B * net.dbytes);
    double* scT = reinterpret_cast<double*>(base);
    base += qn_align((size_t)B * net.sdoubles * sizeof(double));
    int* flags = reinterpret_cast<int*>(base);
    base += qn_align((size_t)B * sizeof(int));
    double* dwl_slab = reinterpret_cast<double*>(base);
    const bool fold_last = (h == 128 || QN_WIDE_FOLD_ALL) && gradW != nullptr && last_done != nullptr;
    WideBwdArgs a;
    a.p = d->p; a.act_stride = act_stride; a.dz_stride = dz_stride; a.B = B; a.Nb = Nb; a.Ns = Ns; a.d = d->dims[0];
    a.nhid = d->nlayers - 1; a.has_bias = d->has_bias;
    a.nsplit = fa.nsplit; a.rows_per_split = fa.rows_per_split; a.iters = fa.iters;
    QN_HIP_CHECK(hipMemsetAsync(flags, 0, (size_t)B * sizeof(int), st));
    hipLaunchKernelGGL((k_i8_slice_wT<QN_I8_LMIN>), dim3(B, nhh, h / 16), dim3(256), 0, st, net, W, WdT, scT, flags);
    const int dp = a.d <= 2 ? 2 : 4;
    const size_t lds = wideb_lds_bytes(h / 64, a.nhid);
    using kfn = void (*)(WideBwdArgs, const double*, const double*, const int32_t*, const unsigned char*, const double*,
                         const int*, const double*, const double*, double*, double*, double*, int);
    kfn kern;
    if (d->act == QN_ACT_TANH)
        kern = h == 128 ? (dp == 2 ? k_i8_wide_bwd<2, 2, QN_I8_LMIN, true> : k_i8_wide_bwd<2, 4, QN_I8_LMIN, true>)
                        : (dp == 2 ? k_i8_wide_bwd<4, 2, QN_I8_LMIN, true> : k_i8_wide_bwd<4, 4, QN_I8_LMIN, true>);
    else
        kern = h == 128 ? (dp == 2 ? k_i8_wide_bwd<2, 2, QN_I8_LMIN, false> : k_i8_wide_bwd<2, 4, QN_I8_LMIN, false>)
                        : (dp == 2 ? k_i8_wide_bwd<4, 2, QN_I8_LMIN, false> : k_i8_wide_bwd<4, 4, QN_I8_LMIN, false>);
    if (int rc = wide_arm(reinterpret_cast<const void*>(kern), lds)) return rc;
    hipLaunchKernelGGL(kern, dim3(qn_fused_grid(a.nsplit, B)), dim3(WWG), lds, st, a, W, X, row_idx, (const unsigned char*)WdT,
                       (const double*)scT, (const int*)flags, act0, dz_last, dz0, dump, fold_last ? dwl_slab : (double*)nullptr,
                       d->act);
    if (fold_last) {
        const int L = d->nlayers;
        hipLaunchKernelGGL(k_wide_dwl_sum, dim3(B), dim3(192), 0, st, (const double*)dwl_slab, a.nsplit, h, d->has_bias, d->p,
                           d->offW[L - 1], d->offB[L - 1], gradW);
        *last_done = 1;
    }
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}
#endif  // QN_WIDE_PART != 1
