// Fused float64 forward + BACKWARD for 64-wide tanh networks with every 64 x 64 product -- the hidden layers of the forward
// pass, dA = W^T dZ and dW = dZ A^T -- as SLICED EXACT PRODUCTS on the int8 matrix pipe (v_mfma_i32_16x16x64_i8): the
// gradient of BASELINE configs[1] (64 chains, 3x64, N = 4096), i.e. what logPostGrad costs an HMC / MALA proposal
// (quinn/solvers/nn_mcmc.py:73-98 -> quinn/nns/nnwrap.py:128-150, called L + 1 times per proposal in quinn/mcmc/hmc.py:48-60).
//
// Why.  k_fused_bwd_f64 (qn_fused.hip) pays 384 float64 MFMAs per 16 data rows on the vector pipe (24.6 k cycles) plus the
// activations: 0.46-0.48 of the float64 MFMA peak.  As sliced int8 products the same six GEMMs are 6 x 104 MFMAs = 10 k
// cycles on the OTHER pipe (qn_fused_i8.hip explains the arithmetic), next to ~3.5 k vector instructions.
//
// Organisation: workgroup = 4 waves = one chain x 64 data rows per iteration, one wave per SIMD (launch bound 1: the
// weight digit planes of W AND W^T, 2 x 24 KB per hidden matrix, plus two transposition stashes fill the LDS); lane
// (q, c) of a wave holds features 16 t + 4 q + r of data row c, as in k_fused_fwd_i8.
//   forward   a_1 on the VALU, a_2 .. a_NH by digit products (A operand: W digits from LDS, B operand: the previous
//             layer's digits, in registers); every a_l is kept as float64 (for 1 - a^2) and a_1 .. a_{NH-1} as digits.
//   backward, for l = NH-1 .. 1, with dZ = dZ_{l+1} in registers (float64, accumulator layout):
//     scale   2^G > every |dZ| of the workgroup's 64 rows (wave maxima exchanged through LDS ahead of the barrier the
//             stash needs anyway); dZ is sliced ONCE with it: digits m = round(dZ 2^(46 - G)) serve both products;
//     dW_l    contraction over the 64 DATA ROWS, which sit on lanes: the digit words of dZ and of a_l (4 features of
//             one row each) go through a 4 x 4 byte transposition inside lane quads (2 DPP moves + 2 v_perm per word:
//             4 rows of one feature) into two LDS stashes [6][64 features][64 rows], the layout of a weight digit
//             plane; wave w then owns output rows 16 w .. 16 w + 15: 4 tiles x 26 exact products, recombined into
//             float64 accumulators once per iteration (the scale changes with the iteration);
//     db_l    lane-local sums of dZ, reduced over lanes once at the end;
//     dA      A operand: digit planes of W^T (sliced with one scale per COLUMN of W), B operand: the digits of dZ;
//             dZ_l = 2^G scale_i sum . (1 - a_l^2).
//   thin first / last layer (d <= 2 inputs, one output): lane-local accumulators as in k_fused_bwd_f64's fast path.
// Output conventions of k_fused_bwd_f64: SSE partial per (chain, row split), gradient slab [B][nsplit][p].
//
// Accuracy: operands are rounded to 2^-47 of their scale (activations: 1; weights: row / column maximum; dZ: the
// workgroup maximum of the layer) -- a norm-wise 47-bit bound like the forward kernel's; measured against the float64
// kernels ~1e-13 of max |g| (tests: 1e-10).
//
// Anything outside the fast path's contract -- a weight or input that is not finite and < 2^500, a weight >= 2^20 in a
// sliced matrix, a layer whose activations are all tiny (qn_i8_slice.h), a gradient beyond 2^500 -- FLAGS the (chain,
// split); the caller then launches k_fused_bwd_f64 with the flags, which recomputes every split of a flagged chain in
// plain float64 (and returns at once for all others).
#include "qn_common.h"
#include "qn_fused_args.h"
#include "qn_math.h"
#include "qn_i8_slice.h"

namespace {

// Slot swizzle of this kernel's digit planes AND transposition stashes: slot_swz (qn_i8_slice.h: rows of equal R & 3 in distinct
// slots -> conflict-free ds_read_b128 fragments), optionally with one more bit, R >> 1 & 1 (QN_BWD_STASH_SWZ = 1).
// Where this kernel's 12.6 % SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE come from (rounds 3 and 4 could not place them): a ds_write_b32
// is banked (a / 4) % 32 within each 32-lane half (every ds_write; the b128 reads: % 64 per 16-lane group), and under slot_swz
// alone the stash rows R and R + 2 of a half-wave meet on one bank -- 2-way on EVERY transposed stash write (1.57 M writes x 2 extra
// cycles = the 3.1 M conflict cycles per launch above the forward kernel's 0.24 M of tanh-table lookups).  The extra bit removes them
// (measured: 3.38 M -> 0.237 M, 12.6 % -> 1.0 %; it is constant over the rows a staging half-wave writes and over each read group's
// rows of equal R & 3, so those stay conflict-free) -- and the launch gets 0.6 % SLOWER in the same call (219.4 -> 220.7 us, twice,
// tools/ab_bwd_stash_swz.sh): the LDS is not what this kernel waits for (vector issue is), and the writes' second cycle was hidden.
// Kept as an A/B switch, default off.
#ifndef QN_BWD_STASH_SWZ
#define QN_BWD_STASH_SWZ 0
#endif
__device__ __forceinline__ int bwd_swz(int row) { return slot_swz(row) ^ (QN_BWD_STASH_SWZ ? (row >> 1) & 1 : 0); }

constexpr int BWG = 256;

#ifdef QN_BWD8_STAMPS
// diagnostic build only (tools/bwd8_stamps.py): per-phase cycles of wave 0 of workgroup 0
#define QN_STAMP(k)                                                                          \
    do {                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                  \
        const long long now_ = __builtin_amdgcn_s_memtime();                                 \
        stamp_acc[k] += now_ - stamp_prev;                                                   \
        stamp_prev = now_;                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    } while (0)
#else
#define QN_STAMP(k) do { } while (0)
#endif

// LDS, doubles first: W0 [64][DP] | b0 [64] | Wl [64] | bl, pad | red [8] | wave exponents [4] | {scale, bias} (NH-1) x [64][2]
// | column scales (NH-1) x [64] | tanh table; then bytes: (NH-1) planes of W | (NH-1) planes of W^T | stash dZ | stash A
// (3 or 4 inputs: W0 takes 128 doubles more and the image at three hidden layers would be 128 BYTES over the 160 KB of a CU; the
// tanh table then ends at n = 1264 (|x| clamped to 19.75 instead of 20): tanh is 1.0 to the last bit from 19.07 on, so the
// results are the same)
constexpr int TANH_TAB_SHORT_N = 1265;
__host__ __device__ constexpr int bwd_tab_doubles(int dp) { return dp <= 2 ? ((TANH_TAB + 1) & ~1) : ((TANH_TAB_SHORT_N + 1) & ~1); }
__host__ __device__ constexpr int bwd_head_doubles(int dp, int nhid) {
    return ((H * dp + H + H + 2 + 8 + 4 + (nhid - 1) * 3 * H + 1) & ~1) + bwd_tab_doubles(dp);
}
__host__ __device__ constexpr size_t bwd_lds_bytes(int dp, int nhid) {
    return sizeof(double) * (size_t)bwd_head_doubles(dp, nhid) + (size_t)(2 * (nhid - 1) + 2) * LAYER_BYTES;
}

// `magic`: kMagic held in a register pair the compiler cannot see through (a known constant makes it emit v_fmac_f64 behind
// a move that re-materialises the addend: with one wave per SIMD every instruction is paid in full; tools/ubench_xpose.hip)
__device__ __forceinline__ void slice4_scaled(const double (&a)[4], double scale, double magic, int (&S)[NS]) {
    int lo[4], hi[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double x;                                                   // (one v_fma_f64; see above)
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(x) : "v"(a[r]), "s"(scale), "v"(magic));
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

// 4 x 4 byte transposition inside a lane quad: in: lane u holds bytes (features) 0..3 of its data row; out: lane u holds
// feature u of the quad's four rows (byte v = row of lane v).  selA / selB: the lane's v_perm selectors (by lane & 1, lane & 2).
__device__ __forceinline__ int quad_xpose(int x, int selA, int selB) {
    const int y = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false);       // quad_perm [1,0,3,2]: lane u ^ 1
    const int z = __builtin_amdgcn_perm(y, x, selA);
    const int w = __builtin_amdgcn_update_dpp(0, z, 0x4E, 0xF, 0xF, false);       // quad_perm [2,3,0,1]: lane u ^ 2
    return __builtin_amdgcn_perm(w, z, selB);
}

// Scheduling request for the region that ends here (between two sched_barrier(0)): the fragment reads first, then NMFMA
// times one MFMA followed by VPM vector instructions -- the next tile's products between the instructions of this tile's
// epilogue (what qn_fused_i8.hip / qn_wide_i8.hip do with hand-written stages).
#ifndef QN_BWD8_VPM_FWD
#define QN_BWD8_VPM_FWD 5
#endif
#ifndef QN_BWD8_VPM_DW
#define QN_BWD8_VPM_DW 2
#endif
#ifndef QN_BWD8_VPM_DA
#define QN_BWD8_VPM_DA 3
#endif
template <int NMFMA, int VPM>
__device__ __forceinline__ void interleave_hint_nolds() {
#pragma unroll
    for (int i = 0; i < NMFMA; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
    }
}
template <int NMFMA, int VPM>
__device__ __forceinline__ void interleave_hint() {
    __builtin_amdgcn_sched_group_barrier(0x100, NS, 0);
    interleave_hint_nolds<NMFMA, VPM>();
}

// The value is computed HERE (an empty volatile asm: IR-level code motion would otherwise sink an epilogue whose result
// is not needed before the end of the iteration out of the region its MFMAs are interleaved with).
__device__ __forceinline__ void pin(double& v) { asm volatile("" : "+v"(v)); }

// level sums (units of 256^LMIN) -> one float64: pairs of levels are added in int32 first (K = 64: a level is < 2^23)
template <int NLEV>
__device__ __forceinline__ double recombine(const v4i (&acc)[NLEV], int r) {
    double ts = (NLEV & 1) ? (double)acc[NLEV - 1][r] : (double)(acc[NLEV - 2][r] + (acc[NLEV - 1][r] << 8));
#pragma unroll
    for (int l = ((NLEV & 1) ? NLEV - 3 : NLEV - 4); l >= 0; l -= 2) ts = fma(ts, 65536.0, (double)(acc[l][r] + (acc[l + 1][r] << 8)));
    return ts;
}

// The same for the four elements of a tile, STAGE-MAJOR: every step is written for all four elements before the next step,
// so that a wave that is alone on its SIMD always has four independent instructions between a result and its use (written
// element by element the compiler keeps each element's dependent chain together and the wave waits out every latency).
template <int NLEV>
__device__ __forceinline__ void recombine4(const v4i (&acc)[NLEV], double (&ts)[4], double c65536) {
    // c65536: 65536.0 in a register pair the compiler cannot see through (as a literal it makes each of these fma's an
    // 8-byte v_fmac_f64; an inline-asm fma would drop out of the scheduling requests' count of vector instructions)
#pragma unroll
    for (int r = 0; r < 4; ++r) ts[r] = (NLEV & 1) ? (double)acc[NLEV - 1][r] : (double)(acc[NLEV - 2][r] + (acc[NLEV - 1][r] << 8));
#pragma unroll
    for (int l = ((NLEV & 1) ? NLEV - 3 : NLEV - 4); l >= 0; l -= 2)
#pragma unroll
        for (int r = 0; r < 4; ++r) ts[r] = fma(ts[r], c65536, (double)(acc[l][r] + (acc[l + 1][r] << 8)));
}
// qn_tanh_f64_tab64 (qn_math.h) for N arguments at once, stage-major; same operations, same results
template <int N, bool SHORT_TAB = false>
__device__ __forceinline__ void tanh_tab64_n(const double (&z)[N], double (&out)[N], const double* __restrict__ tab, double magic52, double cm13) {
    double ax[N], zm[N], Tt[N], bb[N], b2[N], pp[N], b3[N], tb[N], num[N], den[N], y0[N], e0[N];
#pragma unroll
    for (int r = 0; r < N; ++r) asm("v_min_f64 %0, |%1|, %2" : "=v"(ax[r]) : "v"(z[r]), "s"(SHORT_TAB ? 19.75 : 20.0));
#pragma unroll
    for (int r = 0; r < N; ++r) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(zm[r]) : "v"(ax[r]), "s"(64.0), "v"(magic52));      // magic52 = 1.5 * 2^52 (see slice4_scaled)
#pragma unroll
    for (int r = 0; r < N; ++r) Tt[r] = tab[__double2loint(zm[r])];
#pragma unroll
    for (int r = 0; r < N; ++r) bb[r] = fma(zm[r] - 6755399441055744.0, -0.015625, ax[r]);
#pragma unroll
    for (int r = 0; r < N; ++r) b2[r] = bb[r] * bb[r];
#pragma unroll
    for (int r = 0; r < N; ++r) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(pp[r]) : "v"(b2[r]), "s"(1.33333333333333333e-01), "v"(cm13));      // cm13 = -1/3
#pragma unroll
    for (int r = 0; r < N; ++r) b3[r] = bb[r] * b2[r];
#pragma unroll
    for (int r = 0; r < N; ++r) tb[r] = fma(b3[r], pp[r], bb[r]);
    // (reciprocal-free tail of qn_tanh_f64_tab64: T + (1 - T^2) tb (1 - e)(1 + e^2 + e^4), e = T tb)
#pragma unroll
    for (int r = 0; r < N; ++r) num[r] = Tt[r] * tb[r];                 // e
#pragma unroll
    for (int r = 0; r < N; ++r) den[r] = fma(-Tt[r], Tt[r], 1.0);       // 1 - T^2
#pragma unroll
    for (int r = 0; r < N; ++r) e0[r] = num[r] * num[r];                // e^2
#pragma unroll
    for (int r = 0; r < N; ++r) y0[r] = fma(-tb[r], num[r], tb[r]);     // tb (1 - e)
#pragma unroll
    for (int r = 0; r < N; ++r) e0[r] = fma(e0[r], e0[r], e0[r]);       // e^2 + e^4
#pragma unroll
    for (int r = 0; r < N; ++r) y0[r] = fma(y0[r], e0[r], y0[r]);       // u
#pragma unroll
    for (int r = 0; r < N; ++r) out[r] = __builtin_copysign(fma(den[r], y0[r], Tt[r]), z[r]);
}
// quad_xpose for N words, step-major (a DPP move needs two wait states behind the instruction that wrote its source)
template <int N>
__device__ __forceinline__ void quad_xpose_n(const int (&x)[N], int (&out)[N], int selA, int selB) {
    int y[N], z[N], w[N];
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = __builtin_amdgcn_mov_dpp(x[i], 0xB1, 0xF, 0xF, true);
#pragma unroll
    for (int i = 0; i < N; ++i) z[i] = __builtin_amdgcn_perm(y[i], x[i], selA);
#pragma unroll
    for (int i = 0; i < N; ++i) w[i] = __builtin_amdgcn_mov_dpp(z[i], 0x4E, 0xF, 0xF, true);
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = __builtin_amdgcn_perm(w[i], z[i], selB);
}

// Stage chain `Wb`: thin layers as float64, every hidden matrix as digit planes of W (one scale per row) and of W^T (one
// scale per column).  Returns whether this thread saw a weight outside the fast path's contract.
// |v| < 2^20 / 2^100 (and not NaN): the bounds on every weight / input of a relu or identity network (qn_fused_i8.hip: stage())
__device__ __forceinline__ bool bounded20(double v) { return (unsigned)(__double2hiint(v) & 0x7fffffff) < 0x41300000u; }
__device__ __forceinline__ bool bounded100(double v) { return (unsigned)(__double2hiint(v) & 0x7fffffff) < 0x46300000u; }

template <int NH, int DP, int LMIN, bool UNB = false>
__device__ __forceinline__ int stage_bwd(double* __restrict__ lds, unsigned char* __restrict__ wq, unsigned char* __restrict__ wqT,
                                         const double* __restrict__ Wb, const FusedArgs& a) {
    int bad = 0;
    auto chk = [&](double v) { bad |= UNB ? !bounded20(v) : !qn_bounded(v); return v; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = a.d, nb = a.has_bias ? 1 : 0;
    const int64_t gb0 = (int64_t)H * d, gHH = gb0 + nb * H;
    const int64_t gWl = gHH + (int64_t)(NH - 1) * (H * H + nb * H), gbl = gWl + H;
    const int lb0 = H * DP, lWl = lb0 + H, lbl = lWl + H, lsb = lbl + 2 + 8 + 4, lsT = lsb + (NH - 1) * 2 * H;
    for (int e = tid; e < H * DP; e += BWG) {
        const int j = e / DP, k = e % DP;
        lds[e] = k < d ? chk(Wb[j * d + k]) : 0.0;
    }
    for (int e = tid; e < H; e += BWG) {
        lds[lb0 + e] = nb ? chk(Wb[gb0 + e]) : 0.0;
        lds[lWl + e] = chk(Wb[gWl + e]);
    }
    if (tid == 0) lds[lbl] = nb ? chk(Wb[gbl]) : 0.0;
    const int q16 = lane & 15, m4 = q16 >> 2, g4 = q16 & 3;                // quad (m, g): k-slots 16 m + 4 g + {0..3}
    // every matrix load is issued before the first one is sliced (one memory round trip instead of 2 (NH - 1))
    double vall[NH - 1][2][4][4];
#pragma unroll
    for (int layer = 1; layer < NH; ++layer) {
        const double* Wg = Wb + gHH + (int64_t)(layer - 1) * (H * H + nb * H);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = 16 * u + 4 * wave + (lane >> 4);
            const double2* src = reinterpret_cast<const double2*>(Wg + row * H + 16 * m4 + 4 * g4);
            const double2 v01 = src[0], v23 = src[1];
            vall[layer - 1][0][u][0] = v01.x; vall[layer - 1][0][u][1] = v01.y; vall[layer - 1][0][u][2] = v23.x; vall[layer - 1][0][u][3] = v23.y;
#pragma unroll
            for (int r = 0; r < 4; ++r) vall[layer - 1][1][u][r] = Wg[(16 * m4 + 4 * g4 + r) * H + row];
        }
    }
#pragma unroll
    for (int layer = 1; layer < NH; ++layer) {
        const double* Wg = Wb + gHH + (int64_t)(layer - 1) * (H * H + nb * H);
#pragma unroll
        for (int tr = 0; tr < 2; ++tr) {                                   // 0: planes of W (row scales), 1: planes of W^T (column scales)
            unsigned char* plane = (tr ? wqT : wq) + (layer - 1) * LAYER_BYTES;
            double (&v)[4][4] = vall[layer - 1][tr];
            // step-major over the thread's four items (see stage() of qn_fused_i8.hip)
            double bias[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) bias[u] = (tr == 0 && q16 == 0 && nb) ? Wg[H * H + 16 * u + 4 * wave + (lane >> 4)] : 0.0;
            unsigned ex[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ex[u] = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (tr == 0) chk(v[u][r]);
                    ex[u] = max(ex[u], ((unsigned)__double2hiint(v[u][r]) & 0x7fffffffu) >> 20);
                }
            }
            row16_max_u32_n<4>(ex);
            int e[4];
            double an[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                e[u] = (int)ex[u] - 1022;                                  // 2^e > every |entry| of the row (of W or of W^T)
                bad |= e[u] > I8_MAX_WEIGHT_EXP;
                e[u] = e[u] < -900 ? -900 : e[u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) an[u][r] = ldexp(v[u][r], -e[u]);
            int S[4][NS];
            slice4_n<4>(an, S);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = 16 * u + 4 * wave + (lane >> 4);
                unsigned char* dst = plane + row * H + 16 * (g4 ^ bwd_swz(row)) + 4 * m4;
#pragma unroll
                for (int k = 0; k < NS; ++k) *reinterpret_cast<int*>(dst + k * SLICE_BYTES) = S[u][k];
                if (q16 == 0) {
                    const double sc = ldexp(1.0, e[u] - 2 * QB + 8 * LMIN);   // integer sum (units of 256^LMIN) -> product with unit-scale digits
                    if (tr == 0) {
                        lds[lsb + (layer - 1) * 2 * H + 2 * row] = sc;
                        lds[lsb + (layer - 1) * 2 * H + 2 * row + 1] = chk(bias[u]);
                    } else {
                        lds[lsT + (layer - 1) * H + row] = sc;
                    }
                }
            }
        }
    }
    return bad;
}

// DD = number of inputs (1..4; the LDS image of W0 is 2 columns wide for DD <= 2, 4 otherwise)
// ACT = relu / identity (round 4): the activations are unbounded, so (a) the forward products slice a layer's outputs with one
// scale per DATA ROW (row maximum over the lane's 16 values, then across the 4 lane groups; the next layer's integer sums are
// multiplied by it), after the layer's last tile; (b) the weight-gradient products, which contract over the data rows and
// cannot carry a per-row scale, slice the same float64 activations a second time with one scale for the workgroup's 64 rows
// (their exponent maximum travels through LDS with the dZ exponent, same barrier); (c) the derivative is a select.
template <int NH, int DD, int LMIN, int ACT = QN_ACT_TANH>
__global__ __launch_bounds__(BWG, 1) void k_fused_bwd_i8(FusedArgs a, const double* __restrict__ W, const double* __restrict__ X,
                                                        const double* __restrict__ Y, const int32_t* __restrict__ row_idx,
                                                        double* __restrict__ pred_out, double* __restrict__ partial,
                                                        double* __restrict__ slab, int* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NLEV = 2 * (NS - 1) - LMIN + 1, NPROD = nprod(LMIN), NM = NH - 1, DP = DD <= 2 ? 2 : 4;
    constexpr bool SHORT_TAB = DP > 2;
    constexpr bool TANH = ACT == QN_ACT_TANH;
    double* lds = reinterpret_cast<double*>(smem);
    int b, split;
    if (!qn_fused_wg(a.nsplit, a.B, &b, &split)) return;
    const int d = a.d, nb = a.has_bias ? 1 : 0;
    constexpr int offb0 = H * DP, offWl = offb0 + H, offbl = offWl + H, offred = offbl + 2, offgx = offred + 8, offsb = offgx + 4,
                  offsT = offsb + NM * 2 * H;
    double* tanh_tab = lds + ((offsT + NM * H + 1) & ~1);
    unsigned char* wq = reinterpret_cast<unsigned char*>(lds + bwd_head_doubles(DP, NH));
    unsigned char* wqT = wq + NM * LAYER_BYTES;
    unsigned char* SD = wqT + NM * LAYER_BYTES;
    unsigned char* SA = SD + LAYER_BYTES;
    double* red = lds + offred;
    int* gx = reinterpret_cast<int*>(lds + offgx);
    const double* Wb = W + (int64_t)b * a.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, c = lane & 15;
#ifdef QN_BWD8_STAMPS
    long long stamp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long stamp_prev = __builtin_amdgcn_s_memtime();
#endif
    // data of the NEXT iteration is fetched while the current one computes (an HBM / L2 round trip per iteration otherwise);
    // the first fetch is in flight during the staging
    double xn[DD], yn;
    int n_n;
    bool valid_n;
    auto fetch = [&](int it) {
        n_n = split * a.rows_per_split + it * 64 + 16 * wave + c;
        valid_n = n_n < a.Nb;
        const int nn = valid_n ? n_n : 0;
        const int64_t rr = row_idx ? (int64_t)row_idx[(int64_t)b * a.Nb + nn] : (int64_t)nn;
#pragma unroll
        for (int k = 0; k < DD; ++k) xn[k] = X[rr * DD + k];
        yn = Y[rr];
    };
    fetch(0);
    if constexpr (TANH)
        for (int e = tid; e < (SHORT_TAB ? TANH_TAB_SHORT_N : QN_TANH_TAB64_N); e += BWG) tanh_tab[e] = qn_tanh_table64_g[e];
    const bool w_bad = block_or(stage_bwd<NH, DP, LMIN, !TANH>(lds, wq, wqT, Wb, a), red + 6);
    if (w_bad) {                                                        // the float64 kernel recomputes the whole chain
        if (tid == 0) {
            flags[b * a.nsplit + split] = 1;
            partial[(int64_t)b * a.nsplit + split] = 0.0;
        }
        return;
    }
    QN_STAMP(10);
    const int lofs = c * H + 16 * (q ^ bwd_swz(c));                     // this lane's 16 bytes inside a 16-row tile of a plane / stash
    // stash write: this lane's transposed word = feature 16 t + 4 q + (c & 3) of rows 16 wave + 4 (c >> 2) .. + 3
    const int wofs = (4 * q + (c & 3)) * H + 16 * (wave ^ bwd_swz(4 * q + (c & 3))) + 4 * (c >> 2);
    const int selA = (c & 1) ? 0x03070105 : 0x06020400, selB = (c & 2) ? 0x03020706 : 0x05040100;
    int bad_run = 0;
    double sse = 0.0;
    double magic52 = 6755399441055744.0, magicS = kMagic, cm13 = -3.33333333333333333e-01, c65536 = 65536.0;      // constants as opaque register pairs
    asm volatile("" : "+v"(magic52), "+v"(magicS), "+v"(cm13), "+v"(c65536));

    double dWacc[NM][4][4];                                             // rows 16 wave + 4 q + r, columns 16 ti + c
#pragma unroll
    for (int l = 0; l < NM; ++l)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) dWacc[l][ti][r] = 0.0;
    double accB0[T][4], accWl[T][4], accW0[DD][T][4], accBl = 0.0;     // lane-local: feature 16 t + 4 q + r, summed over this lane's rows
    double dbacc[NM][4];                                                // hidden biases: rows 16 wave + 4 q + r (every lane of the group alike)
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            accWl[t][r] = 0.0;
            accB0[t][r] = 0.0;
#pragma unroll
            for (int k = 0; k < DD; ++k) accW0[k][t][r] = 0.0;
        }
#pragma unroll
    for (int l = 0; l < NM; ++l)
#pragma unroll
        for (int r = 0; r < 4; ++r) dbacc[l][r] = 0.0;
    // B operand "every feature = 1.0" (digit 5 = 64, all others 0): its products with the digits of dZ are the row sums of dZ
    const v4i ones5 = {0x40404040, 0x40404040, 0x40404040, 0x40404040};

    auto load_frags = [&](v4i (&Af)[NS], const unsigned char* tile) {
#pragma unroll
        for (int wi = 0; wi < NS; ++wi) Af[wi] = *reinterpret_cast<const v4i*>(tile + wi * SLICE_BYTES);
    };
    auto products = [&](v4i (&acc)[NLEV], const v4i (&Af)[NS], const v4i (&Bf)[NS]) {
#pragma unroll
        for (int k = 0; k < NPROD; ++k) issue_product<LMIN>(k, acc, Af, Bf);
    };

    // one tile's worth of matrix work: 6 fragment reads + the kept digit products into `acc`
    auto mfma_tile = [&](v4i (&acc)[NLEV], const unsigned char* tile, const v4i (&Bop)[NS]) {
        v4i Af[NS];
        load_frags(Af, tile);
        products(acc, Af, Bop);
    };
    for (int it = 0; it < a.iters; ++it) {
        const int n = n_n;
        const bool valid = valid_n;
        double xk[DD];
#pragma unroll
        for (int k = 0; k < DD; ++k) {
            xk[k] = xn[k];
            bad_run |= TANH ? !qn_bounded(xk[k]) : !bounded100(xk[k]);
        }
        const double yv = yn;
        if (it + 1 < a.iters) fetch(it + 1);
        QN_STAMP(0);
        // ------------------------------------------------------------------ forward
        double act[NH][T][4];
        v4i Bd[NM][NS];
        double rsrow = 1.0;                                             // relu / identity: 2^f_n, this row's scale of the forward B operand
        // relu / identity: the row's 64 activations of layer `la` (16 in this lane) -> B operand with the row's own scale
        auto slice_row = [&](int la) {
            double m = 0.0;
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmax(m, fabs(act[la][t][r]));
            m = fmax(m, __shfl_xor(m, 16, 64));
            m = fmax(m, __shfl_xor(m, 32, 64));
            int E = (__double2hiint(m) >> 20) & 0x7ff;                  // |v| < 2^(E - 1022) for every v of the row
            E = E < 122 ? 122 : E;
            const double sl = __hiloint2double((2091 - E) << 20, 0);    // 2^(46 - f), f = E - 1022
            rsrow = __hiloint2double((E + 1) << 20, 0);                 // 2^f
#pragma unroll
            for (int t = 0; t < T; ++t) {
                int S[NS];
                slice4_scaled(act[la][t], sl, magicS, S);
#pragma unroll
                for (int k = 0; k < NS; ++k) Bd[la][k][t] = S[k];
            }
        };
        {
            int top = 0;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                double z[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) z[r] = lds[offb0 + 16 * t + 4 * q + r];
#pragma unroll
                for (int k = 0; k < DD; ++k)
#pragma unroll
                    for (int r = 0; r < 4; ++r) z[r] = fma(lds[(16 * t + 4 * q + r) * DP + k], xk[k], z[r]);
                if constexpr (TANH) {
                    tanh_tab64_n<4, SHORT_TAB>(z, act[0][t], tanh_tab, magic52, cm13);
                    int S[NS];
                    slice4_scaled(act[0][t], 0x1p46, magicS, S);
#pragma unroll
                    for (int k = 0; k < NS; ++k) Bd[0][k][t] = S[k];
                    top |= top_digits_large(S[NS - 1]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) act[0][t][r] = ACT == QN_ACT_RELU ? fmax(z[r], 0.0) : z[r];
                }
            }
            if constexpr (TANH) bad_run |= !__any(top != 0);            // all activations of the wave's rows tiny: see qn_i8_slice.h
            else slice_row(0);
        }
        QN_STAMP(1);
        // hidden layers: the products of tile t + 1 are issued BETWEEN the vector instructions of tile t's epilogue (one wave
        // per SIMD: a burst of MFMAs holds the wave's in-order issue, a run of vector instructions leaves the matrix pipe idle)
        double pd = 0.0;
#pragma unroll
        for (int l = 1; l < NH; ++l) {
            const unsigned char* plane = wq + (l - 1) * LAYER_BYTES + lofs;
            const double* sb = lds + offsb + (l - 1) * 2 * H + 2 * 4 * q;
            v4i accs[2][NLEV];
            int top = 0;
            mfma_tile(accs[0], plane, Bd[l - 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                if (t + 1 < T) mfma_tile(accs[(t + 1) & 1], plane + (t + 1) * 16 * H, Bd[l - 1]);
                {
                    double2 sc[4];
                    double ts[4], z[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[r] = *reinterpret_cast<const double2*>(sb + 32 * t + 2 * r);
                    recombine4<NLEV>(accs[t & 1], ts, c65536);
#pragma unroll
                    for (int r = 0; r < 4; ++r) z[r] = TANH ? fma(ts[r], sc[r].x, sc[r].y) : fma(ts[r] * rsrow, sc[r].x, sc[r].y);
                    if constexpr (TANH) {
                        tanh_tab64_n<4, SHORT_TAB>(z, act[l][t], tanh_tab, magic52, cm13);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) act[l][t][r] = ACT == QN_ACT_RELU ? fmax(z[r], 0.0) : z[r];
                    }
                    if (l == NM) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) pd = fma(lds[offWl + 16 * t + 4 * q + r], act[l][t][r], pd);
                    }
                }
                if constexpr (TANH) {
                    if (l < NM) {
                        int S[NS];
                        slice4_scaled(act[l][t], 0x1p46, magicS, S);
#pragma unroll
                        for (int k = 0; k < NS; ++k) Bd[l][k][t] = S[k];
                        top |= top_digits_large(S[NS - 1]);
                    }
                }
                if (t + 1 < T) interleave_hint<NPROD, TANH ? QN_BWD8_VPM_FWD : 2>();
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (TANH) {
                if (l < NM) bad_run |= !__any(top != 0);
            } else {
                if (l < NM) slice_row(l);                               // (this layer's products are done: rsrow may change)
            }
        }
        QN_STAMP(2);
        // ------------------------------------------------------------------ last layer, residual
        double delta;
        {
            pd += __shfl_xor(pd, 16, 64);
            pd += __shfl_xor(pd, 32, 64);
            const double pr = pd + lds[offbl];
            const double res = pr - yv;
            delta = valid ? 2.0 * res : 0.0;
            if (valid && q == 0) {
                sse += res * res;
                if (pred_out) pred_out[(int64_t)b * a.Nb + n] = pr;
            }
        }
        // ------------------------------------------------------------------ backward
        double dzb[2][T][4];                                            // dZ_{l+1} in dzb[(NH - 1 - l) & 1] (no copies between layers)
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double av = act[NH - 1][t][r];
                accWl[t][r] = fma(delta, av, accWl[t][r]);
                dzb[0][t][r] = (lds[offWl + 16 * t + 4 * q + r] * delta) *
                               (TANH ? fma(-av, av, 1.0) : ACT == QN_ACT_RELU ? (av > 0.0 ? 1.0 : 0.0) : 1.0);
            }
        if (q == 0) accBl += delta;
        QN_STAMP(3);
#pragma unroll
        for (int l = NH - 1; l >= 1; --l) {
            // dz = dZ_{l+1} (gradient at the pre-activations of a_{l+1}); matrix W_l = planes l - 1; a_l = act[l - 1]
            double (&dz)[T][4] = dzb[(NH - 1 - l) & 1];
            double (&dzn)[T][4] = dzb[(NH - l) & 1];
            unsigned ex = 0;
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) ex = max(ex, ((unsigned)__double2hiint(dz[t][r]) & 0x7fffffffu) >> 20);
            ex = wave_max_u32(ex);
            if (lane == 0) gx[wave] = (int)ex;
            if constexpr (!TANH) {                                      // exponent maximum of the wave's a_l (dW operand: one scale per 64 rows)
                unsigned exa = 0;
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) exa = max(exa, ((unsigned)__double2hiint(act[l - 1][t][r]) & 0x7fffffffu) >> 20);
                exa = wave_max_u32(exa);
                if (lane == 0) gx[4 + wave] = (int)exa;
            }
            // the transposed digit words of a_l need nothing from the other waves: formed ahead of the barrier (where the
            // wave would wait anyway), written behind it  (tanh; relu / identity slice a_l behind the barrier, with the workgroup's scale)
            int At[NS * T];
            if constexpr (TANH) {
                int Ain[NS * T];
#pragma unroll
                for (int k = 0; k < NS; ++k)
#pragma unroll
                    for (int t = 0; t < T; ++t) Ain[k * T + t] = Bd[l - 1][k][t];
                quad_xpose_n<NS * T>(Ain, At, selA, selB);
            }
            __syncthreads();                                            // exponents visible; the previous stash readers are done
            QN_STAMP(4);
            int E = max(max(gx[0], gx[1]), max(gx[2], gx[3]));          // |dZ| < 2^(E - 1022) for the workgroup's 64 rows
            bad_run |= E >= 1022 + 500;
            E = E < 122 ? 122 : (E > 1600 ? 1600 : E);
            const double sl = __hiloint2double((2091 - E) << 20, 0);    // 2^(46 - G), G = E - 1022
            const double rs = __hiloint2double((E + 1) << 20, 0);       // 2^G
            double rsa = 1.0;                                           // relu / identity: 2^Ga, the scale of the dW product's a_l operand
            if constexpr (!TANH) {
                int Ea = max(max(gx[4], gx[5]), max(gx[6], gx[7]));
                Ea = Ea < 122 ? 122 : Ea;
                const double sla = __hiloint2double((2091 - Ea) << 20, 0);
                rsa = __hiloint2double((Ea + 1) << 20, 0);
                int Ain[NS * T];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    int S[NS];
                    slice4_scaled(act[l - 1][t], sla, magicS, S);
#pragma unroll
                    for (int k = 0; k < NS; ++k) Ain[k * T + t] = S[k];
                }
                quad_xpose_n<NS * T>(Ain, At, selA, selB);
            }
#pragma unroll
            for (int k = 0; k < NS; ++k)
#pragma unroll
                for (int t = 0; t < T; ++t) *reinterpret_cast<int*>(SA + k * SLICE_BYTES + t * 16 * H + wofs) = At[k * T + t];
            v4i D[NS];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                int S[NS], St[NS];
                slice4_scaled(dz[t], sl, magicS, S);
                quad_xpose_n<NS>(S, St, selA, selB);
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    D[k][t] = S[k];
                    *reinterpret_cast<int*>(SD + k * SLICE_BYTES + t * 16 * H + wofs) = St[k];
                }
            }
            QN_STAMP(5);
            __syncthreads();
            QN_STAMP(6);
            // ---- one pipeline of nine product groups: dW_l (4 column tiles: rows 16 wave .. + 15 of dZ x features of a_l, K = the
            // 64 data rows), db_l (the six products of dZ with the constant top digit of 1.0: row sums), dA = W_l^T dZ (4 tiles);
            // each group's products are issued between the vector instructions of the previous group's epilogue
            const double sdb = rs * __hiloint2double((1023 + 8 * LMIN - 2 * QB) << 20, 0);      // digits at 2^-46 each, levels in units of 256^LMIN
            const double sdw = sdb * rsa;                                                       // (the a_l operand's own scale; the row-sum operand "1.0" has none)
            const unsigned char* planeT = wqT + (l - 1) * LAYER_BYTES + lofs;
            const double* sT = lds + offsT + (l - 1) * H + 4 * q;
            v4i Az[NS], accs[2][NLEV], accb[NS];
            load_frags(Az, SD + wave * 16 * H + lofs);
            auto dw_group = [&](v4i (&acc)[NLEV], int ti) {
                v4i Bf[NS];
                load_frags(Bf, SA + ti * 16 * H + lofs);
                products(acc, Az, Bf);
            };
            auto dw_epilogue = [&](const v4i (&acc)[NLEV], int ti) {
                double ts[4];
                recombine4<NLEV>(acc, ts, c65536);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dWacc[l - 1][ti][r] = fma(ts[r], sdw, dWacc[l - 1][ti][r]);
                    pin(dWacc[l - 1][ti][r]);
                }
            };
            auto da_epilogue = [&](const v4i (&acc)[NLEV], int t) {
                double ts[4], g[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) g[r] = sT[16 * t + r] * rs;
                recombine4<NLEV>(acc, ts, c65536);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    g[r] *= TANH ? fma(-act[l - 1][t][r], act[l - 1][t][r], 1.0) : ACT == QN_ACT_RELU ? (act[l - 1][t][r] > 0.0 ? 1.0 : 0.0) : 1.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dzn[t][r] = ts[r] * g[r];
                    pin(dzn[t][r]);
                }
            };
            dw_group(accs[0], 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
                if (ti + 1 < 4) {
                    dw_group(accs[(ti + 1) & 1], ti + 1);
                } else {
#pragma unroll
                    for (int wi = 0; wi < NS; ++wi) accb[wi] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Az[wi], ones5, (v4i){0, 0, 0, 0}, 0, 0, 0);
                }
                dw_epilogue(accs[ti & 1], ti);
                if (ti + 1 < 4) interleave_hint<NPROD, QN_BWD8_VPM_DW>();
                else interleave_hint_nolds<NS, 8>();
                __builtin_amdgcn_sched_barrier(0);
            }
            {   // first tile of dA under the epilogue of db
                mfma_tile(accs[0], planeT, D);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double ts = (double)(accb[4][r] + (accb[5][r] << 8));
                    ts = fma(ts, 65536.0, (double)(accb[2][r] + (accb[3][r] << 8)));
                    ts = fma(ts, 65536.0, (double)(accb[0][r] + (accb[1][r] << 8)));
                    dbacc[l - 1][r] = fma(ts, sdb * (double)(1 << (8 * (5 - LMIN))), dbacc[l - 1][r]);
                    pin(dbacc[l - 1][r]);
                }
                interleave_hint<NPROD, 2>();
                __builtin_amdgcn_sched_barrier(0);
            }
            QN_STAMP(7);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                if (t + 1 < T) mfma_tile(accs[(t + 1) & 1], planeT + (t + 1) * 16 * H, D);
                da_epilogue(accs[t & 1], t);
                if (t + 1 < T) interleave_hint<NPROD, QN_BWD8_VPM_DA>();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        double (&dz)[T][4] = dzb[(NH - 1) & 1];                          // dZ_1
        // ------------------------------------------------------------------ first layer
        QN_STAMP(8);
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                accB0[t][r] += dz[t][r];
#pragma unroll
                for (int k = 0; k < DD; ++k) accW0[k][t][r] = fma(dz[t][r], xk[k], accW0[k][t][r]);
            }
        QN_STAMP(9);
    }

#ifdef QN_BWD8_STAMPS
    if (lane == 0 && blockIdx.x == 0) {
        long long* dbg = reinterpret_cast<long long*>(partial + a.dbg_off) + 12 * wave;
        for (int k = 0; k < 12; ++k) dbg[k] = stamp_acc[k];
    }
#endif
    // ---------------------------------------------------------------------- write the partial gradient
    double* out = slab + ((int64_t)b * a.nsplit + split) * a.p;
    const int64_t gb0 = (int64_t)H * d, gHH = gb0 + nb * H;
    const int64_t gWl = gHH + (int64_t)NM * (H * H + nb * H), gbl = gWl + H;
#pragma unroll
    for (int l = 0; l < NM; ++l) {
        double* og = out + gHH + (int64_t)l * (H * H + nb * H);
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) og[(int64_t)(16 * wave + 4 * q + r) * H + 16 * ti + c] = dWacc[l][ti][r];
    }
    // lane-local sums: over the 16 row lanes of each lane group, then over the 4 waves through LDS (the stashes are free)
    __syncthreads();
    constexpr int NV = 2 + DD;                                          // vectors of 64: b0, Wl, the DP columns of W0
    double* R = reinterpret_cast<double*>(SD);                          // [NV][4 waves][64] + [4]
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double v[NV];
            v[0] = accB0[t][r];
            v[1] = accWl[t][r];
#pragma unroll
            for (int k = 0; k < DD; ++k) v[2 + k] = accW0[k][t][r];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1)
#pragma unroll
                for (int u = 0; u < NV; ++u) v[u] += __shfl_xor(v[u], m, 64);
            if (c == 0) {
                const int f = 16 * t + 4 * q + r;
#pragma unroll
                for (int u = 0; u < NV; ++u) R[(u * 4 + wave) * H + f] = v[u];
            }
        }
    const double bl_w = wave_sum(accBl);
    sse = wave_sum(sse);
    if (lane == 0) {
        R[NV * 4 * H + wave] = bl_w;
        red[wave] = sse;
    }
    const int any_bad = __any(bad_run) ? 1 : 0;
    if (lane == 0) gx[wave] = any_bad;
    __syncthreads();
    if (tid < H) {
        double s[NV];
#pragma unroll
        for (int u = 0; u < NV; ++u) s[u] = ((R[(u * 4 + 0) * H + tid] + R[(u * 4 + 1) * H + tid]) + R[(u * 4 + 2) * H + tid]) + R[(u * 4 + 3) * H + tid];
        if (nb) out[gb0 + tid] = s[0];
        out[gWl + tid] = s[1];
#pragma unroll
        for (int k = 0; k < DD; ++k) out[(int64_t)tid * DD + k] = s[2 + k];
    }
    if (nb && c == 0) {
#pragma unroll
        for (int l = 0; l < NM; ++l)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[gHH + (int64_t)l * (H * H + H) + H * H + 16 * wave + 4 * q + r] = dbacc[l][r];
    }
    if (tid == 0) {
        if (nb) out[gbl] = ((R[NV * 4 * H] + R[NV * 4 * H + 1]) + R[NV * 4 * H + 2]) + R[NV * 4 * H + 3];
        partial[(int64_t)b * a.nsplit + split] = (red[0] + red[1]) + (red[2] + red[3]);
        flags[b * a.nsplit + split] = gx[0] | gx[1] | gx[2] | gx[3];
    }
}

}  // namespace

// ---- what qn_fused.hip needs to dispatch to this kernel
bool qn_fused_bwd_i8_applies(int Hh, int nhid, int act, int d, int o) {
#ifdef QN_I8_TANH_ONLY
    if (act != QN_ACT_TANH) return false;                      // A/B builds: relu on the float64-MFMA kernel
#endif
    // (identity networks -- linear models -- keep the float64-MFMA kernel: not worth a third set of instances)
    return Hh == H && (act == QN_ACT_TANH || act == QN_ACT_RELU) && (nhid == 2 || nhid == 3) && d >= 1 && d <= 4 && o == 1;
}
size_t qn_fused_bwd_i8_lds_bytes(int nhid, int d) { return bwd_lds_bytes(d <= 2 ? 2 : 4, nhid); }
template <int ACT>
static qn_bwd_i8_fn pick_bwd_i8(int nhid, int d) {
    if (d == 1) return nhid == 2 ? k_fused_bwd_i8<2, 1, QN_I8_LMIN, ACT> : k_fused_bwd_i8<3, 1, QN_I8_LMIN, ACT>;
    if (d == 2) return nhid == 2 ? k_fused_bwd_i8<2, 2, QN_I8_LMIN, ACT> : k_fused_bwd_i8<3, 2, QN_I8_LMIN, ACT>;
    if (d == 3) return nhid == 2 ? k_fused_bwd_i8<2, 3, QN_I8_LMIN, ACT> : k_fused_bwd_i8<3, 3, QN_I8_LMIN, ACT>;
    return nhid == 2 ? k_fused_bwd_i8<2, 4, QN_I8_LMIN, ACT> : k_fused_bwd_i8<3, 4, QN_I8_LMIN, ACT>;
}
qn_bwd_i8_fn qn_fused_bwd_i8_kernel(int nhid, int d, int act) {
    return act == QN_ACT_RELU ? pick_bwd_i8<QN_ACT_RELU>(nhid, d) : pick_bwd_i8<QN_ACT_TANH>(nhid, d);
}
static_assert(bwd_lds_bytes(4, 3) <= 160 * 1024 && bwd_lds_bytes(2, 3) <= 160 * 1024, "the LDS image of the three-hidden-layer kernel has to fit a CU");
