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
// accumulator -> float64 (tanh) -> digits -> operand without leaving the lane.  First layer (d <= 8: DP = 2, 4 or 8 padded input columns) and last layer
// (o <= 4) on the VALU as in k_fused_fwd_f64; the last hidden layer's outputs are consumed as float64 (no slicing).
//
// Anything that could make a NaN / inf (a weight or input that is not finite and < 2^500) leaves the fast path: the
// wave evaluates its 32 rows with a plain float64 loop from the original weights (slow_tile), IEEE semantics as the
// layer-wise kernels.
#include "qn_common.h"
#include "qn_fused_args.h"
#include "qn_math.h"
#include "qn_i8_slice.h"
#include <mutex>
#include <type_traits>
#include <unordered_set>
#include <utility>

#ifndef QN_I8_G
#define QN_I8_G 1              // 16-row groups per wave iteration (register budget of the pipelined epilogue: one)
#endif
#ifndef QN_I8_VPM
#define QN_I8_VPM 7            // vector instructions scheduled behind each MFMA of the pipelined epilogue
#endif

namespace {

// Stage chain `Wb`: thin layers as float64, hidden matrices as digit planes, the tanh table.  Returns whether this thread saw
// a weight that is not finite and < 2^500.
// Every global load of the staging is issued before the first result is used, the next hidden matrix's while the current
// one is sliced: the first version fetched the table, the thin pieces and each matrix one after the other -- five memory
// round trips, most of the ~6.5 us a workgroup spent here (8 % of a cfg2 launch; with 8 chains per launch, where a
// workgroup has ONE iteration of rows behind its staging, 30 %).
// |v| < 2^20 (and not NaN): the bound on EVERY weight of a relu / identity network (its activations are not bounded by 1: with
// weights below 2^20, inputs below 2^100 and at most 15 layers no product or 64-term sum can overflow -- 2^(100 + 15 x 26))
__device__ __forceinline__ bool bounded20(double v) { return (unsigned)(__double2hiint(v) & 0x7fffffff) < 0x41300000u; }
__device__ __forceinline__ bool bounded100(double v) { return (unsigned)(__double2hiint(v) & 0x7fffffff) < 0x46300000u; }

template <int DP, int LMIN, bool UNB = false>
__device__ __forceinline__ int stage(double* __restrict__ lds, unsigned char* __restrict__ wq, double* __restrict__ tanh_tab,
                                     const double* __restrict__ Wb, const FusedArgs& a, long long* stamps = nullptr) {
    int bad = 0;
    auto chk = [&](double v) { bad |= UNB ? !bounded20(v) : !qn_bounded(v); return v; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = a.d, o = a.o, nb = a.has_bias ? 1 : 0;
    const int64_t gb0 = (int64_t)H * d, gHH = gb0 + nb * H;
    const int64_t gWl = gHH + (int64_t)(a.nhid - 1) * (H * H + nb * H), gbl = gWl + (int64_t)o * H;
    const int lb0 = H * DP, lWl = lb0 + H, lbl = lWl + OMAX * H, lsb = thin_doubles(DP);
    const int q16 = lane & 15, m4 = q16 >> 2, g4 = q16 & 3;                // quad (m, g): features 16 m + 4 g + {0..3}
    // hidden matrices: a thread takes 4 consecutive input features of one row = the 4 bytes of one dword of every digit
    // plane (k-slot map of the header), the 16 lanes of a DPP row take one matrix row: the row's largest exponent is 4
    // DPP steps, the digits come out of slice4 already packed, 6 ds_write_b32 per item (byte stores of single digits
    // were 4-way bank conflicts and, with a ds_bpermute row maximum, 19 % of the kernel)
    auto load_matrix = [&](int layer, double (&v)[4][4]) {
        const double* Wg = Wb + gHH + (int64_t)(layer - 1) * (H * H + nb * H);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = 16 * u + 4 * wave + (lane >> 4);
            const double2* src = reinterpret_cast<const double2*>(Wg + row * H + 16 * m4 + 4 * g4);
            const double2 v01 = src[0], v23 = src[1];
            v[u][0] = v01.x; v[u][1] = v01.y; v[u][2] = v23.x; v[u][3] = v23.y;
        }
    };
    // ---- issue: first hidden matrix, thin pieces, table
    double vA[4][4], vB[4][4];                                            // (two named sets: a runtime-indexed array would live in scratch)
    load_matrix(1, vA);
    constexpr int TPT = (QN_TANH_TAB64_N + WGT - 1) / WGT;
    double tt[TPT];
#pragma unroll
    for (int k = 0; k < TPT; ++k) tt[k] = tid + k * WGT < QN_TANH_TAB64_N ? qn_tanh_table64_g[tid + k * WGT] : 0.0;
    constexpr int W0PT = (H * DP + WGT - 1) / WGT;                         // 1
    double w0v[W0PT], wlv, b0v = 0.0, blv = 0.0;
#pragma unroll
    for (int k = 0; k < W0PT; ++k) {
        const int e = tid + k * WGT, j = e / DP, kk = e % DP;
        w0v[k] = (e < H * DP && kk < d) ? Wb[j * d + kk] : 0.0;
    }
    wlv = tid < o * H ? Wb[gWl + tid] : 0.0;                               // (OMAX * H = WGT)
    if (tid < H && nb) b0v = Wb[gb0 + tid];
    if (tid < OMAX && tid < o && nb) blv = Wb[gbl + tid];
#ifdef QN_FWD8_STAMPS
    stamps[0] = __builtin_amdgcn_s_memrealtime();                        // loads issued
#endif
    // ---- consume
#pragma unroll
    for (int k = 0; k < TPT; ++k)
        if (tid + k * WGT < QN_TANH_TAB64_N) tanh_tab[tid + k * WGT] = tt[k];
#pragma unroll
    for (int k = 0; k < W0PT; ++k)
        if (tid + k * WGT < H * DP) lds[tid + k * WGT] = chk(w0v[k]);
    lds[lWl + tid] = chk(wlv);
    if (tid < H) lds[lb0 + tid] = chk(b0v);
    if (tid < OMAX) lds[lbl + tid] = chk(blv);
#ifdef QN_FWD8_STAMPS
    __builtin_amdgcn_s_waitcnt(0x0F70);
    stamps[1] = __builtin_amdgcn_s_memrealtime();                        // table + thin pieces have arrived
#endif
    // (step-major over the thread's four items: exponents, row maxima, scaling, digits, stores -- written item by item the
    // compiler keeps each item's dependent chain together and a matrix took 2.3 us per workgroup, three times its issue time)
    auto slice_matrix = [&](int layer, const double (&v)[4][4]) {
        const double* Wg = Wb + gHH + (int64_t)(layer - 1) * (H * H + nb * H);
        unsigned char* plane = wq + (layer - 1) * LAYER_BYTES;
        double* sb = lds + lsb + (layer - 1) * 2 * H;
        double bias[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) bias[u] = (q16 == 0 && nb) ? Wg[H * H + 16 * u + 4 * wave + (lane >> 4)] : 0.0;
        unsigned ex[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ex[u] = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                chk(v[u][r]);
                ex[u] = max(ex[u], ((unsigned)__double2hiint(v[u][r]) & 0x7fffffffu) >> 20);
            }
        }
        row16_max_u32_n<4>(ex);
        int e[4];
        double an[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // 2^e > every |W_ji| of the row, from the largest biased exponent field E: |w| < 2^(E - 1022)
            e[u] = (int)ex[u] - 1022;
            bad |= e[u] > I8_MAX_WEIGHT_EXP;                           // (an outlier weight: see qn_i8_slice.h)
            e[u] = e[u] < -900 ? -900 : e[u];                          // (all-zero / denormal rows: any scale will do)
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) an[u][r] = ldexp(v[u][r], -e[u]);    // exact, |an| < 1
        int S[4][NS];
        slice4_n<4>(an, S);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = 16 * u + 4 * wave + (lane >> 4);
            unsigned char* dst = plane + row * H + 16 * (g4 ^ slot_swz(row)) + 4 * m4;
#pragma unroll
            for (int k = 0; k < NS; ++k) *reinterpret_cast<int*>(dst + k * SLICE_BYTES) = S[u][k];
            if (q16 == 0) {
                sb[2 * row] = ldexp(1.0, e[u] - 2 * QB + 8 * LMIN);    // integer sum (in units of 256^LMIN) -> W_j . a
                sb[2 * row + 1] = chk(bias[u]);
            }
        }
    };
#ifdef QN_I8_SKIP_STAGE
    if (a.nhid > 100)             // timing-only build: what the digit staging costs
#endif
    for (int layer = 1; layer < a.nhid; layer += 2) {                     // the next matrix is in flight while this one is sliced
        if (layer + 1 < a.nhid) load_matrix(layer + 1, vB);
        slice_matrix(layer, vA);
#ifdef QN_FWD8_STAMPS
        stamps[2] = __builtin_amdgcn_s_memrealtime();                    // first matrix sliced
#endif
        if (layer + 1 < a.nhid) {
            if (layer + 2 < a.nhid) load_matrix(layer + 2, vA);
            slice_matrix(layer + 1, vB);
        }
    }
    return bad;
}

// The wave's 32 rows in plain float64 from the ORIGINAL weights (rare: non-finite / huge weights or inputs).  Lane j
// owns feature j of one row at a time; the previous layer's activations are broadcast through `scr` (128 doubles).
template <int DP, int ACT = QN_ACT_TANH>
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
        auto actf = [](double v) { return ACT == QN_ACT_TANH ? qn_tanh_f64(v) : ACT == QN_ACT_RELU ? qn_relu<double>(v) : v; };
        double act = actf(z);
        for (int layer = 1; layer < nhid; ++layer) {
            double* cur = scr + 64 * (layer & 1);
            cur[lane] = act;
            __builtin_amdgcn_s_waitcnt(0xC07F);                     // lgkmcnt(0): the wave's own LDS writes have landed
            const double* Wg = Wb + gHH + (int64_t)(layer - 1) * (H * H + nb * H);
            z = nb ? Wg[H * H + lane] : 0.0;
            for (int i = 0; i < H; ++i) z = fma(Wg[lane * H + i], cur[i], z);
            act = actf(z);
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

// ACT = tanh: activations in [-1, 1], sliced with the fixed scale 2^-46 inside each tile's epilogue.  ACT = relu / identity
// (round 4; the reference's DEFAULT activation is relu, quinn/nns/mlp.py:23): activations are unbounded, so every data row gets
// its own scale 2^f_n > max_j |a_j(n)| over the layer's 64 features -- a layer's 16 outputs per lane stay float64 until its last
// tile is done (row maximum in-lane over the 4 tiles, then across the 4 lane groups), are sliced then (slice_rows), and the next
// layer's integer sums are multiplied by 2^f_n.  Same norm-wise 47-bit bound per row; no tanh table, no tiny-activation rule.
template <int DP, int LMIN, int OM, int ACT = QN_ACT_TANH>
__global__ __launch_bounds__(WGT, 2) void k_fused_fwd_i8(FusedArgs a, const double* __restrict__ W,
                                                        const double* __restrict__ X, const double* __restrict__ Y,
                                                        const int32_t* __restrict__ row_idx, double* __restrict__ pred_out,
                                                        double* __restrict__ partial, unsigned long long* __restrict__ arrive,
                                                        double* __restrict__ sse_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NLEV = 2 * (NS - 1) - LMIN + 1;       // levels LMIN .. 10
    constexpr bool TANH = ACT == QN_ACT_TANH;
    double* lds = reinterpret_cast<double*>(smem);
    const int NH = a.nhid, d = a.d, o = a.o;
    int b, split;
    if (!qn_fused_wg(a.nsplit, a.B, &b, &split)) return;
    const int offb0 = H * DP, offWl = offb0 + H, offbl = offWl + OMAX * H, offred = offbl + OMAX, offsb = thin_doubles(DP);
    double* tanh_tab = lds + ((offsb + (NH - 1) * 2 * H + 1) & ~1);
    double* scratch = tanh_tab + ((TANH_TAB + 1) & ~1);
    unsigned char* wq = reinterpret_cast<unsigned char*>(lds + head_doubles(DP, NH));
    double* red = lds + offred;
    const double* Wb = W + (int64_t)b * a.p;
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
                xbad_n |= TANH ? !qn_bounded(xn[g][k]) : !bounded100(xn[g][k]);
            }
#pragma unroll
            for (int qo = 0; qo < OM; ++qo) yn[g][qo] = qo < o ? Y[rr * o + qo] : 0.0;
        }
    };
#ifdef QN_FWD8_STAMPS
    const long long st0 = __builtin_readcyclecounter();          // diagnostic build (tools/fwd8_stamps.py): s_memrealtime-like clock per workgroup
    const long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    double magic52 = 6755399441055744.0, magicS = kMagic;       // rounding constants as opaque register pairs (see the epilogue)
    asm volatile("" : "+v"(magic52), "+v"(magicS));
    // 32-bit constants as SCALAR registers the compiler cannot see through: as literals they make every instruction 8 bytes,
    // 7.2-7.5 issue cycles instead of 5.2 (tools/ubench_xpose.hip) -- 96 recombination fma's (65536.0), 48 residual fma's
    // (-1/64) and 42 digit xor's per 16 rows
    int k80;
    asm volatile("s_mov_b32 %0, 0x80808080" : "=s"(k80));
    fetch(0);                                                   // (in flight during the staging)
#ifdef QN_FWD8_STAMPS
    long long sst[4] = {0, 0, 0, 0};
    const int sbad = stage<DP, LMIN, !TANH>(lds, wq, tanh_tab, Wb, a, sst);
    sst[3] = __builtin_amdgcn_s_memrealtime();                           // second matrix sliced
    const bool w_bad = block_or(sbad, red + 6);
    const long long rt1 = __builtin_amdgcn_s_memrealtime();
#else
    const bool w_bad = block_or(stage<DP, LMIN, !TANH>(lds, wq, tanh_tab, Wb, a), red + 6);
#endif
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
            sse += slow_tile<DP, ACT>(a.Nb, a.d, a.o, a.nhid, a.has_bias, lds, scratch + 128 * wave, Wb, X, Y, row_idx,
                                 split * a.rows_per_split + (it * (WGT / 64) + wave) * 16 * G, b, pred_out);
            continue;
        }
        // ---- first layer (VALU): a_1 = tanh(W0 x + b0), sliced straight into the B operand
        v4i Bc[G][NS];
        int top = 0;                                                    // OR of the layer's top digits (see below)
        double rs[G];                                                   // relu / identity: 2^f_n, the row scale of the current B operand
        double Vr[TANH ? 1 : G][TANH ? 1 : T][4];                       // relu / identity: a layer's outputs until its row maximum is known
        double amax[G];
#pragma unroll
        for (int g = 0; g < G; ++g) { rs[g] = 1.0; amax[g] = 0.0; }
        // relu / identity: row maximum over the 4 lane groups -> exponent -> digits of the row's 64 activations (B operand)
        auto slice_rows = [&](int g, v4i (&Bo)[NS]) {
            if constexpr (!TANH) {
                double m = amax[g];
                m = fmax(m, __shfl_xor(m, 16, 64));
                m = fmax(m, __shfl_xor(m, 32, 64));
                int E = (__double2hiint(m) >> 20) & 0x7ff;             // |v| < 2^(E - 1022) for every v of the row
                E = E < 122 ? 122 : E;
                const double sl = __hiloint2double((2091 - E) << 20, 0);        // 2^(46 - f), f = E - 1022
                rs[g] = __hiloint2double((E + 1) << 20, 0);                     // 2^f
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    int lo[4], hi[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double x = fma(Vr[g][t][r], sl, kMagic);
                        lo[r] = __double2loint(x);
                        hi[r] = __double2hiint(x);
                    }
                    const int p01 = __builtin_amdgcn_perm(lo[1], lo[0], 0x05010400), q01 = __builtin_amdgcn_perm(lo[1], lo[0], 0x07030602);
                    const int p23 = __builtin_amdgcn_perm(lo[3], lo[2], 0x05010400), q23 = __builtin_amdgcn_perm(lo[3], lo[2], 0x07030602);
                    const int r01 = __builtin_amdgcn_perm(hi[1], hi[0], 0x05010400), r23 = __builtin_amdgcn_perm(hi[3], hi[2], 0x05010400);
                    Bo[0][t] = __builtin_amdgcn_perm(p23, p01, 0x05040100) ^ 0x80808080;
                    Bo[1][t] = __builtin_amdgcn_perm(p23, p01, 0x07060302) ^ 0x80808080;
                    Bo[2][t] = __builtin_amdgcn_perm(q23, q01, 0x05040100) ^ 0x80808080;
                    Bo[3][t] = __builtin_amdgcn_perm(q23, q01, 0x07060302) ^ 0x80808080;
                    Bo[4][t] = __builtin_amdgcn_perm(r23, r01, 0x05040100) ^ 0x80808080;
                    Bo[5][t] = __builtin_amdgcn_perm(r23, r01, 0x07060302);
                }
                amax[g] = 0.0;
            }
        };
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                double av[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * t + 4 * q + r;
                    double z = lds[offb0 + j];
#pragma unroll
                    for (int k = 0; k < DP; ++k) z = fma(lds[j * DP + k], xk[g][k], z);
                    if constexpr (TANH) av[r] = qn_tanh_f64_tab64(z, tanh_tab);
                    else {
                        av[r] = ACT == QN_ACT_RELU ? fmax(z, 0.0) : z;  // (finite here: unbounded weights / inputs took the plain loop)
                        Vr[g][t][r] = av[r];
                        amax[g] = fmax(amax[g], fabs(av[r]));
                    }
                }
                if constexpr (TANH) {
                    int S[NS];
                    slice4(av, S);
#pragma unroll
                    for (int k = 0; k < NS; ++k) Bc[g][k][t] = S[k];
                    top |= top_digits_large(S[NS - 1]);
                }
            }
            slice_rows(g, Bc[g]);
        }
        // Activations are sliced with the FIXED scale 2^-46 (tanh outputs lie in [-1, 1]): absolute error 2^-47, which is a
        // RELATIVE error only as long as a layer's activations are not all tiny.  A layer whose top digit is zero for every
        // activation of the wave's rows (top_digits_large: all |a| < ~2^-4.7, e.g. very small weights without bias) sends
        // the wave's rows through the plain float64 loop instead (slow_tile); otherwise the error stays <= 1.8e-13 of
        // the largest activation.  Costs two instructions per tile.
        bool redo = TANH && !__any(top != 0);
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
                            const v4i (&B)[NS], const double* sbt, const double* wlt, int (&S)[NS], double (&prt)[OM], double rsg,
                            double (&avo)[4]) {
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
                        if constexpr (NLEV >= 5) { constexpr int l = ((NLEV & 1) ? NLEV - 3 : NLEV - 4); { const double cv_ = (double)(acc[l][r] + (acc[l + 1][r] << 8)); asm("v_fma_f64 %0, %1, %2, %3" : "=v"(ts[r]) : "v"(ts[r]), "s"(65536.0), "v"(cv_)); } }
                        break;
                    case 2:
                        if constexpr (NLEV >= 5) { constexpr int l = ((NLEV & 1) ? NLEV - 5 : NLEV - 6); { const double cv_ = (double)(acc[l][r] + (acc[l + 1][r] << 8)); asm("v_fma_f64 %0, %1, %2, %3" : "=v"(ts[r]) : "v"(ts[r]), "s"(65536.0), "v"(cv_)); } }
                        break;
                    case 3:
                        if constexpr (NLEV == 7) { const double cv_ = (double)(acc[0][r] + (acc[1][r] << 8)); asm("v_fma_f64 %0, %1, %2, %3" : "=v"(ts[r]) : "v"(ts[r]), "s"(65536.0), "v"(cv_)); }
                        if constexpr (TANH) z[r] = fma(ts[r], sc[r].x, sc[r].y);
                        else z[r] = fma(ts[r] * rsg, sc[r].x, sc[r].y);       // (x 2^f_n: the row scale of the B operand)
                        break;
                    case 4:
                        if constexpr (!TANH) { av[r] = ACT == QN_ACT_RELU ? fmax(z[r], 0.0) : z[r]; break; }
                        asm("v_min_f64 %0, |%1|, %2" : "=v"(ax[r]) : "v"(z[r]), "s"(20.0));
                        // (one v_fma_f64 with the addend in an opaque register pair: for a known constant hipcc emits v_fmac_f64
                        // behind moves that re-materialise it -- 142 vector instructions fewer per 16 rows, +0.9 % A/B in one call)
                        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(zm[r]) : "v"(ax[r]), "s"(64.0), "v"(magic52));
                        break;
                    case 5:
                        if constexpr (!TANH) break;
                        Tt[r] = tanh_tab[__double2loint(zm[r])];
                        { const double nf_ = zm[r] - magic52; asm("v_fma_f64 %0, %1, %2, %3" : "=v"(bb[r]) : "v"(nf_), "s"(-0.015625), "v"(ax[r])); }
                        break;
                    case 6:
                        if constexpr (!TANH) break;
                        b2[r] = bb[r] * bb[r];
                        break;
                    case 7:
                        if constexpr (!TANH) break;
                        pp[r] = fma(b2[r], 1.33333333333333333e-01, -3.33333333333333333e-01);
                        b2[r] = bb[r] * b2[r];
                        break;
                    case 8:
                        if constexpr (!TANH) break;
                        tb[r] = fma(b2[r], pp[r], bb[r]);
                        break;
                    // (qn_tanh_f64_tab64 of qn_math.h, reciprocal-free tail: T + (1 - T^2) tb (1 - e)(1 + e^2 + e^4), e = T tb)
                    case 9:
                        if constexpr (!TANH) break;
                        num[r] = Tt[r] * tb[r];                       // e
                        den[r] = fma(-Tt[r], Tt[r], 1.0);             // 1 - T^2
                        break;
                    case 10:
                        if constexpr (!TANH) break;
                        e0[r] = num[r] * num[r];                      // e^2
                        y0[r] = fma(-tb[r], num[r], tb[r]);           // tb (1 - e)
                        break;
                    case 11:
                        if constexpr (!TANH) break;
                        e0[r] = fma(e0[r], e0[r], e0[r]);             // e^2 + e^4
                        break;
                    case 12:
                        if constexpr (!TANH) break;
                        y0[r] = fma(y0[r], e0[r], y0[r]);             // u
                        break;
                    case 13:
                        if constexpr (!TANH) break;
                        num[r] = fma(den[r], y0[r], Tt[r]);
                        break;
                    case 14:
                        if constexpr (!TANH) break;
                        av[r] = __builtin_copysign(num[r], z[r]);
                        break;

                    case 15:
                        if constexpr (LAST) {
#pragma unroll
                            for (int qo = 0; qo < OM; ++qo)
                                if (qo < o) prt[qo] = fma(wlt[qo * H + r], av[r], prt[qo]);
                        } else if constexpr (TANH) {
                            double x;
                            asm("v_fma_f64 %0, %1, %2, %3" : "=v"(x) : "v"(av[r]), "s"(0x1p46), "v"(magicS));
                            lo[r] = __double2loint(x);
                            hi[r] = __double2hiint(x);
                        } else {
                            avo[r] = av[r];
                        }
                        break;
                    default: break;
                    }
                }
                if (st == 0 && NEXT) load_frags(Af, tile_next);          // (no product before stage 2: the reads are in flight)
                if constexpr (!LAST && TANH) {
                    if (st == 16) {
                        p01 = __builtin_amdgcn_perm(lo[1], lo[0], 0x05010400); q01 = __builtin_amdgcn_perm(lo[1], lo[0], 0x07030602);
                        p23 = __builtin_amdgcn_perm(lo[3], lo[2], 0x05010400); q23 = __builtin_amdgcn_perm(lo[3], lo[2], 0x07030602);
                    }
                    if (st == 17) {
                        r01 = __builtin_amdgcn_perm(hi[1], hi[0], 0x05010400); r23 = __builtin_amdgcn_perm(hi[3], hi[2], 0x05010400);
                        S[0] = __builtin_amdgcn_perm(p23, p01, 0x05040100) ^ k80;
                        S[1] = __builtin_amdgcn_perm(p23, p01, 0x07060302) ^ k80;
                    }
                    if (st == 18) {
                        S[2] = __builtin_amdgcn_perm(q23, q01, 0x05040100) ^ k80;
                        S[3] = __builtin_amdgcn_perm(q23, q01, 0x07060302) ^ k80;
                    }
                    if (st == 19) {
                        S[4] = __builtin_amdgcn_perm(r23, r01, 0x05040100) ^ k80;
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
            int topl = 0;
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
                    double avo[4];
                    const unsigned char* nxt = plane + (t + 1) * 16 * H;
                    if (t == T - 1) {
                        if (t & 1) epilogue(last_tag, std::false_type{}, accB, accA, nxt, Bc[g], sb + 32 * t, wl + 16 * t, S, part[g], rs[g], avo);
                        else epilogue(last_tag, std::false_type{}, accA, accB, nxt, Bc[g], sb + 32 * t, wl + 16 * t, S, part[g], rs[g], avo);
                    } else {
                        if (t & 1) epilogue(last_tag, std::true_type{}, accB, accA, nxt, Bc[g], sb + 32 * t, wl + 16 * t, S, part[g], rs[g], avo);
                        else epilogue(last_tag, std::true_type{}, accA, accB, nxt, Bc[g], sb + 32 * t, wl + 16 * t, S, part[g], rs[g], avo);
                    }
                    if constexpr (!LAST && TANH) {
#pragma unroll
                        for (int k = 0; k < NS; ++k) Bn[k][t] = S[k];
                    }
                    if constexpr (!LAST && !TANH) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            Vr[g][t][r] = avo[r];
                            amax[g] = fmax(amax[g], fabs(avo[r]));
                        }
                    }
                }
                if constexpr (!LAST && TANH) {
#pragma unroll
                    for (int k = 0; k < NS; ++k) Bc[g][k] = Bn[k];
                    topl |= (top_digits_large(Bn[NS - 1][0]) | top_digits_large(Bn[NS - 1][1])) | (top_digits_large(Bn[NS - 1][2]) | top_digits_large(Bn[NS - 1][3]));
                }
                if constexpr (!LAST && !TANH) slice_rows(g, Bc[g]);   // (the layer's products are done: its B operand may be overwritten)
            }
            if constexpr (!LAST && TANH) redo |= !__any(topl != 0);
        };
        for (int layer = 1; layer < NH - 1; ++layer) hidden_layer(std::false_type{}, layer);
        hidden_layer(std::true_type{}, NH - 1);
        if (redo) {                                                     // wave-uniform; rare
            sse += slow_tile<DP, ACT>(a.Nb, a.d, a.o, a.nhid, a.has_bias, lds, scratch + 128 * wave, Wb, X, Y, row_idx,
                                 split * a.rows_per_split + (it * (WGT / 64) + wave) * 16 * G, b, pred_out);
            continue;
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
#ifdef QN_FWD8_STAMPS
    const long long rt2 = __builtin_amdgcn_s_memrealtime();
#endif
    sse = wave_sum(sse);
    if (lane == 0) red[wave] = sse;
    __syncthreads();
#ifdef QN_FWD8_STAMPS
    if (threadIdx.x == 0 && pred_out) {
        double* dbg = pred_out + (int64_t)a.B * a.Nb * a.o - 8 * (int64_t)gridDim.x + 8 * (int64_t)blockIdx.x;     // the tail of the prediction buffer
        dbg[0] = (double)rt0; dbg[1] = (double)rt1; dbg[2] = (double)rt2; dbg[3] = (double)__builtin_amdgcn_s_memrealtime();
        for (int k = 0; k < 4; ++k) dbg[4 + k] = (double)sst[k];
        (void)st0;
    }
#endif
    if (threadIdx.x < 64) qn_sse_finish(partial, arrive, sse_out, b, split, a.nsplit, (red[0] + red[1]) + (red[2] + red[3]));
}

}  // namespace

// ---- what qn_fused.hip needs to dispatch to this kernel
int qn_fused_i8_rows_per_iteration() { return (WGT / 64) * 16 * G; }
bool qn_fused_i8_applies(int Hh, int nhid, int act, int d, int o) {
    // (1..4 outputs: the 4-output instance took ~170 spilled registers in round 2; with the shorter tanh tail and the
    // constants out of the way it fits -- 216 registers, no scratch)
    if (Hh != H || (act != QN_ACT_TANH && act != QN_ACT_RELU && act != QN_ACT_IDENTITY) || nhid < 2 || d > 8 || o < 1 || o > OMAX) return false;
#ifdef QN_I8_TANH_ONLY
    if (act != QN_ACT_TANH) return false;                      // A/B builds: relu / identity on the float64-MFMA kernel
#endif
    if (act != QN_ACT_TANH && nhid > 15) return false;         // (the overflow bound of the unbounded activations: stage())
    return qn_fused_i8_lds_bytes(d, nhid) <= 160 * 1024;
}
size_t qn_fused_i8_lds_bytes(int d, int nhid) {
    const int dp = d <= 2 ? 2 : d <= 4 ? 4 : 8;
    return sizeof(double) * (size_t)head_doubles(dp, nhid) + (size_t)(nhid - 1) * LAYER_BYTES;
}
template <int ACT>
static qn_fwd_fn pick_i8(int d, int o) {
    // (5..8 inputs, round 4: the first layer is a VALU dot over DP padded input columns)
    if (d > 4) return o > 1 ? k_fused_fwd_i8<8, QN_I8_LMIN, OMAX, ACT> : k_fused_fwd_i8<8, QN_I8_LMIN, 1, ACT>;
    if (o > 1) return d <= 2 ? k_fused_fwd_i8<2, QN_I8_LMIN, OMAX, ACT> : k_fused_fwd_i8<4, QN_I8_LMIN, OMAX, ACT>;
    return d <= 2 ? k_fused_fwd_i8<2, QN_I8_LMIN, 1, ACT> : k_fused_fwd_i8<4, QN_I8_LMIN, 1, ACT>;
}
qn_fwd_fn qn_fused_i8_kernel(int d, int o, int act) {
    return act == QN_ACT_RELU ? pick_i8<QN_ACT_RELU>(d, o) : act == QN_ACT_IDENTITY ? pick_i8<QN_ACT_IDENTITY>(d, o) : pick_i8<QN_ACT_TANH>(d, o);
}

// =====================================================================================================================
// Layer-wise int8-slice FORWARD for wide tanh networks (hidden widths that are multiples of 64: cfg3..5, h = 128 / 256),
// where a chain's digit planes no longer fit LDS.  Same arithmetic as the fused kernel above, organised as a GEMM per
// layer with the activations kept in HBM twice: as float64 [B][h][Nb] (what the backward pass, the last layer and the
// exceptional path read) and as digit planes [B][6][Nb][h] bytes (row-major per data row, k-slot order inside every
// 64-feature chunk: the MFMA B operand of the next layer is 16 consecutive bytes per lane).
//   k_i8_slice_w   once per call: digit planes [6][h_out][h_in] (slot-swizzled per row, the LDS image of a tile) and
//                  {scale, bias} of every hidden->hidden matrix of every chain; flags chains with unbounded weights
//   k_i8_first     first layer on the VALU: a_1 = tanh(W0 x + b0) -> float64 + digits; flags chains that meet a NaN
//   k_i8_gemm      one hidden->hidden layer: workgroup = 64 features x 64 data rows (4 waves x 16 rows x 4 tiles), K in
//                  chunks of 64: weight digits of the chunk through double-buffered LDS (LDS-DMA), activation digits
//                  straight from HBM into the B operand, 26 exact products per tile and chunk into 7 int32 level
//                  accumulators that live across the chunks; then recombine, scale + bias, tanh, float64 + digits out.
// A float64 GEMM tile of this size costs 64 x 64 = 4096 cycles of the vector pipe per 64 of K; here it is 104 MFMAs x
// 16 cycles on the int8 pipe.  Flagged chains take a plain float64 loop per tile (IEEE semantics of the f64 kernels).
namespace {

struct I8First {
    int64_t p, offW, offB;
    int d, h, Nb, has_bias, rows_per_wg;
};
// first layer: a wave = 16 data rows per pass, lane (q, c) = features 16 t + 4 q + r of row c; d <= 16 inputs; the
// workgroup (8 waves) keeps W0, b0 and the tanh table in LDS and walks rows_per_wg rows
__global__ __launch_bounds__(512) void k_i8_first(I8First a, const double* __restrict__ W, const double* __restrict__ X,
                                                 const int32_t* __restrict__ row_idx, double* __restrict__ act_out,
                                                 unsigned char* __restrict__ ad_out, int* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) char smem1[];
    double* tanh_tab = reinterpret_cast<double*>(smem1);
    double* w0 = tanh_tab + ((TANH_TAB + 1) & ~1);                  // [h][d + 1]: weights then bias
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, c = lane & 15;
    const double* Wb = W + (int64_t)b * a.p;
    qn_tanh_table64_stage(tanh_tab, tid, 512);
    const int dd = a.d + 1;
    int bad = 0;
    for (int e = tid; e < a.h * dd; e += 512) {
        const int j = e / dd, k = e % dd;
        const double v = k < a.d ? Wb[a.offW + (int64_t)j * a.d + k] : (a.has_bias ? Wb[a.offB + j] : 0.0);
        bad |= !qn_bounded(v);
        w0[e] = v;
    }
    __syncthreads();
    const int64_t plane = (int64_t)a.Nb * a.h;
    const int nbeg = blockIdx.x * a.rows_per_wg;
    bool small = false;
    for (int n0 = nbeg; n0 < nbeg + a.rows_per_wg && n0 < a.Nb; n0 += 128) {
        const int n = n0 + 16 * wave + c;
        const bool live = n < a.Nb;
        const int nn = live ? n : a.Nb - 1;
        const int64_t rr = row_idx ? (int64_t)row_idx[(int64_t)b * a.Nb + nn] : (int64_t)nn;
        double x[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            x[k] = k < a.d ? X[rr * a.d + k] : 0.0;
            bad |= !qn_bounded(x[k]);
        }
        unsigned amx = 0;                                           // largest |a_1| (high word) of the wave's 16 rows
        for (int kc = 0; kc < a.h / 64; ++kc) {
            int D[NS][4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                double av[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 64 * kc + 16 * m + 4 * q + r;
                    double z = w0[j * dd + a.d];
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        if (k < a.d) z = fma(w0[j * dd + k], x[k], z);
                    // (unbounded weights / inputs are flagged: those chains are recomputed by the float64 path below)
                    av[r] = qn_tanh_f64_tab64(z, tanh_tab);
                    amx = max(amx, (unsigned)__double2hiint(av[r]) & 0x7fffffffu);
                    if (live) act_out[((int64_t)b * a.h + j) * a.Nb + n] = av[r];
                }
                int S[NS];
                slice4(av, S);
#pragma unroll
                for (int k = 0; k < NS; ++k) D[k][m] = S[k];
            }
            if (live && ad_out) {
                unsigned char* dst = ad_out + (int64_t)b * NS * plane + (int64_t)n * a.h + 64 * kc + 16 * q;
#pragma unroll
                for (int k = 0; k < NS; ++k) *reinterpret_cast<v4i*>(dst + k * plane) = (v4i){D[k][0], D[k][1], D[k][2], D[k][3]};
            }
        }
        // the digits carry an absolute error of 2^-47: rows whose activations are all tiny (< 2^-5, but not all zero) would
        // lose relative accuracy in the next layer -- bit 2: the chain's hidden layers run in plain float64
        amx = wave_max_u32(amx);
        small |= amx != 0 && amx < TINY_ACT_HI;
    }
    if (__any(bad) && lane == 0) atomicOr(&flags[b], 2);            // bit 1: the first layer must be redone in plain float64
    if (small && lane == 0) atomicOr(&flags[b], 4);
}
// flagged chains (bit 1 of k_i8_first: an unbounded first-layer weight or input): the first layer again with the
// NaN-propagating float64 tanh; hidden layers of flagged chains take the float64 path of k_i8_gemm
__global__ __launch_bounds__(256) void k_i8_first_slow(I8First a, const double* __restrict__ W, const double* __restrict__ X,
                                                      const int32_t* __restrict__ row_idx, double* __restrict__ act_out,
                                                      const int* __restrict__ flags) {
    const int b = blockIdx.y;
    if (!(flags[b] & 2)) return;
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= a.Nb) return;
    const int64_t rr = row_idx ? (int64_t)row_idx[(int64_t)b * a.Nb + n] : (int64_t)n;
    const double* Wb = W + (int64_t)b * a.p;
    for (int j = 0; j < a.h; ++j) {
        double z = a.has_bias ? Wb[a.offB + j] : 0.0;
        for (int k = 0; k < a.d; ++k) z = fma(Wb[a.offW + (int64_t)j * a.d + k], X[rr * a.d + k], z);
        act_out[((int64_t)b * a.h + j) * a.Nb + n] = qn_tanh_f64(z);
    }
}

struct I8Gemm {
    int64_t p, offW, offB, wd_chain, wd_off, sb_chain, sb_off;
    int h_in, h_out, Nb, has_bias;
    int mtiles, rsegs, rows_per_wg, outer_total;   // 1-D XCD-aware grid: the m-tiles of one (row segment, chain) on ONE XCD
};
__device__ __forceinline__ void glds16b(const unsigned char* src, unsigned char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// One hidden->hidden layer.  Workgroup = 8 waves = (chain, 64 output features, a segment of rows_per_wg data rows): the
// tile's weight digits for the WHOLE K ([KC][6][64 rows][64 B], 24 KB per 64 of K) are fetched once into LDS and stay
// there; the workgroup then walks its rows 128 at a time, each wave 16 rows x 64 features: activation digits straight
// from HBM into the B operand (next chunk prefetched), 4 x 26 exact products per chunk into 4 x 7 int32 level
// accumulators, then recombine, scale + bias, tanh, float64 + digits out.  No barrier inside the row loop.
template <int LMIN, int KC>
__global__ __launch_bounds__(512, 1) void k_i8_gemm(I8Gemm g, const double* __restrict__ W, const unsigned char* __restrict__ Wd,
                                                   const double* __restrict__ sb, const unsigned char* __restrict__ ad_in,
                                                   const double* __restrict__ act_in, int* flags,
                                                   double* __restrict__ act_out, unsigned char* __restrict__ ad_out) {
    constexpr int NLEV = 2 * (NS - 1) - LMIN + 1, NPROD = nprod(LMIN);
    constexpr int TILE_B = NS * 64 * 64;                   // one K-chunk of the tile's weight digits: [6][64 rows][64 B]
    extern __shared__ __attribute__((aligned(16))) char smem8[];
    unsigned char* wbuf = reinterpret_cast<unsigned char*>(smem8);                       // KC x 24 KB
    double* tanh_tab = reinterpret_cast<double*>(smem8 + KC * TILE_B);
    double* sbt = tanh_tab + ((TANH_TAB + 1) & ~1);                                      // [64][2]
    const int seq = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int mt = seq % g.mtiles, outer = (seq / g.mtiles) * 8 + xcd;
    if (outer >= g.outer_total) return;
    const int b = outer / g.rsegs, m0 = mt * 64, nbeg = (outer % g.rsegs) * g.rows_per_wg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, c = lane & 15;
    const int64_t in_plane = (int64_t)g.Nb * g.h_in, out_plane = (int64_t)g.Nb * g.h_out;

    if (flags[b]) {
        // exceptional chain (a weight or input that is not finite and < 2^500): the tile in plain float64
        const double* Wg = W + (int64_t)b * g.p + g.offW;
        for (int n0 = nbeg; n0 < nbeg + g.rows_per_wg && n0 < g.Nb; n0 += 128) {
            const int n = n0 + 16 * wave + c;
            if (n >= g.Nb) continue;
            for (int t = 0; t < 4; ++t)
                for (int r = 0; r < 4; ++r) {
                    const int j = m0 + 16 * t + 4 * q + r;
                    double z = g.has_bias ? W[(int64_t)b * g.p + g.offB + j] : 0.0;
                    for (int i = 0; i < g.h_in; ++i) z = fma(Wg[(int64_t)j * g.h_in + i], act_in[((int64_t)b * g.h_in + i) * g.Nb + n], z);
                    act_out[((int64_t)b * g.h_out + j) * g.Nb + n] = qn_tanh_f64(z);
                }
        }
        return;                                            // (the next layer of this chain takes this path too: no digits needed)
    }

    {   // the tile's digits: KC x 24 wave-instructions of 1 KB = 16 rows x 64 B each, LDS-DMA
        const unsigned char* wd = Wd + (int64_t)b * g.wd_chain + g.wd_off;
        const int64_t wplane = (int64_t)g.h_out * g.h_in;
        for (int i = wave; i < KC * 24; i += 8) {
            const int kc = i / 24, ii = i % 24, dig = ii >> 2, r16 = ii & 3;
            glds16b(wd + dig * wplane + (int64_t)(m0 + 16 * r16 + (lane >> 2)) * g.h_in + 64 * kc + 16 * (lane & 3), wbuf + i * 1024);
        }
    }
    qn_tanh_table64_stage(tanh_tab, tid, 512);
    if (tid < 128) sbt[tid] = sb[(int64_t)b * g.sb_chain + g.sb_off + 2 * m0 + tid];
    __syncthreads();                                       // (waits for the DMA too: vmcnt(0) ahead of the barrier)

    const int lofs = c * 64 + 16 * (q ^ slot_swz(c));
    const unsigned char* adb = ad_in + (int64_t)b * NS * in_plane + 16 * q;
    auto loadB = [&](int nrow, int kc, v4i (&B)[NS]) {
#pragma unroll
        for (int k = 0; k < NS; ++k) B[k] = *reinterpret_cast<const v4i*>(adb + k * in_plane + (int64_t)nrow * g.h_in + 64 * kc);
    };
    auto row_of = [&](int n0) { const int n = n0 + 16 * wave + c; return n < g.Nb ? n : g.Nb - 1; };
    v4i Bf[NS], Bnx[NS];
    loadB(row_of(nbeg), 0, Bf);
    for (int n0 = nbeg; n0 < nbeg + g.rows_per_wg && n0 < g.Nb; n0 += 128) {
        const int n = n0 + 16 * wave + c;
        const bool live = n < g.Nb;
        const int nn = live ? n : g.Nb - 1;
        v4i acc[4][NLEV];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int l = 0; l < NLEV; ++l) acc[t][l] = (v4i){0, 0, 0, 0};
#pragma unroll 1
        for (int kc = 0; kc < KC; ++kc) {                   // (not unrolled: the fragment loads of all chunks would be hoisted)
            if (kc + 1 < KC) loadB(nn, kc + 1, Bnx);
            else loadB(row_of(n0 + 128), 0, Bnx);           // first chunk of the next pass (clamped: harmless if there is none)
            const unsigned char* tile = wbuf + kc * TILE_B + lofs;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                v4i Af[NS];
#pragma unroll
                for (int wi = 0; wi < NS; ++wi) Af[wi] = *reinterpret_cast<const v4i*>(tile + wi * 4096 + t * 1024);
#pragma unroll
                for (int k = 0; k < NPROD; ++k) issue_product<LMIN, NLEV, false>(k, acc[t], Af, Bf);
            }
#pragma unroll
            for (int k = 0; k < NS; ++k) Bf[k] = Bnx[k];
        }
        // ---- epilogue: recombine (every level on its own: with K > 64 the pair sums would not fit int32), scale +
        // bias, tanh, float64 out, digits out
        int D[NS][4];
        unsigned amx = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            double av[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double ts = (double)acc[t][NLEV - 1][r];
#pragma unroll
                for (int l = NLEV - 2; l >= 0; --l) ts = fma(ts, 256.0, (double)acc[t][l][r]);
                const int jl = 16 * t + 4 * q + r;
                const double2 sc = *reinterpret_cast<const double2*>(sbt + 2 * jl);
                av[r] = qn_tanh_f64_tab64(fma(ts, sc.x, sc.y), tanh_tab);
                amx = max(amx, (unsigned)__double2hiint(av[r]) & 0x7fffffffu);
                if (live) act_out[((int64_t)b * g.h_out + m0 + jl) * g.Nb + n] = av[r];
            }
            if (ad_out) {
                int S[NS];
                slice4(av, S);
#pragma unroll
                for (int k = 0; k < NS; ++k) D[k][t] = S[k];
            }
        }
        if (ad_out && live) {
            unsigned char* dst = ad_out + (int64_t)b * NS * out_plane + (int64_t)n * g.h_out + m0 + 16 * q;
#pragma unroll
            for (int k = 0; k < NS; ++k) *reinterpret_cast<v4i*>(dst + k * out_plane) = (v4i){D[k][0], D[k][1], D[k][2], D[k][3]};
        }
        if (ad_out) {                                      // (as k_i8_first: tiny activations -> the chain's next layers in float64)
            amx = wave_max_u32(amx);
            if (amx != 0 && amx < TINY_ACT_HI && lane == 0) atomicOr(&flags[b], 4);
        }
    }
}

}  // namespace

// ---- host side of the layer-wise int8-slice forward (called by qn_generic.hip for float64 tanh networks)
namespace {
bool i8net_of(const qn_desc* d, I8Net* net) {
    const int L = d->nlayers;
    if (d->kind != QN_KIND_MLP || d->act != QN_ACT_TANH || L < 3 || d->dims[0] > 16) return false;
    for (int l = 1; l + 1 < L; ++l)
        if (d->dims[l] != 128 && d->dims[l] != 256) return false;         // K resident in LDS: 2 or 4 chunks of 64
    net->nl = L - 2;
    net->p = d->p; net->has_bias = d->has_bias;
    int64_t db = 0, sd = 0;
    for (int l = 1; l + 1 < L; ++l) {
        if (d->dims[l] % 64 || d->dims[l + 1] % 64) return false;
        const int li = l - 1;
        net->h[li] = d->dims[l]; net->h[li + 1] = d->dims[l + 1];
        net->offW[li] = d->offW[l]; net->offB[li] = d->offB[l];
        net->offD[li] = db; net->offS[li] = sd;
        db += (int64_t)NS * d->dims[l] * d->dims[l + 1];
        sd += 2 * (int64_t)d->dims[l + 1];
    }
    net->dbytes = (db + 255) / 256 * 256;
    net->sdoubles = sd;
    return true;
}
constexpr size_t i8gemm_lds(int kc) { return (size_t)kc * NS * 64 * 64 + sizeof(double) * (((TANH_TAB + 1) & ~1) + 128); }
// raise the dynamic-LDS limit of a kernel once per process (not per launch: the launch path stays free of non-stream
// API calls, so it can be captured into a HIP graph)
int i8_arm(const void* fn, size_t bytes) {
    static std::mutex mu;
    static std::unordered_set<uint64_t> armed;              // (kernel, device): the attribute is per device
    if (bytes <= 64 * 1024) return QN_OK;
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

bool qn_i8_layers_apply(const qn_desc* d) {
    I8Net net;
    return i8net_of(d, &net);
}
// bytes of: weight digit planes | {scale, bias} pairs | chain flags | two activation digit buffers
size_t qn_i8_layers_workspace(const qn_desc* d, int B, int Nb) {
    I8Net net;
    if (!i8net_of(d, &net)) return 0;
    int hmax = 0;
    for (int li = 0; li <= net.nl; ++li) hmax = net.h[li] > hmax ? net.h[li] : hmax;
    return qn_align((size_t)B * net.dbytes) + qn_align((size_t)B * net.sdoubles * sizeof(double)) + qn_align((size_t)B * sizeof(int)) +
           2 * qn_align((size_t)B * NS * Nb * hmax);
}
// forward through the first layer and every hidden->hidden layer: act[l] (float64 [B][dims[l+1]][Nb], l = 0 .. L-2) are
// written as the float64 layer-wise kernels would write them
int qn_i8_layers_forward(const qn_desc* d, const double* W, const double* X, const int32_t* row_idx, int B, int Nb,
                         double* const* act, void* ws, hipStream_t st) {
    I8Net net;
    if (!i8net_of(d, &net)) return QN_EUNSUPPORTED;
    int hmax = 0;
    for (int li = 0; li <= net.nl; ++li) hmax = net.h[li] > hmax ? net.h[li] : hmax;
    char* base = static_cast<char*>(ws);
    unsigned char* Wd = reinterpret_cast<unsigned char*>(base);
    base += qn_align((size_t)B * net.dbytes);
    double* sb = reinterpret_cast<double*>(base);
    base += qn_align((size_t)B * net.sdoubles * sizeof(double));
    int* flags = reinterpret_cast<int*>(base);
    base += qn_align((size_t)B * sizeof(int));
    unsigned char* ad[2];
    ad[0] = reinterpret_cast<unsigned char*>(base);
    ad[1] = ad[0] + qn_align((size_t)B * NS * Nb * hmax);
#ifndef QN_I8_LMIN
#define QN_I8_LMIN 4
#endif
    QN_HIP_CHECK(hipMemsetAsync(flags, 0, (size_t)B * sizeof(int), st));
    hipLaunchKernelGGL((k_i8_slice_w<QN_I8_LMIN>), dim3(B, net.nl, 8), dim3(256), 0, st, net, W, Wd, sb, flags);
    I8First f;
    f.p = d->p; f.offW = d->offW[0]; f.offB = d->offB[0]; f.d = d->dims[0]; f.h = d->dims[1]; f.Nb = Nb; f.has_bias = d->has_bias;
    // rows per workgroup: enough workgroups for the chip (>= 2 per CU), at least one 128-row pass each
    int rpw = 128 * (int)(((int64_t)Nb * B + 512LL * 128 - 1) / (512LL * 128));
    if (rpw > 1024) rpw = 1024;
    f.rows_per_wg = rpw;
    const int rsegs = (Nb + rpw - 1) / rpw;
    const size_t lds1 = sizeof(double) * (((TANH_TAB + 1) & ~1) + (size_t)f.h * (f.d + 1));
    hipLaunchKernelGGL(k_i8_first, dim3(rsegs, B), dim3(512), lds1, st, f, W, X, row_idx, act[0], ad[0], flags);
    hipLaunchKernelGGL(k_i8_first_slow, dim3((Nb + 255) / 256, B), dim3(256), 0, st, f, W, X, row_idx, act[0], (const int*)flags);
    for (int li = 0; li < net.nl; ++li) {
        I8Gemm g;
        g.p = d->p; g.offW = net.offW[li]; g.offB = net.offB[li]; g.wd_chain = net.dbytes; g.wd_off = net.offD[li];
        g.sb_chain = net.sdoubles; g.sb_off = net.offS[li]; g.h_in = net.h[li]; g.h_out = net.h[li + 1]; g.Nb = Nb;
        g.has_bias = d->has_bias;
        unsigned char* out_digits = li + 1 < net.nl ? ad[(li + 1) & 1] : nullptr;      // the last hidden layer feeds float64 consumers
        g.mtiles = g.h_out / 64; g.rsegs = rsegs; g.rows_per_wg = rpw; g.outer_total = rsegs * B;
        const unsigned grid = (unsigned)(((g.outer_total + 7) / 8) * 8 * g.mtiles);
        if (g.h_in == 128) {
            auto kern = k_i8_gemm<QN_I8_LMIN, 2>;
            if (int rc = i8_arm(reinterpret_cast<const void*>(kern), i8gemm_lds(2))) return rc;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), i8gemm_lds(2), st, g, W, Wd, sb, (const unsigned char*)ad[li & 1],
                               (const double*)act[li], flags, act[li + 1], out_digits);
        } else {
            auto kern = k_i8_gemm<QN_I8_LMIN, 4>;
            if (int rc = i8_arm(reinterpret_cast<const void*>(kern), i8gemm_lds(4))) return rc;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), i8gemm_lds(4), st, g, W, Wd, sb, (const unsigned char*)ad[li & 1],
                               (const double*)act[li], flags, act[li + 1], out_digits);
        }
    }
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}
