// Device math shared by every kernel family (so the generic and the fused paths agree).
#pragma once
#include <hip/hip_runtime.h>

// ---------------------------------------------------------------------------------------------
// tanh for float64.
//
// On gfx950 the float64 MFMA and the float64 VALU share one pipe (tools/ubench: a wave's
// v_fma_f64 stream makes no progress while another wave of the SIMD runs v_mfma_f64_16x16x4_f64),
// so every VALU cycle of the activation is paid in full next to the GEMM.  The device library's
// tanh costs ~125 instructions; this one ~31 (~150 cycles per wave-instruction-element):
//
//   tanh(x) = sign(x) * -E / (2 + E),  E = expm1(-2|x|)
//   -2|x| = n ln2 + r, |r| <= ln2/2: n by the 1.5*2^52 trick (one fma; its low mantissa word IS n,
//           so 2^n is built with one integer op -- no v_rndne / v_cvt / v_ldexp);
//   expm1(r) = r + r^2 P(r), P = degree-10 near-minimax fit (8.5e-19 relative);
//   E = 2^n expm1(r) + (2^n - 1)   (2^n - 1 exact; E in [-1, 0], no cancellation);
//   quotient: v_rcp_f64 (2^-24) + one cubic Newton step y(1 + e + e^2) (2^-73), then one product.
// |x| is clamped to 20 (tanh = 1 to the last bit from 19.07 on), NaN is propagated by an integer
// mask on the high word (no v_cmp / v_cndmask: 20+ cycles a pair on this chip).
// Measured against an 80-bit reference on [-20, 20]: max error < 3 ulp, mean 0.3 ulp
// (tests/test_gpu_math.py).
//
// NANSAFE = false drops the NaN mask (the clamp's v_min_f64 returns 20 for a NaN input, so a NaN would
// come out as +-1) and copies the sign with one v_bfi_b32: 4 instructions and the compare/select
// pair fewer.  Same values for every non-NaN input, +-inf included.  The fused kernels use it only
// after proving that no NaN can reach an activation (finite, bounded weights and inputs: qn_bounded).
template <bool NANSAFE>
__device__ __forceinline__ double qn_tanh_f64_impl(double x) {
    const double kClamp = 20.0;
    double ax;
    asm("v_min_f64 %0, |%1|, %2" : "=v"(ax) : "v"(x), "s"(kClamp));
    const double kMagic = 6755399441055744.0;                         // 1.5 * 2^52
    const double zm = fma(ax, -2.8853900817779268, kMagic);           // -2|x| log2(e), rounded to integer
    const double n = zm - kMagic;
    const double t = -2.0 * ax;
    double r = fma(n, -6.93147180369123816490e-01, t);
    r = fma(n, -1.90821492927058770002e-10, r);
    double p = 2.09146793765839349e-09;
    p = fma(p, r, 2.51052063739570109e-08);
    p = fma(p, r, 2.75572736613486373e-07);
    p = fma(p, r, 2.75572554257464351e-06);
    p = fma(p, r, 2.48015873255333634e-05);
    p = fma(p, r, 1.98412698748004929e-04);
    p = fma(p, r, 1.38888888888837525e-03);
    p = fma(p, r, 8.33333333332614105e-03);
    p = fma(p, r, 4.16666666666666713e-02);
    p = fma(p, r, 1.66666666666666713e-01);
    p = fma(p, r, 0.5);
    const double em1r = fma(r * r, p, r);
    const double s = __hiloint2double((__double2loint(zm) + 1023) << 20, 0);   // 2^n, n in [-58, 0]
    const double E = fma(s, em1r, s - 1.0);
    const double den = 2.0 + E;                                        // in [1, 2]
    double y = __builtin_amdgcn_rcp(den);                              // 2^-24
    const double e0 = fma(-den, y, 1.0);
    y = fma(y, fma(e0, e0, e0), y);                                    // cubic step: error e0^3 = 2^-73
    double q = -E * y;                                                 // >= 0
    if constexpr (!NANSAFE) return __builtin_copysign(q, x);
    const int xh = __double2hiint(x);
    const int nanmask = (0x7ff00000 - (xh & 0x7fffffff)) >> 31;        // all ones iff x is NaN
    const int qh = (__double2hiint(q) | (xh & 0x80000000)) | nanmask;
    return __hiloint2double(qh, __double2loint(q));
}
__device__ __forceinline__ double qn_tanh_f64(double x) { return qn_tanh_f64_impl<true>(x); }
__device__ __forceinline__ double qn_tanh_f64_finite(double x) { return qn_tanh_f64_impl<false>(x); }

// |v| < 2^500 (and not NaN): with every weight and input bounded like this, no product or 64-term sum
// inside the network can overflow, so no inf - inf and therefore no NaN can appear downstream.
__device__ __forceinline__ bool qn_bounded(double v) {
    return (unsigned)(__double2hiint(v) & 0x7fffffff) < 0x5f300000u;   // exponent field < 1023 + 500
}

// ---------------------------------------------------------------------------------------------
// Table-assisted tanh for float64 (kernels that have LDS to spare): 18 DP instructions + v_rcp instead of 25.
//   |x| = a + b,  a = n/16 (n = round(16|x|) by the 1.5*2^52 trick; its low mantissa word IS n),  |b| <= 1/32 exact;
//   tanh(a) from a table of 321 correctly rounded values in LDS (qn_tanh_table.h; one ds_read_b64 -- the LDS
//   pipe is idle next to the DP pipe in these kernels);
//   tanh(b) = b + b^3 (-1/3 + 2/15 b^2 - 17/315 b^4 + 62/2835 b^6)   (next term 8e-18 relative at |b| = 1/32);
//   tanh(a + b) = (tanh a + tanh b) / (1 + tanh a tanh b): no cancellation (tanh a >= 0.0624 > |tanh b| for n >= 1,
//   and for n = 0 the result is tanh b itself), quotient by v_rcp_f64 + a residual-corrected division.
//   (A/B-tested on one box, tools/ab_run.sh: the variant T + (1 - T^2) tb / (1 + T tb) with a second table, one DP
//   instruction fewer, runs at the same speed -- the kernel is no longer limited by the last VALU instruction.)
// Same contract as qn_tanh_f64_impl: NANSAFE = false for arguments that cannot be NaN (+-inf included).
#include "qn_tanh_table.h"
static __device__ const double qn_tanh_table_g[QN_TANH_TAB_N] = {QN_TANH_TAB_VALUES};
#define QN_TANH_LDS_DOUBLES (QN_TANH_TAB_N + 1)
// copy the table into LDS (call with all threads of the block, then synchronise)
__device__ __forceinline__ void qn_tanh_table_stage(double* lds_tab, int tid, int nthreads) {
    for (int e = tid; e < QN_TANH_TAB_N; e += nthreads) lds_tab[e] = qn_tanh_table_g[e];
}
template <bool NANSAFE>
__device__ __forceinline__ double qn_tanh_f64_tab(double x, const double* __restrict__ lds_tab) {
    const double kClamp = 20.0;
    double ax;
    asm("v_min_f64 %0, |%1|, %2" : "=v"(ax) : "v"(x), "s"(kClamp));
    const double kMagic = 6755399441055744.0;                          // 1.5 * 2^52
    const double zm = fma(ax, 16.0, kMagic);                           // 16|x| rounded to an integer n in 0..320
    const double T = lds_tab[__double2loint(zm)];                      // tanh(n / 16)
    const double b = fma(zm - kMagic, -0.0625, ax);                    // exact
    const double b2 = b * b;
    double q = 2.18694885361552028e-02;                                // 62/2835
    q = fma(q, b2, -5.39682539682539683e-02);                          // -17/315
    q = fma(q, b2, 1.33333333333333333e-01);                           // 2/15
    q = fma(q, b2, -3.33333333333333333e-01);                          // -1/3
    const double tb = fma(b * b2, q, b);
    const double num = T + tb;
    const double den = fma(T, tb, 1.0);                                // in [0.97, 1.03]
    // quotient with ONE final rounding: y1 = 1/den to 2^-48, r0 = num y1, exact residual num - den r0 folded back
    // (a plain num * (1/den) rounds the reciprocal and the product: a full ulp more, which matters here because num
    // and den can sit in the binade above the result)
    const double y0 = __builtin_amdgcn_rcp(den);                       // 2^-24
    const double y1 = fma(y0, fma(-den, y0, 1.0), y0);
    const double r0 = num * y1;
    const double r = fma(fma(-den, r0, num), y1, r0);                  // >= 0
    if constexpr (!NANSAFE) return __builtin_copysign(r, x);
    const int xh = __double2hiint(x);
    const int nanmask = (0x7ff00000 - (xh & 0x7fffffff)) >> 31;        // all ones iff x is NaN
    const int rh = (__double2hiint(r) | (xh & 0x80000000)) | nanmask;
    return __hiloint2double(rh, __double2loint(r));
}

// ---------------------------------------------------------------------------------------------
// tanh to ABSOLUTE accuracy ~2^-51 for arguments that cannot be NaN: the activation of the int8-slice forward kernel
// (qn_fused_i8.hip), whose next step rounds the result to a multiple of 2^-46 anyway.  14 DP instructions + v_rcp_f64
// (qn_tanh_f64_tab: 18 + v_rcp): a table of tanh(n / 64) (1281 values, 10 KB of LDS) leaves |b| <= 1/128, so
//   tanh(b) = b + b^3 (-1/3 + 2/15 b^2)                 (next term 17/315 b^7 < 2^-53),
//   tanh(a + b) = (T + t) / (1 + T t),  |T t| <= 2^-7:  v_rcp_f64 (2^-24) + one cubic Newton step (2^-72), one product.
// Relative error of the quotient ~1.5 ulp; no residual correction (it bought the last half ulp of qn_tanh_f64_tab).
#include "qn_tanh_table64.h"
static __device__ const double qn_tanh_table64_g[QN_TANH_TAB64_N] = {QN_TANH_TAB64_VALUES};
#define QN_TANH64_LDS_DOUBLES (QN_TANH_TAB64_N + 1)
__device__ __forceinline__ void qn_tanh_table64_stage(double* lds_tab, int tid, int nthreads) {
    for (int e = tid; e < QN_TANH_TAB64_N; e += nthreads) lds_tab[e] = qn_tanh_table64_g[e];
}
__device__ __forceinline__ double qn_tanh_f64_tab64(double x, const double* __restrict__ lds_tab) {
    const double kClamp = 20.0;
    double ax;
    asm("v_min_f64 %0, |%1|, %2" : "=v"(ax) : "v"(x), "s"(kClamp));
    const double kMagic = 6755399441055744.0;                          // 1.5 * 2^52
    const double zm = fma(ax, 64.0, kMagic);                           // 64|x| rounded to an integer n in 0..1280
    const double T = lds_tab[__double2loint(zm)];                      // tanh(n / 64)
    const double b = fma(zm - kMagic, -0.015625, ax);                  // exact, |b| <= 1/128
    const double b2 = b * b;
    const double tb = fma(b * b2, fma(b2, 1.33333333333333333e-01, -3.33333333333333333e-01), b);
    // tanh(n/64 + b) = T + (1 - T^2) u,  u = tb / (1 + e),  e = T tb, |e| < 2^-7:  1 / (1 + e) = (1 - e)(1 + e^2 + e^4) + O(e^6).
    // The dropped term is (1 - T^2) T^6 tb^7 <= 0.105 x 2^-49 = 2^-52.2 (largest at T^2 = 3/4): within one unit in the last
    // place where tanh > 0.5 -- and no v_rcp_f64 (16.7 issue cycles against 5.5 for a float64 fma, and the head of a chain of
    // four dependent instructions): +1.6 % on the headline kernel, A/B in one call.  (The quotient form (T + tb) / (1 + T tb)
    // with a Newton step on v_rcp_f64 was here through round 2.)
    const double e = T * tb;
    const double e2 = e * e;
    const double w = fma(-tb, e, tb);                                  // tb (1 - e)
    const double u = fma(w, fma(e2, e2, e2), w);
    return __builtin_copysign(fma(fma(-T, T, 1.0), u, T), x);
}

// relu with the reference's NaN semantics (torch.nn.ReLU: relu(NaN) = NaN, relu(x <= 0) = +0); `z > 0 ? z : 0` swallowed a NaN
template <typename T> __device__ __forceinline__ T qn_relu(T z) { return z > T(0) ? z : (z != z ? z : T(0)); }
// dz . act'(a) through the stored OUTPUT a = act(z), with the reference's semantics: tanh multiplies by 1 - a^2; relu is a
// SELECT (torch threshold_backward on the output: a <= 0 ? 0 : dz -- an Inf / NaN gradient does not pass a dead unit, a NaN
// output lets it pass), not a product with 0 / 1 (0 . Inf = NaN)
template <typename T> __device__ __forceinline__ T qn_act_bwd(T dz, T a, int act) {
    if (act == 1 /* QN_ACT_TANH */) return dz * (T(1) - a * a);
    if (act == 2 /* QN_ACT_RELU */) return a <= T(0) ? T(0) : dz;
    return dz;
}

// tanh for float32: 1 - 2 / (exp(2|x|) + 1) on the hardware exp2 / rcp (absolute error ~2e-7; NaN propagates through
// v_exp_f32), and below 0.3 -- where that form cancels and would only be ABSOLUTELY accurate: a network with small weights
// and no bias has activations of 1e-4 .. 1e-10, and tests/fuzz_all.py caught predictions that were pure rounding noise --
// the odd Taylor polynomial through x^9 (relative error < 6e-8 at 0.3).
__device__ __forceinline__ float qn_tanh_f32(float x) {
    const float ax = fminf(fabsf(x), 10.0f);
    const float e = __builtin_amdgcn_exp2f(ax * 2.8853900817779268f);
    const float t = fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
    const float x2 = x * x;
    const float p = x * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 2.1869488536155203e-02f, -5.3968253968253971e-02f), 1.3333333333333333e-01f),
                                      -3.3333333333333331e-01f), 1.0f);
    return x != x ? x : (ax < 0.3f ? p : __builtin_copysignf(t, x));
}
