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

// tanh for float32: 1 - 2 / (exp(2|x|) + 1) on the hardware exp2 / rcp (7 instructions, absolute error
// ~2e-7, i.e. float32-level; NaN propagates through v_exp_f32).
__device__ __forceinline__ float qn_tanh_f32(float x) {
    const float ax = fminf(fabsf(x), 10.0f);
    const float e = __builtin_amdgcn_exp2f(ax * 2.8853900817779268f);
    const float t = fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
    return x != x ? x : __builtin_copysignf(t, x);
}
