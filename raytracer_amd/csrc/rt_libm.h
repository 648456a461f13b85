// asinf / atanf / atan2f as the reference's binaries evaluate them: the float routines of glibc 2.35
// (the fdlibm-derived ones: sysdeps/ieee754/flt-32/e_asinf.c, s_atanf.c, e_atan2f.c), restated as
// binary32 operation sequences with IEEE division and square root and no FMA contraction.
//
// Why: Sphere::intersect takes its texture coordinates from atan2f and asinf (src/objects.cpp:47);
// the device library's own versions are within 2 ulp of glibc's, which moves a texel or a photon
// cell now and then.  With these the sphere uv is the same float the reference computes.
//
// Pinned in the build container against the host libm (the same glibc): asinf on EVERY float in
// [-1, 1], atanf on EVERY float, atan2f on 1.6e9 operand pairs (uniform bit patterns and uniform
// values in the unit square) — zero mismatches (tools/libm_model_check.cpp; a reduced sweep runs in
// tests/test_libm_model.py).  The header compiles for the host too, which is how that check runs.
#pragma once

#include <math.h>
#include <stdint.h>

#ifndef RT_LIBM_FN
#ifdef __HIPCC__
#define RT_LIBM_FN __device__ __forceinline__
#else
#define RT_LIBM_FN static inline
#endif
#endif

namespace rtd {

RT_LIBM_FN uint32_t libm_bits(float f) { return __builtin_bit_cast(uint32_t, f); }
RT_LIBM_FN float libm_float(uint32_t u) { return __builtin_bit_cast(float, u); }

// The shared minimax polynomial of e_asinf.c
RT_LIBM_FN float glibc_asinf_poly(float t) {
    return t * (1.666675248e-1f + t * (7.495297643e-2f + t * (4.547037598e-2f + t * (2.417951451e-2f + t * 4.216630880e-2f))));
}

RT_LIBM_FN float glibc_asinf(float x) {
    const float pio2_hi = 1.57079637050628662109375f, pio2_lo = -4.37113900018624283e-8f, pio4_hi = 0.785398185253143310546875f;
    const uint32_t hx = libm_bits(x), ix = hx & 0x7fffffffu;
    if (ix == 0x3f800000u) return x * pio2_hi + x * pio2_lo;   // +-1
    if (ix > 0x3f800000u) return (x - x) / (x - x);            // |x| > 1 or NaN
    if (ix < 0x3f000000u) {                                    // |x| < 0.5
        if (ix < 0x32000000u) return x;                        // |x| < 2^-27
        const float t = x * x;
        return x + x * glibc_asinf_poly(t);
    }
    const float w = 1.0f - fabsf(x);
    float t = w * 0.5f;
    const float p = glibc_asinf_poly(t);
    const float s = sqrtf(t);
    if (ix >= 0x3f79999au) {                                   // |x| >= 0.975
        t = pio2_hi - (2.0f * (s + s * p) - pio2_lo);
    } else {
        const float sh = libm_float(libm_bits(s) & 0xfffff000u);
        const float c = (t - sh * sh) / (s + sh);
        const float pp = 2.0f * s * p - (pio2_lo - 2.0f * c);
        const float q = pio4_hi - 2.0f * sh;
        t = pio4_hi - (pp - q);
    }
    return (hx >> 31) ? -t : t;
}

RT_LIBM_FN float glibc_atanf(float x) {
    const uint32_t hx = libm_bits(x), ix = hx & 0x7fffffffu;
    const bool negative = (hx >> 31) != 0;
    if (ix >= 0x4c000000u) {                                   // |x| >= 2^25
        if (ix > 0x7f800000u) return x + x;
        const float r = 1.5707962513e+00f + 7.5497894159e-08f;
        return negative ? -r : r;
    }
    float hi = 0.0f, lo = 0.0f;
    bool reduced = true;
    if (ix < 0x3ee00000u) {                                    // |x| < 0.4375
        if (ix < 0x31000000u) return x;                        // |x| < 2^-29
        reduced = false;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000u) {                                // |x| < 1.1875
            if (ix < 0x3f300000u) {                            // 7/16 <= |x| < 11/16
                hi = 4.6364760399e-01f, lo = 5.0121582440e-09f;
                x = (2.0f * x - 1.0f) / (2.0f + x);
            } else {                                           // 11/16 <= |x| < 19/16
                hi = 7.8539812565e-01f, lo = 3.7748947079e-08f;
                x = (x - 1.0f) / (x + 1.0f);
            }
        } else if (ix < 0x401c0000u) {                         // |x| < 2.4375
            hi = 9.8279368877e-01f, lo = 3.4473217170e-08f;
            x = (x - 1.5f) / (1.0f + 1.5f * x);
        } else {                                               // 2.4375 <= |x| < 2^25
            hi = 1.5707962513e+00f, lo = 7.5497894159e-08f;
            x = -1.0f / x;
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (3.3333334327e-01f + w * (1.4285714924e-01f + w * (9.0908870101e-02f + w * (6.6610731184e-02f + w * (4.9768779427e-02f + w * 1.6285819933e-02f)))));
    const float s2 = w * (-2.0000000298e-01f + w * (-1.1111110449e-01f + w * (-7.6918758452e-02f + w * (-5.8335702866e-02f + w * -3.6531571299e-02f))));
    if (!reduced) return x - x * (s1 + s2);
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return negative ? -r : r;
}

RT_LIBM_FN float glibc_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const uint32_t hx = libm_bits(x), ix = hx & 0x7fffffffu, hy = libm_bits(y), iy = hy & 0x7fffffffu;
    if (ix > 0x7f800000u || iy > 0x7f800000u) return x + y;    // NaN
    if (hx == 0x3f800000u) return glibc_atanf(y);              // x == 1
    const bool y_neg = (hy >> 31) != 0, x_neg = (hx >> 31) != 0;
    if (iy == 0) return x_neg ? (y_neg ? -pi - tiny : pi + tiny) : y;
    if (ix == 0) return y_neg ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000u) {
        if (iy == 0x7f800000u) {
            const float r = x_neg ? 3.0f * pi_o_4 + tiny : pi_o_4 + tiny;
            return y_neg ? -r : r;
        }
        const float r = x_neg ? pi + tiny : 0.0f;
        return y_neg ? -r : r;
    }
    if (iy == 0x7f800000u) return y_neg ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = (static_cast<int32_t>(iy) - static_cast<int32_t>(ix)) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;                     // |y/x| > 2^60
    else if (x_neg && k < -60) z = 0.0f;                       // |y/x| < 2^-60, x < 0
    else z = glibc_atanf(fabsf(y / x));
    if (!x_neg) return y_neg ? -z : z;
    return y_neg ? (z - pi_lo) - pi : pi - (z - pi_lo);
}

}  // namespace rtd
