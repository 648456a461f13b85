// Device-side float32 arithmetic for the trace kernels (gfx950).
//
// Contract: every function evaluates the operations of the reference expression it names in the
// reference's ORDER, in binary32, with IEEE division and square root and no FMA contraction (the
// library is built with -ffp-contract=off; hipcc's default correctly-rounded f32 div/sqrt is kept).
// Sub-expressions the reference evaluates in double are evaluated in double
// here too.  Citations are to the reference sources under /root/reference.
#pragma once

#include <hip/hip_runtime.h>

#define RT_DEV __device__ __forceinline__

namespace rtd {

constexpr float kPi = 3.14159265358979323846f;   // src/types.h:19
constexpr float kInfinite = 3.402823466e+38f;    // src/types.h:20 (FLT_MAX)
constexpr float kEpsilon = 0.0001f;              // src/types.h:21

struct V3 {
    float x, y, z;
};
struct V2 {
    float x, y;
};
struct Col {
    float r, g, b, a;
};

RT_DEV V3 v3(float x, float y, float z) { return {x, y, z}; }
RT_DEV V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_DEV V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_DEV V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
RT_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// std::hypot(x,y,z) of libstdc++ (float): a*sqrt((x/a)^2+(y/a)^2+(z/a)^2), a = max(|x|,|y|,|z|)
// (src/types.h:73)
RT_DEV float hypot3(float x, float y, float z) {
    x = fabsf(x);
    y = fabsf(y);
    z = fabsf(z);
    const float a = x < y ? (y < z ? z : y) : (x < z ? z : x);
    if (a == 0.0f) return 0.0f;
    const float qx = x / a, qy = y / a, qz = z / a;
    return a * sqrtf(qx * qx + qy * qy + qz * qz);
}
RT_DEV float length(V3 v) { return hypot3(v.x, v.y, v.z); }
// src/types.h:74-80: three divisions, zero vector stays zero
RT_DEV V3 normalized(V3 v) {
    const float len = length(v);
    if (len == 0.0f) return {0.0f, 0.0f, 0.0f};
    return {v.x / len, v.y / len, v.z / len};
}

// Pointers into the scene tables that are read at wave-uniform addresses are converted to the
// constant address space: the data is immutable during a launch, so the loads become scalar loads
// into SGPRs (the tables themselves live in ordinary HBM allocations).
#define RT_CONSTANT __attribute__((address_space(4)))
template <typename T>
RT_DEV const RT_CONSTANT T* to_constant(const T* p) {
    return (const RT_CONSTANT T*)p;   // NOLINT: address-space cast needs the C form
}

// src/types.h:188-202, m = row-major 3x4 (any address space)
template <typename M>
RT_DEV V3 xform_point(M m, V3 p) {
    return {p.x * m[0] + p.y * m[1] + p.z * m[2] + m[3], p.x * m[4] + p.y * m[5] + p.z * m[6] + m[7],
            p.x * m[8] + p.y * m[9] + p.z * m[10] + m[11]};
}
template <typename M>
RT_DEV V3 xform_dir(M m, V3 p) {
    return {p.x * m[0] + p.y * m[1] + p.z * m[2], p.x * m[4] + p.y * m[5] + p.z * m[6],
            p.x * m[8] + p.y * m[9] + p.z * m[10]};
}

RT_DEV Col operator+(Col a, Col b) { return {a.r + b.r, a.g + b.g, a.b + b.b, a.a + b.a}; }
RT_DEV Col operator*(Col a, float s) { return {a.r * s, a.g * s, a.b * s, a.a * s}; }
RT_DEV Col operator*(Col a, Col b) { return {a.r * b.r, a.g * b.g, a.b * b.b, a.a * b.a}; }
RT_DEV Col operator/(Col a, float s) { return {a.r / s, a.g / s, a.b / s, a.a / s}; }

RT_DEV float clampf(float v, float lo, float hi) { return v < lo ? lo : (hi < v ? hi : v); }   // std::clamp

// static_cast<u32>(float) as the reference's x86-64 binaries do it (64-bit truncating convert, low
// 32 bits kept; NaN/out of range -> 0).  Formally UB there for negative inputs.
RT_DEV unsigned cast_u32(float f) {
    if (!(f > -9.2e18f && f < 9.2e18f)) return 0u;
    return static_cast<unsigned>(static_cast<long long>(f));
}
RT_DEV unsigned clamp_u32(unsigned v, unsigned lo, unsigned hi) { return v < lo ? lo : (hi < v ? hi : v); }

// modff: fractional part with the sign of x (truncation toward zero)
RT_DEV float frac_part(float x) {
    if (isinf(x)) return copysignf(0.0f, x);
    return x - truncf(x);
}

// (float)(−b ∓ sqrt(h)) with the sqrt and the sum in double (src/objects.cpp:33,39,187; Y8)
RT_DEV float root_minus(float b, float h) { return static_cast<float>(static_cast<double>(-b) - sqrt(static_cast<double>(h))); }
RT_DEV float root_plus(float b, float h) { return static_cast<float>(static_cast<double>(-b) + sqrt(static_cast<double>(h))); }

// ---- complex<float> as the O-strict reference build evaluates it --------------------------------
// (libstdc++ <complex> on top of libgcc/glibc; models pinned against the real libraries on 5e6
// random operands in the build container)
struct Cx {
    float re, im;
};
RT_DEV Cx cx_mul(Cx a, Cx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }   // __mulsc3, finite case
RT_DEV Cx cx_scale(Cx a, float s) { return {a.re * s, a.im * s}; }
RT_DEV Cx cx_add(Cx a, Cx b) { return {a.re + b.re, a.im + b.im}; }
RT_DEV Cx cx_sub(Cx a, Cx b) { return {a.re - b.re, a.im - b.im}; }
// libgcc __divsc3: evaluated in double with the textbook formula, one rounding to float per part
RT_DEV Cx cx_div(Cx x, Cx y) {
    const double a = x.re, b = x.im, c = y.re, d = y.im;
    const double den = c * c + d * d;
    return {static_cast<float>((a * c + b * d) / den), static_cast<float>((b * c - a * d) / den)};
}
RT_DEV float cx_norm(Cx a) { return a.re * a.re + a.im * a.im; }   // std::norm (libstdc++ 11)
// glibc 2.35 hypotf for finite operands: sqrt in double of the exact sum of squares
RT_DEV float hypot2(float x, float y) {
    return static_cast<float>(sqrt(static_cast<double>(x) * static_cast<double>(x) + static_cast<double>(y) * static_cast<double>(y)));
}
// glibc 2.35 csqrtf, finite operands away from the overflow/underflow rescaling thresholds
// (|re|,|im| <= FLT_MAX/4 and not both < 2*FLT_MIN — Fresnel terms are O(1))
RT_DEV Cx cx_sqrt(Cx z) {
    const float re = z.re, im = z.im;
    if (im == 0.0f) {
        if (re < 0.0f) return {0.0f, copysignf(sqrtf(-re), im)};
        return {fabsf(sqrtf(re)), copysignf(0.0f, im)};
    }
    if (re == 0.0f) {
        const float r = sqrtf(0.5f * fabsf(im));
        return {r, copysignf(r, im)};
    }
    const float d = hypot2(re, im);
    float r, s;
    if (re > 0.0f) {
        r = sqrtf(0.5f * (d + re));
        s = 0.5f * (im / r);
    } else {
        s = sqrtf(0.5f * (d - re));
        r = fabsf(0.5f * (im / s));
    }
    return {r, copysignf(s, im)};
}

}  // namespace rtd
