// Device intersectors and shading helpers (gfx950).  Each function names the reference lines whose
// arithmetic it reproduces; see rt_device_math.h for the arithmetic contract.
#pragma once

#include "rt_device_math.h"
#include "rt_libm.h"
#include "rt_scene_dev.h"

namespace rtd {

struct HitRec {
    float dist;
    V3 point, normal;
    V2 uv, ph;
};

// ---- Möller–Trumbore without determinant epsilon or culling (src/objects.cpp:52-81) -------------
// Returns true with (u, v, t) when the reference would go on to build an Intersection.
struct TriGeom {
    V3 p0, e1, e2;
};
template <typename P>
RT_DEV TriGeom load_tri(P t) {
    return {v3(t->p0[0], t->p0[1], t->p0[2]), v3(t->e1[0], t->e1[1], t->e1[2]), v3(t->e2[0], t->e2[1], t->e2[2])};
}
RT_DEV bool tri_test(const TriGeom& t, V3 o, V3 d, float maxd, float& u, float& v, float& dist) {
    const V3 e1 = t.e1;
    const V3 e2 = t.e2;
    const V3 h = cross(d, e2);
    const float a = dot(e1, h);
    const V3 s = o - t.p0;
    u = dot(s, h) / a;
    if (u < -kEpsilon || u > 1.0f) return false;
    const V3 q = cross(s, e1);
    v = dot(d, q) / a;
    if (v < -kEpsilon || u + v > 1.0f) return false;
    dist = dot(e2, q) / a;
    if (dist < 0 || dist > maxd) return false;
    return true;
}

// The test above split for instruction-level parallelism.  Stage 1 (edge cross product, determinant,
// first barycentric numerator) is straight-line code without division, so several triangles can be
// in flight per lane; it also decides most rejections exactly, without the division:
//   u = fl(nu/a) > 1          <=>  m > |a|              (m = nu with the sign of a folded in)
//   m*3333 < -|a|              =>  u < -3e-4 < -EPSILON
// (proofs in DESIGN.md §4; a == 0, infinities and NaNs fall through to the exact stage).  Survivors
// run the reference's own sequence in tri_stage23.
struct TriStage1 {
    V3 s;
    float a, nu;
    bool maybe;
};
RT_DEV TriStage1 tri_stage1(const TriGeom& t, V3 o, V3 d) {
    TriStage1 r;
    const V3 h = cross(d, t.e2);
    r.a = dot(t.e1, h);
    r.s = o - t.p0;
    r.nu = dot(r.s, h);
    const float m = r.a < 0.0f ? -r.nu : r.nu;
    const float aa = fabsf(r.a);
    r.maybe = !(m > aa || m * 3333.0f < -aa);
    return r;
}
RT_DEV bool tri_stage23(const TriGeom& t, const TriStage1& r, V3 d, float maxd, float& u, float& v, float& dist) {
    u = r.nu / r.a;
    if (u < -kEpsilon || u > 1.0f) return false;
    const V3 q = cross(r.s, t.e1);
    v = dot(d, q) / r.a;
    if (v < -kEpsilon || u + v > 1.0f) return false;
    dist = dot(t.e2, q) / r.a;
    if (dist < 0 || dist > maxd) return false;
    return true;
}

// Interpolated (not yet normalised) vertex normal, the argument of normalized() in src/objects.cpp:86-90
template <typename P>
RT_DEV V3 tri_normal_sum(P s, float u, float v) {
    const float w = 1.0f - (u + v);
    return v3(s->n0[0], s->n0[1], s->n0[2]) * w + v3(s->n1[0], s->n1[1], s->n1[2]) * u + v3(s->n2[0], s->n2[1], s->n2[2]) * v;
}

// dot(d, normalized(nv)) where only its sign is needed (back-face rule, shadow rule): when the
// unnormalised dot product is clear of zero by 10x the worst-case rounding error of both evaluations
// the sign is returned as +-1 without the normalisation; otherwise the reference's expression is
// evaluated.  (|d| = 1; error bounds in DESIGN.md §4.)
RT_DEV float facing(V3 d, V3 nv) {
    const float c = dot(d, nv);
    const float bound = 1e-5f * (fabsf(nv.x) + fabsf(nv.y) + fabsf(nv.z));
    if (c > bound) return 1.0f;
    if (c < -bound) return -1.0f;
    return dot(d, normalized(nv));
}

// Rest of Triangle::intersect (src/objects.cpp:83-98)
RT_DEV void tri_finish(const DevTriShade& s, V3 o, V3 d, float u, float v, float dist, HitRec& h) {
    const float w = 1.0f - (u + v);
    h.dist = dist;
    h.point = o + d * dist;
    h.normal = normalized(tri_normal_sum(&s, u, v));
    h.uv = {s.uv0[0] * w + s.uv1[0] * u + s.uv2[0] * v, s.uv0[1] * w + s.uv1[1] * u + s.uv2[1] * v};
    h.ph = {w, u};
}

// ---- Sphere::intersect (src/objects.cpp:9-50) --------------------------------------------------
template <typename P>
RT_DEV bool sphere_intersect(P s, V3 o_w, V3 d_w, float maxd, HitRec& h) {
    const V3 o = xform_point(s->w2o, o_w);
    const V3 d = normalized(xform_dir(s->w2o, d_w));
    float object_max = kInfinite;
    if (maxd != kInfinite) object_max = length(o - xform_point(s->w2o, o_w + d_w * maxd));
    const V3 center = v3(s->center[0], s->center[1], s->center[2]);
    const V3 vc = o - center;
    const float b = dot(vc, d);
    const float c = dot(vc, vc) - s->radius * s->radius;
    const float disc = b * b - c;
    if (disc < 0.0f) return false;
    float t = root_minus(b, disc);
    if (t > object_max) return false;
    if (t < 0) {
        t = root_plus(b, disc);
        if (t < 0 || t > object_max) return false;
    }
    const V3 p = o + d * t;
    const V3 n = normalized(p - center);
    const V2 uv = {0.5f + glibc_atan2f(n.x, n.z) / (2.0f * kPi), 0.5f - glibc_asinf(n.y) / kPi};   // rt_libm.h: glibc's own sequences
    const V3 pw = xform_point(s->o2w, p);
    h.dist = length(o_w - pw);
    h.point = pw;
    h.normal = normalized(xform_point(s->o2wn, n));
    h.uv = uv;
    h.ph = uv;
    return true;
}

// ---- Julia (src/objects.cpp:101-242) -----------------------------------------------------------
constexpr unsigned kJuliaSearchIterations = 10240;
constexpr float kJuliaConvergence = 0.0001f;
constexpr float kJuliaDivergence = 10000.0f;
constexpr unsigned kJuliaDistanceIterations = 10000;
constexpr float kJuliaGradientDiff = 0.005f;
constexpr unsigned kJuliaGradientIterations = 8;

struct Quat {
    float r, a, b, c;
};

// Julia::estimateDistance (src/objects.cpp:122-135).  The Hamilton square is written term for term
// as src/types.h:246-253 evaluates z*z.  Exact early-out: once d2 is +inf and m2 stays finite the
// returned value is (+/-)0 whatever the remaining iterations do.
// One pass of the loop body; returns true when the loop ends by one of its breaks.
RT_DEV bool julia_iterate(Quat& z, const Quat& c, float& d2, float& m2) {
    d2 *= 4.0f * m2;
    const float nr = z.r * z.r - z.a * z.a - z.b * z.b - z.c * z.c;
    const float na = z.r * z.a + z.a * z.r + z.b * z.c - z.c * z.b;
    const float nb = z.r * z.b - z.a * z.c + z.b * z.r + z.c * z.a;
    const float nc = z.r * z.c + z.a * z.b - z.b * z.a + z.c * z.r;
    z = {nr + c.r, na + c.a, nb + c.b, nc + c.c};
    m2 = z.r * z.r + z.a * z.a + z.b * z.b + z.c * z.c;
    return m2 > 1e10f || d2 == __builtin_inff();
}
RT_DEV float julia_norm2(const Quat& z) { return z.r * z.r + z.a * z.a + z.b * z.b + z.c * z.c; }

// The same loop body on packed FP32 instructions (v_pk_mul_f32 / v_pk_add_f32: two independent IEEE
// operations per lane and instruction, no fusing), 21 VALU instructions instead of 41.  Every
// product and sum of julia_iterate is formed with the same operands in the same order; a product
// needed on both halves of different registers is simply formed twice (x*y == y*x bit for bit).
//   p = (r, a)  q = (b, c)         u = (rr, ra)  v = (aa, ar)  w = (bb, bc)  x = (cc, cb)
//   nr = ((rr - aa) - bb) - cc     na = ((ra + ar) + bc) - cb          -> u - v, then w, then x
//   nb = ((rb - ac) + br) + ca     nc = ((rc + ab) - ba) + cr          -> (rb,rc), (ab,ac), (rb,ab), (ac,rc)
// The squares in u, v, w, x of the NEW z give m2 for the exit test and are the first factors of the
// next pass, so they are kept with the state.
typedef float PkF2 __attribute__((ext_vector_type(2)));
struct JuliaPk {
    PkF2 p, q, u, v, w, x;
    float d2, m2;
};
RT_DEV void julia_pk_squares(JuliaPk& s) {
    asm volatile(
        "v_pk_mul_f32 %0, %4, %4 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %1, %4, %4 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %2, %5, %5 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %3, %5, %5 op_sel:[1,1] op_sel_hi:[1,0]"
        : "=&v"(s.u), "=&v"(s.v), "=&v"(s.w), "=&v"(s.x)
        : "v"(s.p), "v"(s.q));
    s.m2 = s.u.x + s.v.x + s.w.x + s.x.x;
}
RT_DEV JuliaPk julia_pk_begin(const Quat& z) {
    JuliaPk s;
    s.p = {z.r, z.a};
    s.q = {z.b, z.c};
    s.d2 = 1.0f;
    julia_pk_squares(s);
    return s;
}
RT_DEV bool julia_pk_iterate(JuliaPk& s, PkF2 cp, PkF2 cq) {
    s.d2 *= 4.0f * s.m2;
    PkF2 rq, aq, xb, yc, t, n;
    asm volatile(
        "v_pk_mul_f32 %0, %6, %7 op_sel:[0,0] op_sel_hi:[0,1]\n\t"   // (rb, rc)
        "v_pk_mul_f32 %1, %6, %7 op_sel:[1,0] op_sel_hi:[1,1]\n\t"   // (ab, ac)
        "v_pk_mul_f32 %2, %6, %7 op_sel:[0,0] op_sel_hi:[1,0]\n\t"   // (rb, ab)
        "v_pk_mul_f32 %3, %6, %7 op_sel:[1,1] op_sel_hi:[0,1]\n\t"   // (ac, rc)
        "v_pk_add_f32 %4, %8, %9 neg_lo:[0,1]\n\t"                    // (rr - aa, ra + ar)
        "v_pk_add_f32 %5, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"   // (rb - ac, rc + ab)
        "v_pk_add_f32 %4, %4, %10 neg_lo:[0,1]\n\t"                   // (.. - bb, .. + bc)
        "v_pk_add_f32 %5, %5, %2 neg_hi:[0,1]\n\t"                    // (.. + br, .. - ba)
        "v_pk_add_f32 %4, %4, %11 neg_lo:[0,1] neg_hi:[0,1]\n\t"      // (.. - cc, .. - cb)
        "v_pk_add_f32 %5, %5, %3"                                      // (.. + ca, .. + cr)
        : "=&v"(rq), "=&v"(aq), "=&v"(xb), "=&v"(yc), "=&v"(t), "=&v"(n)
        : "v"(s.p), "v"(s.q), "v"(s.u), "v"(s.v), "v"(s.w), "v"(s.x));
    s.p = t + cp;
    s.q = n + cq;
    julia_pk_squares(s);
    return s.m2 > 1e10f || s.d2 == __builtin_inff();
}
// One pass without the exit test (the marcher waves test m2 themselves and look at d2 once per block of four).
RT_DEV void julia_pk_iterate_only(JuliaPk& s, PkF2 cp, PkF2 cq) {
    s.d2 *= 4.0f * s.m2;
    PkF2 rq, aq, xb, yc, t, n;
    asm volatile(
        "v_pk_mul_f32 %0, %6, %7 op_sel:[0,0] op_sel_hi:[0,1]\n\t"   // (rb, rc)
        "v_pk_mul_f32 %1, %6, %7 op_sel:[1,0] op_sel_hi:[1,1]\n\t"   // (ab, ac)
        "v_pk_mul_f32 %2, %6, %7 op_sel:[0,0] op_sel_hi:[1,0]\n\t"   // (rb, ab)
        "v_pk_mul_f32 %3, %6, %7 op_sel:[1,1] op_sel_hi:[0,1]\n\t"   // (ac, rc)
        "v_pk_add_f32 %4, %8, %9 neg_lo:[0,1]\n\t"                    // (rr - aa, ra + ar)
        "v_pk_add_f32 %5, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"   // (rb - ac, rc + ab)
        "v_pk_add_f32 %4, %4, %10 neg_lo:[0,1]\n\t"                   // (.. - bb, .. + bc)
        "v_pk_add_f32 %5, %5, %2 neg_hi:[0,1]\n\t"                    // (.. + br, .. - ba)
        "v_pk_add_f32 %4, %4, %11 neg_lo:[0,1] neg_hi:[0,1]\n\t"      // (.. - cc, .. - cb)
        "v_pk_add_f32 %5, %5, %3"                                      // (.. + ca, .. + cr)
        : "=&v"(rq), "=&v"(aq), "=&v"(xb), "=&v"(yc), "=&v"(t), "=&v"(n)
        : "v"(s.p), "v"(s.q), "v"(s.u), "v"(s.v), "v"(s.w), "v"(s.x));
    s.p = t + cp;
    s.q = n + cq;
    julia_pk_squares(s);
}
// sqrt(m2/d2)*0.5f*log(sqrt(m2)): float division, everything else in double (Y8)
RT_DEV float julia_distance_tail(float m2, float d2) {
    const double tail = sqrt(static_cast<double>(m2 / d2)) * 0.5 * log(sqrt(static_cast<double>(m2)));
    return static_cast<float>(tail);
}

// The same value by a shorter route (one tail per march step and six per normal estimate).
// E = sqrt(q) * 0.5 * ln(S), S = sqrt((double)m2) correctly rounded exactly as the reference forms it,
// is evaluated in double with a relative error below 2^-50 (float-seeded Newton square root and
// reciprocal, atanh series for the logarithm of S; error measured against 80-bit arithmetic on 10^8
// operands by the model in tools/fast_tail_model.c).  The reference's own double expression (glibc
// log within an ulp, two products) is within 2^-50 of the same real number, so both round to the
// same float unless E lies within 2^-44*E of a rounding boundary.  That case, and every operand
// outside the plain range (infinities, NaNs, zeros, denormal results), returns false and the caller
// evaluates the reference's expression itself (julia_distance_tail).
RT_DEV bool julia_distance_tail_fast(float m2, float d2, float& out) {
    if (d2 == __builtin_inff() && m2 > 0.0f && m2 < __builtin_inff()) {
        // interior point: sqrt(m2/inf) = +0 times a finite logarithm, a zero with the logarithm's sign
        out = m2 < 1.0f ? -0.0f : 0.0f;
        return true;
    }
    const float q = m2 / d2;
    if (!(m2 >= 1e-30f && m2 < __builtin_inff() && q >= 1e-30f && q < 1e30f)) return false;
    const double x = static_cast<double>(q);
    double g, h;
    {
        const double y = static_cast<double>(__builtin_amdgcn_rsqf(q));
        g = x * y;
        h = 0.5 * y;
        double r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g);
        h = __builtin_fma(h, r, h);
        r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g);
        h = __builtin_fma(h, r, h);
        const double d = __builtin_fma(-g, g, x);
        g = __builtin_fma(d, h, g);   // sqrt(q)
    }
    int e;
    double m = __builtin_frexp(__builtin_sqrt(static_cast<double>(m2)), &e);   // [0.5, 1)
    if (m < 0.70710678118654752) {
        m *= 2.0;
        e -= 1;
    }
    const double num = m - 1.0, den = m + 1.0;
    double ry = static_cast<double>(__builtin_amdgcn_rcpf(static_cast<float>(den)));
    double er = __builtin_fma(-den, ry, 1.0);
    ry = __builtin_fma(ry, er, ry);
    er = __builtin_fma(-den, ry, 1.0);
    ry = __builtin_fma(ry, er, ry);
    const double s = num * ry;
    const double s2 = s * s;
    double p = 1.0 / 19.0;
    p = __builtin_fma(p, s2, 1.0 / 17.0);
    p = __builtin_fma(p, s2, 1.0 / 15.0);
    p = __builtin_fma(p, s2, 1.0 / 13.0);
    p = __builtin_fma(p, s2, 1.0 / 11.0);
    p = __builtin_fma(p, s2, 1.0 / 9.0);
    p = __builtin_fma(p, s2, 1.0 / 7.0);
    p = __builtin_fma(p, s2, 1.0 / 5.0);
    p = __builtin_fma(p, s2, 1.0 / 3.0);
    p = p * s2;
    const double lnm = 2.0 * __builtin_fma(s, p, s);
    const double ln = __builtin_fma(static_cast<double>(e), 0.69314718055994530942, lnm);
    const double big = g * 0.5 * ln;
    const float f = static_cast<float>(big);
    const unsigned bits = __float_as_uint(f);
    const unsigned ex = bits & 0x7f800000u;
    if ((bits & 0x007fffffu) == 0u || ex < (40u << 23) || ex == 0x7f800000u) return false;
    const double half_ulp = static_cast<double>(__uint_as_float(ex - (24u << 23)));
    const double resid = __builtin_fabs(big - static_cast<double>(f));
    if (!(half_ulp - resid > __builtin_fabs(big) * 0x1p-44)) return false;
    out = f;
    return true;
}
// The march's own tail: the estimator left by escape (m2 > 1e10, so ln(sqrt(m2)) is far from zero).  There
// ln(S) with S = sqrt((double)m2) rounded equals 0.5*ln(m2) to 2^-53, so E = sqrt(q)*0.25*ln(m2) needs neither the
// double square root of m2 nor a double frexp: exponent and mantissa come from the float itself.  Same decision
// rule as julia_distance_tail_fast (the float is returned only when E is clear of every rounding boundary by 2^-44
// of its value); everything else, and every operand outside m2 in [16, 1e30), goes to that function.
// Model and sweep against the reference's expression: tools/fast_tail_model.c (lean), tests/test_fast_tail_model.py.
RT_DEV bool julia_distance_tail_lean(float m2, float d2, float& out) {
    if (d2 == __builtin_inff() && m2 > 0.0f && m2 < __builtin_inff()) {
        // interior point (the estimate ended by the overflow of d2): sqrt(m2/inf) = +0 times a finite logarithm, a zero
        // with the logarithm's sign, as in julia_distance_tail_fast.  Decided here because one such lane in a marcher
        // wave's step phase (2.4 % of C4's steps: six phases in ten hold one) would send the whole wave through the
        // general form's code
        out = m2 < 1.0f ? -0.0f : 0.0f;
        return true;
    }
    if (!(m2 >= 16.0f && m2 < 1e30f)) return false;
    const float q = m2 / d2;
    if (!(q >= 1e-30f && q < 1e30f)) return false;
    const double x = static_cast<double>(q);
    double g;
    {
        const double y = static_cast<double>(__builtin_amdgcn_rsqf(q));
        g = x * y;
        double h = 0.5 * y;
        double r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g);
        h = __builtin_fma(h, r, h);
        r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g);   // sqrt(q)
    }
    int e;
    float mf = __builtin_frexpf(m2, &e);   // [0.5, 1), exact
    if (mf < 0.70710678f) {
        mf *= 2.0f;
        e -= 1;
    }
    const double m = static_cast<double>(mf);
    const double num = m - 1.0, den = m + 1.0;
    double ry = static_cast<double>(__builtin_amdgcn_rcpf(static_cast<float>(den)));
    double er = __builtin_fma(-den, ry, 1.0);
    ry = __builtin_fma(ry, er, ry);
    er = __builtin_fma(-den, ry, 1.0);
    ry = __builtin_fma(ry, er, ry);
    const double s = num * ry;
    const double s2 = s * s;
    double p = 1.0 / 19.0;
    p = __builtin_fma(p, s2, 1.0 / 17.0);
    p = __builtin_fma(p, s2, 1.0 / 15.0);
    p = __builtin_fma(p, s2, 1.0 / 13.0);
    p = __builtin_fma(p, s2, 1.0 / 11.0);
    p = __builtin_fma(p, s2, 1.0 / 9.0);
    p = __builtin_fma(p, s2, 1.0 / 7.0);
    p = __builtin_fma(p, s2, 1.0 / 5.0);
    p = __builtin_fma(p, s2, 1.0 / 3.0);
    p = p * s2;
    const double lnm = 2.0 * __builtin_fma(s, p, s);
    const double ln = __builtin_fma(static_cast<double>(e), 0.69314718055994530942, lnm);   // ln(m2)
    const double big = g * 0.25 * ln;
    const float f = static_cast<float>(big);
    const unsigned bits = __float_as_uint(f);
    const unsigned ex = bits & 0x7f800000u;
    if ((bits & 0x007fffffu) == 0u || ex < (40u << 23) || ex == 0x7f800000u) return false;
    const double half_ulp = static_cast<double>(__uint_as_float(ex - (24u << 23)));
    const double resid = __builtin_fabs(big - static_cast<double>(f));
    if (!(half_ulp - resid > __builtin_fabs(big) * 0x1p-44)) return false;
    out = f;
    return true;
}
RT_DEV float julia_distance_tail_auto(float m2, float d2) {
    float r;
    if (julia_distance_tail_fast(m2, d2, r)) return r;
    return julia_distance_tail(m2, d2);
}
// for the march steps (escaped estimates in the vast majority)
RT_DEV float julia_distance_tail_march(float m2, float d2) {
    float r;
    if (julia_distance_tail_lean(m2, d2, r)) return r;
    return julia_distance_tail_auto(m2, d2);
}

template <bool COUNT>
RT_DEV float julia_distance(Quat z, Quat c, unsigned iterations, unsigned long long& n_iter) {
    JuliaPk s = julia_pk_begin(z);
    const PkF2 cp = {c.r, c.a}, cq = {c.b, c.c};
    for (unsigned i = 0; i < iterations; i++) {
        if (COUNT) n_iter++;
        if (julia_pk_iterate(s, cp, cq)) break;
    }
    return julia_distance_tail_auto(s.m2, s.d2);
}

// Julia::estimateNormal (src/objects.cpp:140-160), object space
template <bool COUNT>
RT_DEV V3 julia_normal(V3 p, float cut, Quat c, unsigned long long& n_iter) {
    const float e = kJuliaGradientDiff;
    const float d0 = julia_distance<COUNT>({p.x + e, p.y + 0.0f, p.z + 0.0f, cut + 0.0f}, c, kJuliaGradientIterations, n_iter);
    const float d1 = julia_distance<COUNT>({p.x + -e, p.y + 0.0f, p.z + 0.0f, cut + 0.0f}, c, kJuliaGradientIterations, n_iter);
    const float d2 = julia_distance<COUNT>({p.x + 0.0f, p.y + e, p.z + 0.0f, cut + 0.0f}, c, kJuliaGradientIterations, n_iter);
    const float d3 = julia_distance<COUNT>({p.x + 0.0f, p.y + -e, p.z + 0.0f, cut + 0.0f}, c, kJuliaGradientIterations, n_iter);
    const float d4 = julia_distance<COUNT>({p.x + 0.0f, p.y + 0.0f, p.z + e, cut + 0.0f}, c, kJuliaGradientIterations, n_iter);
    const float d5 = julia_distance<COUNT>({p.x + 0.0f, p.y + 0.0f, p.z + -e, cut + 0.0f}, c, kJuliaGradientIterations, n_iter);
    return normalized(v3(d0 - d1, d2 - d3, d4 - d5));
}

// First part of Julia::intersect (src/objects.cpp:164-198): ray to object space and the jump to the
// bounding sphere.  Returns false on a miss of the bounding sphere.
// (Measured out of line, round 3: the shader waves of trace_julia_kernel then spill nothing — 153 VGPRs, 19 spilled with it
// inlined — but the call's own traffic, arguments by reference and the callee's saved registers, is larger than the spills
// were: C4 1.53 GB of HBM traffic against 0.88 GB, 26.76 against 26.58 ms in a same-run A/B.  Inline it stays.)
RT_DEV bool julia_enter(const DevJulia& j, V3 o_w, V3 d_w, V3& p, V3& d) {
    p = (xform_point(j.w2o, o_w) - v3(j.position[0], j.position[1], j.position[2])) * j.inv_scale;
    d = normalized(xform_dir(j.w2o, d_w));
    const float bound = sqrtf(3.0f);
    if (length(p) > bound) {
        const float b = dot(p, d);
        const float c = dot(p, p) - bound * bound;
        const float disc = b * b - c;
        if (disc <= 0.0f) return false;
        const float t = root_minus(b, disc);
        if (t < 0) return false;
        p = p + d * t;
    }
    return true;
}

// Last part of Julia::intersect (src/objects.cpp:223-241) for a converged march position p.
template <bool COUNT>
RT_DEV bool julia_finish(const DevJulia& j, V3 o_w, V3 p, V3 d, float maxd, HitRec& h, unsigned long long& n_iter) {
    V3 n = julia_normal<COUNT>(p, j.cut_plane, {j.c[0], j.c[1], j.c[2], j.c[3]}, n_iter);
    if (dot(n, d) > 0.0f) n = n * -1.0f;
    const V3 pw = xform_point(j.o2w, p * j.scale + v3(j.position[0], j.position[1], j.position[2]));
    const float dist = length(pw - o_w);
    if (dist < kEpsilon || dist > maxd) return false;
    h.dist = dist;
    h.point = pw;
    h.normal = normalized(xform_point(j.o2wn, n));
    h.uv = {0.0f, 0.0f};
    h.ph = {0.0f, 0.0f};
    return true;
}

// One march step (src/objects.cpp:206-217).  Returns 0 = keep marching, 1 = converged (hit
// candidate), 2 = gave up (miss).  `i` is the step index before the call.
RT_DEV int julia_advance(float dist, V3& p, V3 d, unsigned& i) {
    if (i == 0 && dist < kJuliaConvergence) {
        dist = 100 * kJuliaConvergence;
    } else if (dist < kJuliaConvergence || dist > kJuliaDivergence) {
        return dist >= kJuliaConvergence ? 2 : 1;
    }
    p = p + d * dist;
    i++;
    if (i >= kJuliaSearchIterations) return dist >= kJuliaConvergence ? 2 : 1;
    return 0;
}
template <bool COUNT>
RT_DEV int julia_step(const DevJulia& j, V3& p, V3 d, unsigned& i, unsigned long long& n_iter) {
    JuliaPk s = julia_pk_begin({p.x, p.y, p.z, j.cut_plane});
    const PkF2 cp = {j.c[0], j.c[1]}, cq = {j.c[2], j.c[3]};
    for (unsigned k = 0; k < kJuliaDistanceIterations; k++) {
        if (COUNT) n_iter++;
        if (julia_pk_iterate(s, cp, cq)) break;
    }
    return julia_advance(julia_distance_tail_auto(s.m2, s.d2), p, d, i);
}

// Whole Julia::intersect in one go (used where no wave-level scheduling is worth it: the
// known-answer kernel and the photon pass).
template <bool COUNT>
RT_DEV bool julia_intersect_inline(const DevJulia& j, V3 o_w, V3 d_w, float maxd, HitRec& h,
                                   unsigned long long& n_steps, unsigned long long& n_iter) {
    V3 p, d;
    if (!julia_enter(j, o_w, d_w, p, d)) return false;
    unsigned i = 0;
    int r;
    do {
        if (COUNT) n_steps++;
        r = julia_step<COUNT>(j, p, d, i, n_iter);
    } while (r == 0);
    if (r == 2) return false;
    return julia_finish<COUNT>(j, o_w, p, d, maxd, h, n_iter);
}

// ---- texture (src/raytracer.cpp:166-199) -------------------------------------------------------
// Color(UColor) = u8 / 255.0f per channel (src/types.h:272), formed on the fly from the RGBA8 texel: q = x*r with
// r = fl(1/255) corrected by one residual step equals the correctly rounded quotient for every x in 0..255
// (checked exhaustively: tests/test_unorm8.py; x*r alone is wrong for 126
// of them).  Textures stay 4 bytes per texel in HBM/L2 instead of 16.
RT_DEV float unorm8(unsigned x) {
    const float xf = static_cast<float>(x);
    const float r = 1.0f / 255.0f;
    const float a = xf * r;
    const float e = __builtin_fmaf(-a, 255.0f, xf);
    return __builtin_fmaf(e, r, a);
}
RT_DEV Col texel_color(unsigned t) { return {unorm8(t & 255u), unorm8((t >> 8) & 255u), unorm8((t >> 16) & 255u), unorm8(t >> 24)}; }
RT_DEV Col texture_bilinear(const DevTexture& tex, const uint32_t* __restrict__ texels, V2 uv) {
    const unsigned w = tex.width, hgt = tex.height;
    uv.x = frac_part(uv.x);
    uv.y = frac_part(uv.y);
    const float px = uv.x * static_cast<float>(w - 1), py = uv.y * static_cast<float>(hgt - 1);
    const unsigned x0 = clamp_u32(cast_u32(px), 0u, w - 1), y0 = clamp_u32(cast_u32(py), 0u, hgt - 1);
    const unsigned x1 = clamp_u32(cast_u32(ceilf(px)), 0u, w - 1), y1 = clamp_u32(cast_u32(ceilf(py)), 0u, hgt - 1);
    const float fx = frac_part(px), fy = frac_part(py);
    const float f1 = (1 - fx) * (1 - fy), f2 = fx * (1 - fy), f3 = (1 - fx) * fy, f4 = fx * fy;
    const uint32_t* base = texels + tex.first_texel;
    const uint32_t t1 = base[static_cast<size_t>(y0) * w + x0], t2 = base[static_cast<size_t>(y0) * w + x1];
    const uint32_t t3 = base[static_cast<size_t>(y1) * w + x0], t4 = base[static_cast<size_t>(y1) * w + x1];
    return texel_color(t1) * f1 + texel_color(t2) * f2 + texel_color(t3) * f3 + texel_color(t4) * f4;
}

// ---- Fresnel (src/raytracer.cpp:243-268, again at src/photonmap.cpp:92-116) ---------------------
// Out of line: its double-precision complex arithmetic is a register peak of the kernels that inline everything else
// (trace_julia_kernel spills 74 VGPRs with it out of line, 132 with it inlined).
//
// A REAL index (no extinction coefficient: the scene format's default, C2's glass) makes every imaginary part of the
// computation a zero of one sign or the other, and the real parts those of real arithmetic: __divsc3's
// (a*c + b*d)/(c*c + d*d) in double becomes (a*c)/(c*c) with both products exact, a correctly rounded double of
// the real number a/c, whose rounding to float is the float quotient itself (53 >= 2*24 + 2 bits: double rounding
// is innocuous); csqrtf takes its im == 0, re > 0 branch (1 - S*S >= 2^-24 when S*S < 1); products with a zero
// imaginary part add or subtract a zero from a non-zero real part; std::norm adds +0.  So for a finite positive
// index the whole term is float arithmetic, bit for bit (test_fresnel_of_real_indices_against_the_oracle); anything
// else (a complex, non-positive, huge or NaN index, a NaN cosine) takes the complex route below.
__device__ __attribute__((noinline)) float fresnel_kr(float refr_re, float refr_im, float dispersion, float cos_rn, float wavelength) {
    {
        const float index = refr_re + wavelength * dispersion;
        if (refr_im == 0.0f && index >= 1e-3f && index <= 1e3f && fabsf(cos_rn) <= 1.0f) {
            const bool inside = cos_rn > 0.0f;
            const float ei = inside ? index : 1.0f, et = inside ? 1.0f : index;
            const float one_minus = 1 - cos_rn * cos_rn;
            const float sin_i = sqrtf(0.0f < one_minus ? one_minus : 0.0f);
            const float st = (ei / et) * sin_i;
            const float st2 = st * st;
            if (st2 < 1.0f) {
                const float ct = sqrtf(-st2 + 1.0f);
                const float ca = fabsf(cos_rn);
                const float tc = et * ca, ic = ei * ca, i_ct = ei * ct, t_ct = et * ct;
                const float rs = (tc - i_ct) / (tc + i_ct);
                const float rp = (ic - t_ct) / (ic + t_ct);
                return (rs * rs + rp * rp) / 2;
            }
            return 1.0f;
        }
    }
    Cx etai = {1.0f, 0.0f};
    Cx etat = {refr_re + wavelength * dispersion, refr_im};
    if (cos_rn > 0.0f) {
        const Cx t = etai;
        etai = etat;
        etat = t;
    }
    const float one_minus = 1 - cos_rn * cos_rn;
    const float sin_i = sqrtf(0.0f < one_minus ? one_minus : 0.0f);   // std::max(0.0f, x)
    const Cx sint = cx_scale(cx_div(etai, etat), sin_i);
    if (cx_norm(sint) < 1.0f) {
        const Cx ss = cx_mul(sint, sint);
        const Cx cost = cx_sqrt({-ss.re + 1.0f, -ss.im});   // operator-(T, complex): -y then += x
        const float ca = fabsf(cos_rn);
        const Cx tc = cx_scale(etat, ca), ic = cx_scale(etai, ca);
        const Cx i_cost = cx_mul(etai, cost), t_cost = cx_mul(etat, cost);
        const Cx rs = cx_div(cx_sub(tc, i_cost), cx_add(tc, i_cost));
        const Cx rp = cx_div(cx_sub(ic, t_cost), cx_add(ic, t_cost));
        return (cx_norm(rs) + cx_norm(rp)) / 2;
    }
    return 1.0f;
}

// ---- refraction / reflection rays (src/raytracer.cpp:270-308) ----------------------------------
RT_DEV bool refraction_ray(V3 dir, V3 point, V3 normal, float refr_re, float dispersion, float cos_rn,
                           float wavelength, V3& o, V3& d) {
    float cos_t = cos_rn;
    V3 nt = normal;
    float index = refr_re + wavelength * dispersion;
    bool outside = false;
    if (cos_rn <= 0.0f) {
        outside = true;
        cos_t *= -1.0f;
        index = 1.0f / index;
    } else {
        nt = nt * -1.0f;
    }
    const float k = 1.0f - index * index * (1.0f - cos_t * cos_t);
    if (!(k >= 0)) return false;
    d = normalized(dir * index + nt * (index * cos_t - sqrtf(k)));
    o = point + normal * (outside ? -kEpsilon : kEpsilon);
    return true;
}

RT_DEV void reflection_ray(V3 dir, V3 point, V3 normal, float cos_rn, V3& o, V3& d) {
    const bool outside = cos_rn <= 0.0f;
    d = normalized(dir - normal * cos_rn * 2.0f);
    o = point + normal * (outside ? kEpsilon : -kEpsilon);
}

// Texel of a photon texture (src/objects.cpp:249-252, 262-265)
RT_DEV unsigned photon_texel(V2 pos, unsigned size) {
    const float s1 = static_cast<float>(size - 1);
    const unsigned x = clamp_u32(cast_u32(pos.x * s1), 0u, size - 1);
    const unsigned y = clamp_u32(cast_u32(pos.y * s1), 0u, size - 1);
    return y * size + x;
}

}  // namespace rtd
