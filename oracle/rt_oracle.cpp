// rt_oracle.cpp — CPU restatement of the reference's per-pixel trace path and photon pass.
//
// *** TEST INFRASTRUCTURE ONLY. ***  Nothing in the product path (raytracer_amd/, include/) links,
// imports or executes this file.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg load liboracle.so, and only as the checker / the timed CPU baseline.
//
// What it restates (reference = clemy/RayTracer, files relative to /root/reference):
//   src/raytracer.cpp:17-113   camera constants + per-pixel sample loops     -> render_pixel()
//   src/raytracer.cpp:115-164  castRay                                       -> cast_ray()
//   src/raytracer.cpp:166-308  texture, Phong, Fresnel, refraction, reflection
//   src/objects.cpp:9-242      Sphere / Triangle / Julia intersectors
//   src/objects.cpp:244-267    photon texture add / get
//   src/photonmap.cpp:16-146   photon emission and tracing
//   src/types.h                float32 vector / matrix / quaternion / colour arithmetic
//
// Pinned build: g++ -O2 -std=c++17 -ffp-contract=off, baseline x86-64 (no FMA, no fast-math) = the
// "O-strict" oracle of SURVEY §8c.  It is pinned against the real reference compiled the same way
// (oracle/Makefile -> oracle/_ref/, oracle/ref_harness.cpp) by tests/test_oracle_golden.py (test_oracle_vs_live_reference*)
// and the golden vectors under tests/golden/.
//
// Precision map: sub-expressions that the reference evaluates in double
// because of unqualified libm calls are evaluated in double here too and marked "Y8".
//
// The one deliberate restructuring: cast_ray() finds the last accepted hit first and shades it
// once.  The reference shades every successively accepted object and overwrites the result
// (src/raytracer.cpp:126-161); shading has no side effects, so the returned value is identical.
#include <algorithm>
#include <atomic>
#include <mutex>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/rt_hip.h"

namespace {

constexpr float PI_F = 3.14159265358979323846f;   // src/types.h:19
constexpr float INFINITE_F = 3.402823466e+38f;    // src/types.h:20
constexpr float EPSILON_F = 0.0001f;              // src/types.h:21

struct V3 {
    float x, y, z;
};
struct V2 {
    float x, y;
};
struct Col {
    float r, g, b, a;
};
struct Quat {
    float r, a, b, c;
};

// ---- src/types.h:70-109 -------------------------------------------------------------------------
inline float hypot3(float x, float y, float z) {   // std::hypot(x,y,z), libstdc++ __hypot3<float>
    x = std::fabs(x);
    y = std::fabs(y);
    z = std::fabs(z);
    const float a = x < y ? (y < z ? z : y) : (x < z ? z : x);
    if (a == 0.0f) return 0.0f;
    return a * std::sqrt((x / a) * (x / a) + (y / a) * (y / a) + (z / a) * (z / a));
}
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 mul(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(V3 v) { return hypot3(v.x, v.y, v.z); }
inline V3 normalized(V3 v) {
    const float len = length(v);
    if (len == 0.0f) return {0.0f, 0.0f, 0.0f};
    return {v.x / len, v.y / len, v.z / len};
}
// src/types.h:188-202
inline V3 xform_point(const rt_mat34& m, V3 p) {
    return {p.x * m.m[0] + p.y * m.m[1] + p.z * m.m[2] + m.m[3],
            p.x * m.m[4] + p.y * m.m[5] + p.z * m.m[6] + m.m[7],
            p.x * m.m[8] + p.y * m.m[9] + p.z * m.m[10] + m.m[11]};
}
inline V3 xform_dir(const rt_mat34& m, V3 p) {
    return {p.x * m.m[0] + p.y * m.m[1] + p.z * m.m[2], p.x * m.m[4] + p.y * m.m[5] + p.z * m.m[6],
            p.x * m.m[8] + p.y * m.m[9] + p.z * m.m[10]};
}
inline V3 v3(const float* p) { return {p[0], p[1], p[2]}; }

// src/types.h:266-305
inline Col cadd(Col a, Col b) { return {a.r + b.r, a.g + b.g, a.b + b.b, a.a + b.a}; }
inline Col cmul(Col a, float s) { return {a.r * s, a.g * s, a.b * s, a.a * s}; }
inline Col cdiv(Col a, float s) { return {a.r / s, a.g / s, a.b / s, a.a / s}; }
inline Col cmulc(Col a, Col b) { return {a.r * b.r, a.g * b.g, a.b * b.b, a.a * b.a}; }
inline Col col4(const float* p) { return {p[0], p[1], p[2], p[3]}; }

// src/types.h:341-356
struct Ray {
    V3 o, d;
};
inline Ray make_ray(V3 origin, V3 direction) { return {origin, normalized(direction)}; }

// src/objects.h:10-16
struct Hit {
    float distance;
    V3 point, normal;
    V2 uv, photon;
};

// static_cast<u32>(float) as g++ emits it on x86-64: cvttss2si to 64 bits, low 32 bits kept.
// Negative inputs are formally UB in the reference; this is what its binaries do.
inline uint32_t cast_u32(float f) {
    if (!(f > -9.2e18f && f < 9.2e18f)) return 0u;   // NaN / out of range -> 0x8000000000000000
    return static_cast<uint32_t>(static_cast<int64_t>(f));
}
inline uint32_t clamp_u32(uint32_t v, uint32_t lo, uint32_t hi) { return v < lo ? lo : (hi < v ? hi : v); }
inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (hi < v ? hi : v); }   // std::clamp

struct Counters {
    uint64_t primary = 0, secondary = 0, shadow = 0, photon_segments = 0, photon_deposits = 0;
    uint64_t sphere_tests = 0, triangle_tests = 0, julia_queries = 0, julia_steps = 0, julia_de_iterations = 0;
    uint64_t julia_normals = 0, shaded_hits = 0, phong_lights = 0, fresnel = 0, texture_lookups = 0;
    void add(const Counters& o) {
        primary += o.primary;
        secondary += o.secondary;
        shadow += o.shadow;
        photon_segments += o.photon_segments;
        photon_deposits += o.photon_deposits;
        sphere_tests += o.sphere_tests;
        triangle_tests += o.triangle_tests;
        julia_queries += o.julia_queries;
        julia_steps += o.julia_steps;
        julia_de_iterations += o.julia_de_iterations;
        julia_normals += o.julia_normals;
        shaded_hits += o.shaded_hits;
        phong_lights += o.phong_lights;
        fresnel += o.fresnel;
        texture_lookups += o.texture_lookups;
    }
};

// ---- src/objects.cpp:9-50 ------------------------------------------------------------------------
bool sphere_intersect(const rt_sphere& s, const Ray& ray, float max_distance, Hit* out, Counters& n) {
    n.sphere_tests++;
    const V3 o = xform_point(s.world_to_object, ray.o);
    const V3 d = normalized(xform_dir(s.world_to_object, ray.d));
    float object_max = INFINITE_F;
    if (max_distance != INFINITE_F) {
        object_max = length(sub(o, xform_point(s.world_to_object, add(ray.o, mul(ray.d, max_distance)))));
    }
    const V3 center = v3(s.center);
    const V3 v = sub(o, center);
    const float b = dot(v, d);
    const float c = dot(v, v) - s.radius * s.radius;
    const float h = b * b - c;
    if (h < 0.0f) return false;
    float distance = static_cast<float>(static_cast<double>(-b) - std::sqrt(static_cast<double>(h)));   // Y8
    if (distance > object_max) return false;
    if (distance < 0) {
        distance = static_cast<float>(static_cast<double>(-b) + std::sqrt(static_cast<double>(h)));    // Y8
        if (distance < 0 || distance > object_max) return false;
    }
    const V3 p = add(o, mul(d, distance));
    const V3 nobj = normalized(sub(p, center));
    const V2 uv = {0.5f + std::atan2(nobj.x, nobj.z) / (2.0f * PI_F), 0.5f - std::asin(nobj.y) / PI_F};
    const V3 world_point = xform_point(s.object_to_world, p);
    out->distance = length(sub(ray.o, world_point));
    out->point = world_point;
    out->normal = normalized(xform_point(s.object_to_world_normals, nobj));
    out->uv = uv;
    out->photon = uv;
    return true;
}

// ---- src/objects.cpp:52-99 -----------------------------------------------------------------------
bool triangle_intersect(const rt_triangle& t, const Ray& ray, float max_distance, Hit* out, Counters& n) {
    n.triangle_tests++;
    const V3 p0 = v3(t.position[0]);
    const V3 e1 = sub(v3(t.position[1]), p0);
    const V3 e2 = sub(v3(t.position[2]), p0);
    const V3 h = cross(ray.d, e2);
    const float a = dot(e1, h);
    const V3 s = sub(ray.o, p0);
    const float u = dot(s, h) / a;
    if (u < -EPSILON_F || u > 1.0f) return false;
    const V3 q = cross(s, e1);
    const float v = dot(ray.d, q) / a;
    if (v < -EPSILON_F || u + v > 1.0f) return false;
    const float distance = dot(e2, q) / a;
    if (distance < 0 || distance > max_distance) return false;
    const float w = 1.0f - (u + v);
    out->distance = distance;
    out->point = add(ray.o, mul(ray.d, distance));
    out->normal = normalized(add(add(mul(v3(t.normal[0]), w), mul(v3(t.normal[1]), u)), mul(v3(t.normal[2]), v)));
    out->uv = {t.uv[0][0] * w + t.uv[1][0] * u + t.uv[2][0] * v, t.uv[0][1] * w + t.uv[1][1] * u + t.uv[2][1] * v};
    out->photon = {w, u};
    return true;
}

// ---- src/objects.cpp:101-242 ---------------------------------------------------------------------
constexpr uint32_t JULIA_SEARCH_ITERATIONS = 10240;
constexpr float JULIA_CONVERGENCE_LIMIT = 0.0001f;
constexpr float JULIA_DIVERGENCE_LIMIT = 10000.0f;
constexpr uint32_t JULIA_DISTANCE_ITERATIONS = 10000;
constexpr float JULIA_GRADIENT_DIFF = 0.005f;
constexpr uint32_t JULIA_GRADIENT_ITERATIONS = 8;

inline Quat qmul(Quat l, Quat r) {   // Hamilton product as written, src/types.h:246-253
    return {l.r * r.r - l.a * r.a - l.b * r.b - l.c * r.c, l.r * r.a + l.a * r.r + l.b * r.c - l.c * r.b,
            l.r * r.b - l.a * r.c + l.b * r.r + l.c * r.a, l.r * r.c + l.a * r.b - l.b * r.a + l.c * r.r};
}
inline Quat qadd(Quat l, Quat r) { return {l.r + r.r, l.a + r.a, l.b + r.b, l.c + r.c}; }
inline float qsqlen(Quat q) { return q.r * q.r + q.a * q.a + q.b * q.b + q.c * q.c; }

float julia_distance(const rt_julia& j, Quat z0, uint32_t iterations, Counters& n) {
    const Quat c = {j.c[0], j.c[1], j.c[2], j.c[3]};
    Quat z = z0;
    float d2 = 1.0f;
    float m2 = qsqlen(z);
    for (uint32_t i = 0; i < iterations; i++) {
        n.julia_de_iterations++;
        d2 *= 4.0f * m2;
        z = qadd(qmul(z, z), c);
        m2 = qsqlen(z);
        if (m2 > 1e10f) break;
    }
    // Y8: sqrt/log are the double versions; m2/d2 is a float division
    const double tail = std::sqrt(static_cast<double>(m2 / d2)) * static_cast<double>(0.5f) *
                        std::log(std::sqrt(static_cast<double>(m2)));
    return static_cast<float>(tail);
}

V3 julia_normal(const rt_julia& j, Quat pos, float diff, Counters& n) {
    n.julia_normals++;
    const Quat probes[6] = {{diff, 0, 0, 0}, {-diff, 0, 0, 0}, {0, diff, 0, 0},
                            {0, -diff, 0, 0}, {0, 0, diff, 0}, {0, 0, -diff, 0}};
    float dist[6];
    for (int k = 0; k < 6; k++) dist[k] = julia_distance(j, qadd(pos, probes[k]), JULIA_GRADIENT_ITERATIONS, n);
    return normalized({dist[0] - dist[1], dist[2] - dist[3], dist[4] - dist[5]});
}

// Diagnostic only (oracle_march_histogram): marches by length.  A march is a chain of dependent steps, so the longest
// ones bound the latency of a frame whatever the machine's throughput.
thread_local uint64_t t_sample_longest_march = 0;      // longest single march of the sample being rendered
thread_local std::vector<uint32_t>* t_march_log = nullptr;   // oracle_pixel_marches: steps of every march, bit 31 = shadow ray
// Diagnostic (oracle_pixel_rays): every castRay of the pixel being rendered, 12 floats a piece:
// recursion, wavelength, origin (3), direction (3), found, object, distance, cosine.
thread_local std::vector<float>* t_ray_log = nullptr;
std::atomic<uint64_t> g_march_hist[16];        // marches of 2^(k-1) < steps <= 2^k
std::atomic<uint64_t> g_march_hist_steps[16];  // ... their steps
std::atomic<uint64_t> g_march_longest_iterations{0};   // estimator iterations of the march with the most of them
void note_march(uint32_t steps, uint64_t iterations) {
    int k = 0;
    while ((1u << k) < steps && k < 15) k++;
    g_march_hist[k].fetch_add(1, std::memory_order_relaxed);
    g_march_hist_steps[k].fetch_add(steps, std::memory_order_relaxed);
    uint64_t seen = g_march_longest_iterations.load(std::memory_order_relaxed);
    while (iterations > seen && !g_march_longest_iterations.compare_exchange_weak(seen, iterations, std::memory_order_relaxed)) {}
}

bool julia_intersect(const rt_julia& j, const Ray& ray, float max_distance, Hit* out, Counters& n) {
    n.julia_queries++;
    const V3 position = v3(j.position);
    V3 p = mul(sub(xform_point(j.world_to_object, ray.o), position), 1.0f / j.scale);
    const V3 d = normalized(xform_dir(j.world_to_object, ray.d));
    const float bound = sqrtf(3);
    if (length(p) > bound) {
        const float b = dot(p, d);
        const float c = dot(p, p) - bound * bound;
        const float h = b * b - c;
        if (h <= 0.0f) return false;
        const float distance = static_cast<float>(static_cast<double>(-b) - std::sqrt(static_cast<double>(h)));   // Y8
        if (distance < 0) return false;
        p = add(p, mul(d, distance));
    }
    float distance = 0.0f;
    uint32_t steps = 0;
    const uint64_t iterations_before = n.julia_de_iterations;
    for (uint32_t i = 0; i < JULIA_SEARCH_ITERATIONS; i++) {
        n.julia_steps++;
        steps++;
        distance = julia_distance(j, {p.x, p.y, p.z, j.cut_plane}, JULIA_DISTANCE_ITERATIONS, n);
        if (i == 0 && distance < JULIA_CONVERGENCE_LIMIT) {
            distance = 100 * JULIA_CONVERGENCE_LIMIT;
        } else if (distance < JULIA_CONVERGENCE_LIMIT || distance > JULIA_DIVERGENCE_LIMIT) {
            break;
        }
        p = add(p, mul(d, distance));
    }
    note_march(steps, n.julia_de_iterations - iterations_before);
    if (t_march_log != nullptr) t_march_log->push_back(steps | (max_distance != INFINITE_F ? 0x80000000u : 0u));
    if (steps > t_sample_longest_march) t_sample_longest_march = steps;
    if (distance >= JULIA_CONVERGENCE_LIMIT) return false;
    V3 normal = julia_normal(j, {p.x, p.y, p.z, j.cut_plane}, JULIA_GRADIENT_DIFF, n);
    if (dot(normal, d) > 0.0f) normal = mul(normal, -1.0f);
    const V3 point = xform_point(j.object_to_world, add(mul(p, j.scale), position));
    const float world_distance = length(sub(point, ray.o));
    if (world_distance < EPSILON_F || world_distance > max_distance) return false;
    out->distance = world_distance;
    out->point = point;
    out->normal = normalized(xform_point(j.object_to_world_normals, normal));
    out->uv = {0, 0};
    out->photon = {0, 0};
    return true;
}

// ---- scene view + photon textures -----------------------------------------------------------------
struct PhotonMaps {
    uint32_t size = 0;
    std::vector<std::vector<Col>> per_object;   // empty vector = never allocated (src/objects.cpp:245-247)
};

struct Tracer {
    const rt_scene_desc& sc;
    const PhotonMaps* photons;
    Counters n;

    bool intersect(uint32_t object, const Ray& ray, float max_distance, Hit* out) {   // src/objects.h:124-126
        const rt_object& o = sc.objects[object];
        switch (o.kind) {
            case RT_OBJ_SPHERE: return sphere_intersect(sc.spheres[o.index], ray, max_distance, out, n);
            case RT_OBJ_TRIANGLE: return triangle_intersect(sc.triangles[o.index], ray, max_distance, out, n);
            default: return julia_intersect(sc.julias[o.index], ray, max_distance, out, n);
        }
    }

    // src/objects.cpp:257-267
    Col get_photon(uint32_t object, V2 pos) const {
        if (photons == nullptr || photons->per_object.empty() || photons->per_object[object].empty()) return {0, 0, 0, 0};
        const uint32_t size = photons->size;
        const uint32_t x = clamp_u32(cast_u32(pos.x * (size - 1)), 0u, size - 1);
        const uint32_t y = clamp_u32(cast_u32(pos.y * (size - 1)), 0u, size - 1);
        return photons->per_object[object][size_t(y) * size + x];
    }

    // src/raytracer.cpp:166-199
    Col texture_bilinear(const rt_texture& tex, V2 uv) {
        n.texture_lookups++;
        const uint32_t w = tex.width, h = tex.height;
        float ignored;
        uv.x = modff(uv.x, &ignored);
        uv.y = modff(uv.y, &ignored);
        const V2 px = {uv.x * (w - 1), uv.y * (h - 1)};
        auto texel = [&](uint32_t x, uint32_t y) -> Col {
            const uint8_t* t = sc.texels + tex.offset + (size_t(y) * w + x) * 4;
            return {t[0] / 255.0f, t[1] / 255.0f, t[2] / 255.0f, t[3] / 255.0f};   // src/types.h:272
        };
        const uint32_t x0 = clamp_u32(cast_u32(px.x), 0u, w - 1), y0 = clamp_u32(cast_u32(px.y), 0u, h - 1);
        const uint32_t x1 = clamp_u32(cast_u32(std::ceil(px.x)), 0u, w - 1);
        const uint32_t y1 = clamp_u32(cast_u32(std::ceil(px.y)), 0u, h - 1);
        const float fx = modff(px.x, &ignored), fy = modff(px.y, &ignored);
        const float f1 = (1 - fx) * (1 - fy), f2 = fx * (1 - fy), f3 = (1 - fx) * fy, f4 = fx * fy;
        return cadd(cadd(cadd(cmul(texel(x0, y0), f1), cmul(texel(x1, y0), f2)), cmul(texel(x0, y1), f3)),
                    cmul(texel(x1, y1), f4));
    }

    // src/raytracer.cpp:201-238
    Col calc_phong(const Ray& ray, const Hit& hit, const rt_material& m) {
        const V3 point = hit.point, normal = hit.normal;
        Col rad = {0, 0, 0, 0};
        const Col mat_color = m.texture < 0 ? col4(m.color) : texture_bilinear(sc.textures[m.texture], hit.uv);
        rad = cadd(rad, cmul(cmulc(col4(sc.ambient), mat_color), m.ka));
        for (uint32_t li = 0; li < sc.n_lights; li++) {
            const rt_light& light = sc.lights[li];
            const V3 lpos = v3(light.position);
            Ray light_ray = light.type == RT_LIGHT_PARALLEL ? make_ray(point, mul(lpos, -1.0f)) : make_ray(point, sub(lpos, point));
            light_ray.o = add(light_ray.o, mul(normal, EPSILON_F));
            const float light_distance = light.type == RT_LIGHT_PARALLEL ? INFINITE_F : length(sub(lpos, light_ray.o));
            n.shadow++;
            n.phong_lights++;
            bool shadowed = false;
            for (uint32_t k = 0; k < sc.n_objects; k++) {
                Hit sh;
                if (intersect(k, light_ray, light_distance, &sh) && dot(light_ray.d, sh.normal) < 0.0f) {
                    shadowed = true;
                    break;   // std::none_of stops at the first front-facing blocker
                }
            }
            if (!shadowed) {
                const Col power = col4(light.power);
                const Col diffuse = cmul(cmul(cmulc(power, mat_color), std::max(dot(light_ray.d, normal), 0.0f)), m.kd);
                const V3 refl = normalized(sub(mul(mul(normal, dot(light_ray.d, normal)), 2), light_ray.d));
                const float spec_base = std::max(dot(refl, mul(ray.d, -1)), 0.0f);
                const float spec_pow = static_cast<float>(std::pow(static_cast<double>(spec_base), static_cast<double>(m.exponent)));   // Y8
                const Col specular = cmul(cmul(power, spec_pow), m.ks);
                rad = cadd(rad, cadd(diffuse, specular));
            }
        }
        return rad;
    }

    // src/raytracer.cpp:243-268 (the same formulas again at src/photonmap.cpp:92-116)
    float calc_fresnel(const rt_material& m, float cos_rn, float wavelength) {
        n.fresnel++;
        std::complex<float> etai = 1;
        std::complex<float> etat = std::complex<float>(m.refraction_re, m.refraction_im) + wavelength * m.dispersion;
        if (cos_rn > 0.0f) std::swap(etai, etat);
        const std::complex<float> sint = etai / etat * sqrtf(std::max(0.0f, 1 - cos_rn * cos_rn));
        if (norm(sint) < 1.0f) {
            const std::complex<float> cost = sqrt(1.0f - sint * sint);
            const float cabs_ = fabsf(cos_rn);
            const std::complex<float> Rs = (etat * cabs_ - etai * cost) / (etat * cabs_ + etai * cost);
            const std::complex<float> Rp = (etai * cabs_ - etat * cost) / (etai * cabs_ + etat * cost);
            return (norm(Rs) + norm(Rp)) / 2;
        }
        return 1.0f;
    }

    // Direction/origin of the refracted ray; false = total internal reflection (src/raytracer.cpp:270-294)
    static bool refraction_ray(const Ray& ray, const Hit& hit, const rt_material& m, float cos_rn, float wavelength, Ray* out) {
        float cos_turned = cos_rn;
        V3 normal_turned = hit.normal;
        float index = m.refraction_re + wavelength * m.dispersion;
        bool outside = false;
        if (cos_rn <= 0.0f) {
            outside = true;
            cos_turned *= -1.0f;
            index = 1.0f / index;
        } else {
            normal_turned = mul(normal_turned, -1.0f);
        }
        const float k = 1.0f - index * index * (1.0f - cos_turned * cos_turned);
        if (!(k >= 0)) return false;
        const V3 dir = add(mul(ray.d, index), mul(normal_turned, index * cos_turned - sqrtf(k)));
        *out = make_ray(hit.point, dir);
        out->o = add(out->o, mul(hit.normal, outside ? -EPSILON_F : EPSILON_F));
        return true;
    }

    // src/raytracer.cpp:296-308
    static Ray reflection_ray(const Ray& ray, const Hit& hit, float cos_rn) {
        const bool outside = cos_rn <= 0.0f;
        Ray r = make_ray(hit.point, sub(ray.d, mul(mul(hit.normal, cos_rn), 2)));
        r.o = add(r.o, mul(hit.normal, outside ? EPSILON_F : -EPSILON_F));
        return r;
    }

    // src/raytracer.cpp:115-164
    Col cast_ray(const Ray& ray, uint32_t recursion, float wavelength) {
        if (recursion > sc.camera.max_bounces) return {0, 0, 0, 0};
        if (recursion == 0) n.primary++; else n.secondary++;
        float max_distance = INFINITE_F;
        bool found = false;
        Hit hit{};
        uint32_t hit_object = 0;
        float hit_cos = 0.0f;
        for (uint32_t k = 0; k < sc.n_objects; k++) {
            Hit h;
            if (!intersect(k, ray, max_distance, &h)) continue;
            const rt_material& m = sc.materials[sc.objects[k].material];
            const float cos_rn = clampf(dot(ray.d, h.normal), -1.0f, 1.0f);
            const float refr_norm = m.refraction_re * m.refraction_re + m.refraction_im * m.refraction_im;   // std::norm
            if (cos_rn >= 0.0f && (m.transmittance == 0.0f || refr_norm == 0.0f)) continue;   // unseen back face
            max_distance = h.distance;
            found = true;
            hit = h;
            hit_object = k;
            hit_cos = cos_rn;
        }
        if (t_ray_log != nullptr) {
            const float rec[12] = {static_cast<float>(recursion), wavelength, ray.o.x, ray.o.y, ray.o.z, ray.d.x, ray.d.y, ray.d.z,
                                   found ? 1.0f : 0.0f, static_cast<float>(hit_object), found ? hit.distance : 0.0f, hit_cos};
            t_ray_log->insert(t_ray_log->end(), rec, rec + 12);
        }
        if (!found) return col4(sc.background);

        n.shaded_hits++;
        const rt_material& m = sc.materials[sc.objects[hit_object].material];
        const float refr_norm = m.refraction_re * m.refraction_re + m.refraction_im * m.refraction_im;
        Col rad = {0, 0, 0, 0};
        if (hit_cos < 0.0f) {
            rad = cadd(rad, calc_phong(ray, hit, m));
            rad = cadd(rad, get_photon(hit_object, hit.photon));
        }
        if ((m.transmittance != 0.0f || m.reflectance != 0.0f) && refr_norm > 0.0f) {
            const float kr = calc_fresnel(m, hit_cos, wavelength);
            if (m.transmittance != 0.0f && kr < 1.0f) {
                Ray rr;
                Col child = {0, 0, 0, 0};
                if (refraction_ray(ray, hit, m, hit_cos, wavelength, &rr)) child = cast_ray(rr, recursion + 1, wavelength);
                rad = cadd(rad, cmul(child, 1.0f - kr));
            }
            if (m.reflectance != 0.0f && kr > 0.0f) {
                const Col child = cast_ray(reflection_ray(ray, hit, hit_cos), recursion + 1, wavelength);
                rad = cadd(rad, cmul(child, kr));
            }
        }
        rad.a = 1.0f;
        return rad;
    }
};

// ---- camera constants, src/raytracer.cpp:17-29 ---------------------------------------------------
struct CameraFrame {
    V2 half_fov, pixel_size;
    std::vector<float> ray_x;   // the reference accumulates rayX += pixelSize.x along the row (B10)
};
CameraFrame camera_frame(const rt_camera& cam) {
    CameraFrame f;
    const float w = static_cast<float>(cam.width), h = static_cast<float>(cam.height);
    f.half_fov = {-tanf(cam.fov), tanf(cam.fov) * (h / w)};
    f.pixel_size = {(-2.0f / w) * f.half_fov.x, (-2.0f / h) * f.half_fov.y};
    f.ray_x.resize(cam.width);
    float ray_x = f.half_fov.x + 0.5f * f.pixel_size.x;
    for (uint32_t x = 0; x < cam.width; x++) {
        f.ray_x[x] = ray_x;
        ray_x += f.pixel_size.x;
    }
    return f;
}

// The two lens-jitter draws of a sample (src/raytracer.cpp:86-89).  The reference takes them from a
// std::minstd_rand seeded by std::random_device per worker thread with rows handed out dynamically:
// its pictures with lens_size != 0 are not reproducible run to run, so there is nothing
// to restate bit for bit.  The product path draws from a counter-based hash of (pixel, sub-pixel,
// seed) instead (rt_kernels.hip: start_sample); the oracle uses the same function so that everything
// around the draw stays comparable bit for bit.  (lens_size == 0 in every pinned config: the draws
// are multiplied by zero there.)
uint32_t g_dof_seed = 0;
uint32_t lens_hash(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
V2 lens_draws(const rt_camera& cam, uint32_t x, uint32_t y, uint32_t sub) {
    const uint32_t h1 = lens_hash((y * cam.width + x) * 64u + sub + g_dof_seed * 0x9e3779b9u), h2 = lens_hash(h1 ^ 0x68bc21ebu);
    return {static_cast<float>(h1 >> 8) * (2.0f / 16777216.0f) - 1.0f, static_cast<float>(h2 >> 8) * (2.0f / 16777216.0f) - 1.0f};
}

// Diagnostic only (oracle_sample_extremes): the sample (one cast_ray from the camera: a chain of dependent queries) with the
// most Julia march steps, and the one with the most estimator iterations: {steps, iterations, queries} of each.
std::atomic<uint64_t> g_sample_most_steps[3], g_sample_most_iterations[3];
std::mutex g_sample_lock;
void note_sample(uint64_t steps, uint64_t iterations, uint64_t queries) {
    if (steps <= g_sample_most_steps[0].load(std::memory_order_relaxed) && iterations <= g_sample_most_iterations[1].load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> hold(g_sample_lock);
    if (steps > g_sample_most_steps[0]) {
        g_sample_most_steps[0] = steps; g_sample_most_steps[1] = iterations; g_sample_most_steps[2] = queries;
    }
    if (iterations > g_sample_most_iterations[1]) {
        g_sample_most_iterations[0] = steps; g_sample_most_iterations[1] = iterations; g_sample_most_iterations[2] = queries;
    }
}
thread_local uint64_t t_pixel_longest_sample = 0;
thread_local float t_pixel_sample0[2] = {0, 0};        // of the pixel's first sample: its steps, its longest march   // steps of the longest sample of the pixel being rendered
struct SampleNote {
    Tracer& tr;
    uint64_t s0, i0, q0;
    explicit SampleNote(Tracer& t) : tr(t), s0(t.n.julia_steps), i0(t.n.julia_de_iterations), q0(t.n.julia_queries) { t_sample_longest_march = 0; }
    ~SampleNote() {
        note_sample(tr.n.julia_steps - s0, tr.n.julia_de_iterations - i0, tr.n.julia_queries - q0);
        if (tr.n.julia_steps - s0 > t_pixel_longest_sample) t_pixel_longest_sample = tr.n.julia_steps - s0;
        if (t_pixel_sample0[0] < 0.0f) {
            t_pixel_sample0[0] = static_cast<float>(tr.n.julia_steps - s0);
            t_pixel_sample0[1] = static_cast<float>(t_sample_longest_march);
        }
    }
};

// One pixel: src/raytracer.cpp:55-112.
Col render_pixel(Tracer& tr, const CameraFrame& f, uint32_t x, uint32_t y) {
    const rt_scene_desc& sc = tr.sc;
    const rt_camera& cam = sc.camera;
    const float ray_y = f.half_fov.y + y * f.pixel_size.y + 0.5f * f.pixel_size.y;
    const float ray_x = f.ray_x[x];
    const uint32_t nsub = cam.supersampling;
    Col radiance = {0, 0, 0, 0};
    for (uint32_t sy = 0; sy < nsub; sy++) {
        for (uint32_t sx = 0; sx < nsub; sx++) {
            const V2 sub_disp = {2.0f * (sx + 1) / (nsub + 1) - 1.0f, 2.0f * (sy + 1) / (nsub + 1) - 1.0f};
            const V2 target_disp = {sub_disp.x * f.pixel_size.x, sub_disp.y * f.pixel_size.y};
            const V3 on_plane = {ray_x + target_disp.x, ray_y + target_disp.y, -1.0f};
            const V3 on_focus = xform_point(cam.camera_to_world, mul(on_plane, cam.focus_distance));
            const V2 draw = lens_draws(cam, x, y, sy * nsub + sx);
            const V2 lens = {(sub_disp.x + draw.x * (1.0f / nsub)) * cam.lens_size, (sub_disp.y + draw.y * (1.0f / nsub)) * cam.lens_size};
            V3 origin = xform_point(cam.camera_to_world, {0.0f, 0.0f, 0.0f});
            origin = {origin.x + lens.x, origin.y + lens.y, origin.z};
            const Ray ray = make_ray(origin, sub(on_focus, origin));
            if (sc.dispersion_mode) {
                int hue_index = 0;
                for (float hue = 0.0f; hue < 360.0f; hue += 45.0f, hue_index++) {
                    const Col hue_color = {sc.hue_rgb[hue_index][0], sc.hue_rgb[hue_index][1], sc.hue_rgb[hue_index][2], 1.0f};
                    SampleNote note(tr);
                    radiance = cadd(radiance, cdiv(cmulc(tr.cast_ray(ray, 0, hue / 180.0f - 1.0f), hue_color), 4.0f));
                }
            } else {
                SampleNote note(tr);
                radiance = cadd(radiance, tr.cast_ray(ray, 0, 0));
            }
        }
    }
    const Col previous = {0, 0, 0, 0};   // a fresh Picture is zero (src/types.h:362-365)
    return cadd(previous, cmul(radiance, 1.0f / (nsub * nsub)));
}

// ---- photon pass, src/photonmap.cpp --------------------------------------------------------------
// Diagnostic (oracle_set_deposit_counts): deposits per texel of the pass, [object][y][x] (the photon pass is sequential).
static uint32_t* g_deposit_counts = nullptr;
struct PhotonTracer {
    const rt_scene_desc& sc;
    PhotonMaps& maps;
    Tracer geo;   // shares intersect() and the Fresnel/refraction/reflection helpers

    PhotonTracer(const rt_scene_desc& s, PhotonMaps& m) : sc(s), maps(m), geo{s, nullptr, {}} {}

    // src/objects.cpp:244-255
    void add_photon(uint32_t object, V2 pos, Col rad) {
        std::vector<Col>& tex = maps.per_object[object];
        const uint32_t size = maps.size;
        if (tex.empty()) tex.assign(size_t(size) * size, Col{0, 0, 0, 0});
        const uint32_t x = clamp_u32(cast_u32(pos.x * (size - 1)), 0u, size - 1);
        const uint32_t y = clamp_u32(cast_u32(pos.y * (size - 1)), 0u, size - 1);
        Col& t = tex[size_t(y) * size + x];
        t = cadd(t, rad);
        geo.n.photon_deposits++;
        if (g_deposit_counts != nullptr) g_deposit_counts[(size_t(object) * size + y) * size + x]++;
    }

    // src/photonmap.cpp:61-146
    void cast_ray(const Ray& ray, uint32_t recursion, float wavelength, Col rad) {
        if (recursion > sc.camera.max_bounces) return;
        geo.n.photon_segments++;
        float max_distance = INFINITE_F;
        bool found = false;
        uint32_t nearest = 0;
        Hit hit{};
        for (uint32_t k = 0; k < sc.n_objects; k++) {
            Hit h;
            if (geo.intersect(k, ray, max_distance, &h)) {
                max_distance = h.distance;
                nearest = k;
                hit = h;
                found = true;
            }
        }
        if (!found) return;
        const rt_material& m = sc.materials[sc.objects[nearest].material];
        const float refr_norm = m.refraction_re * m.refraction_re + m.refraction_im * m.refraction_im;
        if (refr_norm <= 0.0f) {
            if (recursion > 0) add_photon(nearest, hit.photon, rad);
            return;
        }
        const float cos_rn = clampf(dot(ray.d, hit.normal), -1.0f, 1.0f);
        const float kr = geo.calc_fresnel(m, cos_rn, wavelength);
        if (kr < 1.0f) {
            Ray rr;
            if (Tracer::refraction_ray(ray, hit, m, cos_rn, wavelength, &rr)) cast_ray(rr, recursion + 1, wavelength, cmul(rad, 1 - kr));
        }
        cast_ray(Tracer::reflection_ray(ray, hit, cos_rn), recursion + 1, wavelength, cmul(rad, kr));
    }

    // src/photonmap.cpp:20-59.  [begin, end) selects a slice of the root-photon sequence (test support for the
    // multi-GPU split of one pass, rt_photon_generate_part_device); the default is the whole pass.
    void generate(uint64_t begin = 0, uint64_t end = UINT64_MAX) {
        uint64_t root = 0;
        for (uint32_t li = 0; li < sc.n_lights; li++) {
            const rt_light& light = sc.lights[li];
            if (light.type == RT_LIGHT_PARALLEL) continue;
            const float step = 2 * PI_F / sc.photon_steps;
            for (float phi = 0.0f; phi < 2 * PI_F; phi += step) {
                for (float theta = 0.0f; theta < PI_F; theta += step) {
                    const V3 dir = {sinf(theta) * cosf(phi), sinf(theta) * sinf(phi), cosf(theta)};
                    const Ray ray = make_ray(v3(light.position), dir);
                    if (sc.dispersion_mode) {
                        int hue_index = 0;
                        for (float hue = 0.0f; hue < 360.0f; hue += 45.0f, hue_index++, root++) {
                            if (root < begin || root >= end) continue;
                            const Col hue_color = {sc.hue_rgb[hue_index][0], sc.hue_rgb[hue_index][1], sc.hue_rgb[hue_index][2], 1.0f};
                            cast_ray(ray, 0, hue / 180.0f - 1.0f, cdiv(cmul(hue_color, sc.photon_factor), 4));
                        }
                    } else {
                        if (root >= begin && root < end) cast_ray(ray, 0, 0, cmul(Col{1.0f, 1.0f, 1.0f, 1.0f}, sc.photon_factor));
                        root++;
                    }
                }
            }
        }
    }

    // number of root photons of the whole pass (the trip counts of the float loops above)
    uint64_t count_roots() const {
        uint64_t lights = 0;
        for (uint32_t li = 0; li < sc.n_lights; li++) lights += sc.lights[li].type != RT_LIGHT_PARALLEL;
        const float step = 2 * PI_F / sc.photon_steps;
        uint64_t n_phi = 0, n_theta = 0;
        for (float phi = 0.0f; phi < 2 * PI_F; phi += step) n_phi++;
        for (float theta = 0.0f; theta < PI_F; theta += step) n_theta++;
        return lights * n_phi * n_theta * (sc.dispersion_mode ? 8 : 1);
    }
};

void export_counters(const Counters& n, rt_stats* out) {
    if (out == nullptr) return;
    std::memset(out, 0, sizeof *out);
    out->primary_rays = n.primary;
    out->secondary_rays = n.secondary;
    out->shadow_rays = n.shadow;
    out->photon_segments = n.photon_segments;
    out->photon_deposits = n.photon_deposits;
    out->sphere_tests = n.sphere_tests;
    out->triangle_tests = n.triangle_tests;
    out->julia_queries = n.julia_queries;
    out->julia_steps = n.julia_steps;
    out->julia_de_iterations = n.julia_de_iterations;
    out->julia_normals = n.julia_normals;
    out->shaded_hits = n.shaded_hits;
    out->phong_lights = n.phong_lights;
    out->fresnel_evals = n.fresnel;
    out->texture_lookups = n.texture_lookups;
}

}  // namespace

extern "C" {

struct oracle_photons {
    PhotonMaps maps;
};

// ≙ PhotonMapper::generate.  Single-threaded like the reference (deposit order is part of the result).
oracle_photons* oracle_photon_generate(const rt_scene_desc* sc, rt_stats* stats) {
    auto* p = new oracle_photons;
    p->maps.size = sc->photon_texture_size;
    p->maps.per_object.resize(sc->n_objects);
    if (sc->photon_steps > 0.0f) {
        PhotonTracer tracer(*sc, p->maps);
        tracer.generate();
        export_counters(tracer.geo.n, stats);
    } else {
        export_counters(Counters{}, stats);
    }
    return p;
}

// Slice `part` of `n_parts` of the pass (as rt_photon_generate_part_device splits it)
oracle_photons* oracle_photon_generate_part(const rt_scene_desc* sc, uint32_t part, uint32_t n_parts, rt_stats* stats) {
    auto* p = new oracle_photons;
    p->maps.size = sc->photon_texture_size;
    p->maps.per_object.resize(sc->n_objects);
    if (sc->photon_steps > 0.0f && n_parts > 0 && part < n_parts) {
        PhotonTracer tracer(*sc, p->maps);
        const uint64_t all = tracer.count_roots();
        tracer.generate(all * part / n_parts, all * (part + 1) / n_parts);
        export_counters(tracer.geo.n, stats);
    } else {
        export_counters(Counters{}, stats);
    }
    return p;
}

// Replaces the texture of object `object_index` (texture_size^2 RGBA floats): the stand-in for the
// all-reduced atlas in the multi-rank driver tests
void oracle_photon_set(oracle_photons* p, uint32_t object_index, const float* rgba) {
    const size_t n = size_t(p->maps.size) * p->maps.size;
    std::vector<Col>& tex = p->maps.per_object[object_index];
    tex.resize(n);
    std::memcpy(tex.data(), rgba, n * 4 * sizeof(float));
}

void oracle_photon_free(oracle_photons* p) { delete p; }

// Seed of the lens-jitter hash (RT_OPT_DOF_SEED of the product path); default 0
void oracle_set_dof_seed(uint32_t seed) { g_dof_seed = seed; }

// texture_size^2 RGBA floats of object `object_index`; zeros if it never received a photon
void oracle_photon_get(const oracle_photons* p, uint32_t object_index, float* rgba_out) {
    const size_t n = size_t(p->maps.size) * p->maps.size;
    const std::vector<Col>& tex = p->maps.per_object[object_index];
    if (tex.empty()) {
        std::memset(rgba_out, 0, n * 4 * sizeof(float));
    } else {
        std::memcpy(rgba_out, tex.data(), n * 4 * sizeof(float));
    }
}

// ≙ RayTracer::raytrace: renders rows [row_begin, row_end) stepping by row_step, each into its place
// of the full width*height RGBA float image `rgba_out` (other rows untouched).  `threads` workers
// pull rows from an atomic counter like the reference (src/raytracer.cpp:31-53).
void oracle_camera_ray(const rt_scene_desc* sc, uint32_t x, uint32_t y, uint32_t sx, uint32_t sy, float* out);
// Diagnostic: when set, oracle_render_rows writes every pixel's Julia march steps (all of its samples' rays) there.
static float* g_cost_map = nullptr;
void oracle_set_cost_map(float* per_pixel) { g_cost_map = per_pixel; }
static float* g_longest_map = nullptr;   // ... and the steps of its longest sample (a chain of dependent queries)
void oracle_set_longest_map(float* per_pixel) { g_longest_map = per_pixel; }
static float* g_sample0_map = nullptr;   // ... and of its first sample: {steps, longest single march}
void oracle_set_sample0_map(float* per_pixel_pairs) { g_sample0_map = per_pixel_pairs; }
// Diagnostic: Julia march steps of the primary ray through the centre of pixel (x, y) alone (nearest-hit query only).
uint64_t oracle_probe_steps(const rt_scene_desc* sc, uint32_t x, uint32_t y) {
    float r[6];
    rt_scene_desc one = *sc;
    one.camera.supersampling = 1;
    oracle_camera_ray(&one, x, y, 0, 0, r);
    Tracer tr{*sc, nullptr, {}};
    const Ray ray = make_ray(v3(r), v3(r + 3));
    for (uint32_t o = 0; o < sc->n_objects; o++) {
        Hit h{};
        if (sc->objects[o].kind == RT_OBJ_JULIA) (void)tr.intersect(o, ray, INFINITE_F, &h);
    }
    return tr.n.julia_steps;
}

void oracle_render_rows(const rt_scene_desc* sc, const oracle_photons* photons, uint32_t row_begin,
                        uint32_t row_end, uint32_t row_step, int threads, float* rgba_out, rt_stats* stats) {
    const CameraFrame frame = camera_frame(sc->camera);
    if (row_end > sc->camera.height) row_end = sc->camera.height;
    if (row_step == 0) row_step = 1;
    std::atomic<uint32_t> next{0};
    const uint32_t n_rows = row_end > row_begin ? (row_end - row_begin + row_step - 1) / row_step : 0;
    std::vector<Counters> per_thread(threads > 0 ? threads : 1);
    auto worker = [&](int id) {
        Tracer tr{*sc, photons ? &photons->maps : nullptr, {}};
        uint32_t i;
        while ((i = next.fetch_add(1, std::memory_order_relaxed)) < n_rows) {
            const uint32_t y = row_begin + i * row_step;
            for (uint32_t x = 0; x < sc->camera.width; x++) {
                const uint64_t steps_before = tr.n.julia_steps;
                t_pixel_longest_sample = 0;
                t_pixel_sample0[0] = -1.0f;
                const Col c = render_pixel(tr, frame, x, y);
                if (g_sample0_map != nullptr) {
                    g_sample0_map[(size_t(y) * sc->camera.width + x) * 2] = t_pixel_sample0[0];
                    g_sample0_map[(size_t(y) * sc->camera.width + x) * 2 + 1] = t_pixel_sample0[1];
                }
                if (g_cost_map != nullptr) g_cost_map[size_t(y) * sc->camera.width + x] = static_cast<float>(tr.n.julia_steps - steps_before);
                if (g_longest_map != nullptr) g_longest_map[size_t(y) * sc->camera.width + x] = static_cast<float>(t_pixel_longest_sample);
                float* px = rgba_out + (size_t(y) * sc->camera.width + x) * 4;
                px[0] = c.r;
                px[1] = c.g;
                px[2] = c.b;
                px[3] = c.a;
            }
        }
        per_thread[id] = tr.n;
    };
    if (threads <= 1) {
        worker(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) pool.emplace_back(worker, t);
        for (auto& t : pool) t.join();
    }
    Counters total;
    for (const Counters& c : per_thread) total.add(c);
    export_counters(total, stats);
}

void oracle_render(const rt_scene_desc* sc, const oracle_photons* photons, int threads, float* rgba_out, rt_stats* stats) {
    oracle_render_rows(sc, photons, 0, sc->camera.height, 1, threads, rgba_out, stats);
}

// ---- known-answer entry points (one function of the path each) ------------------------------------

// Object::intersect of scene object `object_index` for Ray(origin, direction) (direction is
// normalised by the Ray constructor).  out[11] = distance, point xyz, normal xyz, uv, photon uv.
int oracle_intersect(const rt_scene_desc* sc, uint32_t object_index, const float* origin, const float* direction,
                     float max_distance, float* out) {
    Tracer tr{*sc, nullptr, {}};
    Hit h{};
    const Ray ray = make_ray(v3(origin), v3(direction));
    if (!tr.intersect(object_index, ray, max_distance, &h)) return 0;
    const float vals[11] = {h.distance, h.point.x, h.point.y, h.point.z, h.normal.x, h.normal.y,
                            h.normal.z, h.uv.x,    h.uv.y,    h.photon.x, h.photon.y};
    std::memcpy(out, vals, sizeof vals);
    return 1;
}

float oracle_fresnel(const rt_material* m, float cos_rn, float wavelength) {
    rt_scene_desc dummy{};
    Tracer tr{dummy, nullptr, {}};
    return tr.calc_fresnel(*m, cos_rn, wavelength);
}

// castRay for one explicit ray; out = RGBA
void oracle_cast_ray(const rt_scene_desc* sc, const oracle_photons* photons, const float* origin, const float* direction,
                     float wavelength, float* out) {
    Tracer tr{*sc, photons ? &photons->maps : nullptr, {}};
    const Col c = tr.cast_ray(make_ray(v3(origin), v3(direction)), 0, wavelength);
    out[0] = c.r;
    out[1] = c.g;
    out[2] = c.b;
    out[3] = c.a;
}

// Camera ray of sample (x, y, sx, sy): out[6] = origin xyz, direction xyz
void oracle_camera_ray(const rt_scene_desc* sc, uint32_t x, uint32_t y, uint32_t sx, uint32_t sy, float* out) {
    const CameraFrame f = camera_frame(sc->camera);
    const rt_camera& cam = sc->camera;
    const uint32_t nsub = cam.supersampling;
    const float ray_y = f.half_fov.y + y * f.pixel_size.y + 0.5f * f.pixel_size.y;
    const V2 sub_disp = {2.0f * (sx + 1) / (nsub + 1) - 1.0f, 2.0f * (sy + 1) / (nsub + 1) - 1.0f};
    const V3 on_plane = {f.ray_x[x] + sub_disp.x * f.pixel_size.x, ray_y + sub_disp.y * f.pixel_size.y, -1.0f};
    const V3 on_focus = xform_point(cam.camera_to_world, mul(on_plane, cam.focus_distance));
    const V3 origin = xform_point(cam.camera_to_world, {0.0f, 0.0f, 0.0f});
    const Ray ray = make_ray(origin, sub(on_focus, origin));
    const float vals[6] = {ray.o.x, ray.o.y, ray.o.z, ray.d.x, ray.d.y, ray.d.z};
    std::memcpy(out, vals, sizeof vals);
}

// Photon scan tables exactly as the nested float loops of src/photonmap.cpp:38-40 produce them.
// Returns the counts; fills up to `cap` entries of each table when the pointers are non-NULL.
void oracle_photon_scan(float steps, uint32_t cap, float* phi_out, uint32_t* n_phi, float* theta_out, uint32_t* n_theta) {
    const float step = 2 * PI_F / steps;
    uint32_t np = 0, nt = 0;
    for (float phi = 0.0f; phi < 2 * PI_F; phi += step) {
        if (phi_out && np < cap) phi_out[np] = phi;
        np++;
    }
    for (float theta = 0.0f; theta < PI_F; theta += step) {
        if (theta_out && nt < cap) theta_out[nt] = theta;
        nt++;
    }
    *n_phi = np;
    *n_theta = nt;
}

// Diagnostic: marches by length since the last call (k: 2^(k-1) < steps <= 2^k): counts[16], steps[16], and the most
// estimator iterations any single march made.  Not part of any parity check.
void oracle_march_histogram(uint64_t* counts, uint64_t* steps, uint64_t* longest_iterations) {
    for (int k = 0; k < 16; k++) {
        counts[k] = g_march_hist[k].exchange(0);
        steps[k] = g_march_hist_steps[k].exchange(0);
    }
    *longest_iterations = g_march_longest_iterations.exchange(0);
}

// Diagnostic: the Julia marches of pixel (x, y), all of its samples in order: steps, bit 31 set for shadow rays.  Returns
// how many there were (at most `cap` are written).
uint32_t oracle_pixel_marches(const rt_scene_desc* sc, uint32_t x, uint32_t y, uint32_t* out, uint32_t cap) {
    std::vector<uint32_t> log;
    t_march_log = &log;
    Tracer tr{*sc, nullptr, {}};
    const CameraFrame frame = camera_frame(sc->camera);
    (void)render_pixel(tr, frame, x, y);
    t_march_log = nullptr;
    for (uint32_t i = 0; i < log.size() && i < cap; i++) out[i] = log[i];
    return static_cast<uint32_t>(log.size());
}

void oracle_set_deposit_counts(uint32_t* per_texel) { g_deposit_counts = per_texel; }

// Diagnostic: every castRay of pixel (x, y), all of its samples in order, 12 floats a piece (recursion, wavelength, origin,
// direction, found, object, distance, cosine).  Returns how many there were (at most `cap` are written).
uint32_t oracle_pixel_rays(const rt_scene_desc* sc, const oracle_photons* photons, uint32_t x, uint32_t y, float* out, uint32_t cap) {
    std::vector<float> log;
    t_ray_log = &log;
    Tracer tr{*sc, photons ? &photons->maps : nullptr, {}};
    const CameraFrame frame = camera_frame(sc->camera);
    (void)render_pixel(tr, frame, x, y);
    t_ray_log = nullptr;
    const uint32_t n = static_cast<uint32_t>(log.size() / 12);
    for (uint32_t i = 0; i < n && i < cap; i++) std::memcpy(out + 12 * i, log.data() + 12 * i, 12 * sizeof(float));
    return n;
}

// Diagnostic: {steps, iterations, Julia queries} of the sample with the most march steps and of the one with the most
// estimator iterations since the last call.  Not part of any parity check.
void oracle_sample_extremes(uint64_t* out6) {
    for (int k = 0; k < 3; k++) {
        out6[k] = g_sample_most_steps[k].exchange(0);
        out6[3 + k] = g_sample_most_iterations[k].exchange(0);
    }
}

}  // extern "C"
