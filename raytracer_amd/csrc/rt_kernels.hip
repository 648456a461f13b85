// rt_kernels.hip — the trace path as CDNA4 (gfx950) kernels.
//
// Replaces (reference = clemy/RayTracer, /root/reference):
//   RayTracer::Instance::Thread::raytraceLine / castRay / calcPhong / calcFresnel / calcRefraction /
//   calcReflection                                   src/raytracer.cpp:55-308
//   PhotonMapper::generate / castRay                 src/photonmap.cpp:20-146
//
// Execution model (not a translation of the reference's recursion):
//   * one lane = one primary sample (pixel, sub-pixel, hue) at a time; persistent waves pull 8x8-pixel
//     tiles from a global queue; a wave finishes its tile before it takes the next (RT_REFILL_MIN);
//   * the recursion is a per-lane post-order DFS with an explicit frame stack in HBM
//     ([frame][field][lane], coalesced), so child radiance is combined in the reference's order;
//   * the expensive operation — one ray against every scene object — is a single wave-uniform
//     object loop shared by nearest-hit, shadow and photon queries: geometry records are read with
//     scalar loads into SGPRs, every lane with a pending query of any kind takes part;
//   * Julia sets are marched outside that loop by dedicated marcher waves fed through LDS rings
//     (trace_julia_kernel);
//   * per-sample results go to a [sample][pixel] buffer in HBM and are summed per pixel in the
//     reference's order by a second kernel.
#include <hip/hip_runtime.h>

#include "rt_intersect.h"

namespace rtd {

// register budget of the trace kernel: waves per SIMD the compiler must leave room for
#ifndef RT_PHOTON_WAVES_PER_SIMD
#define RT_PHOTON_WAVES_PER_SIMD 3
#endif
// idle lanes a wave collects before it takes new samples.  64 = a wave takes a whole 8x8 tile, finishes
// it and only then takes the next one: measured against refilling every lane as soon as it is idle
// (1), C2 2.58 -> 2.12 ms and C3 19.3 -> 12.1 ms — lanes that start together stay in the same kind of
// phase (trace, shadow 0, shadow 1 ...), so a resolve phase runs one branch instead of all of them,
// which is worth more than the idle lanes cost
#ifndef RT_JULIA_SHADER_PRIO
#define RT_JULIA_SHADER_PRIO 0
#endif
#ifndef RT_REFILL_MIN
#define RT_REFILL_MIN 64
#endif
#ifndef RT_TRI_GROUP
#define RT_TRI_GROUP 4
#endif
static_assert(RT_TRI_GROUP == kTriChunk, "the triangle loop steps by one chunk of tri_chunks");

enum QueryKind : int { Q_TRACE = 0, Q_SHADOW = 1, Q_PHOTON = 2 };
enum LaneState : int { S_IDLE = 0, S_QUERY = 1, S_WAIT = 2 /* a Julia job of this lane is with the marchers */, S_RESOLVE = 4,
                       S_PARK = 8 /* trace_julia_kernel: the answer to this shadow query comes from a helper context */ };

constexpr int kFrameFields = 11;
constexpr unsigned kNoTriangle = 0xffffffffu;

struct Counters {
    unsigned long long v[C_COUNT];
};

// The tables the shading code reads per lane (HBM, or their LDS copy for small scenes).
struct ShadeTables {
    const DevMaterial* materials;
    const DevLight* lights;
    const DevTexture* textures;
    const uint32_t* object_material;
    const int32_t* object_photon_slot;
    const DevTriShade* tri_shade;
};
RT_DEV ShadeTables tables_of(const DevScene& sc, const unsigned char* base) {
    // base == sc.shade_blob leaves the pointers where they are; an LDS base rebases them
    const unsigned char* g = sc.shade_blob;
    ShadeTables t;
    t.materials = reinterpret_cast<const DevMaterial*>(base + (reinterpret_cast<const unsigned char*>(sc.materials) - g));
    t.lights = reinterpret_cast<const DevLight*>(base + (reinterpret_cast<const unsigned char*>(sc.lights) - g));
    t.textures = reinterpret_cast<const DevTexture*>(base + (reinterpret_cast<const unsigned char*>(sc.textures) - g));
    t.object_material = reinterpret_cast<const uint32_t*>(base + (reinterpret_cast<const unsigned char*>(sc.object_material) - g));
    t.object_photon_slot = reinterpret_cast<const int32_t*>(base + (reinterpret_cast<const unsigned char*>(sc.object_photon_slot) - g));
    t.tri_shade = reinterpret_cast<const DevTriShade*>(base + (reinterpret_cast<const unsigned char*>(sc.tri_shade) - g));
    return t;
}

// Everything one lane carries between phases.
struct Lane {
    int state;
    // current query
    V3 qo, qd;
    float qmax;
    int qkind;
    // flags are kept as 32-bit lane values (VGPRs): as bools they become SGPR lane masks that the
    // object loop has to re-merge with three scalar instructions each per iteration
    unsigned qactive;    // still testing objects (0 once a shadow ray is blocked)
    unsigned blocked;
    // best accepted hit of the current query
    unsigned found;
    HitRec best;
    unsigned best_obj;
    float best_cos;
    // a winning triangle is kept as (triangle, u, v, distance) until the query is over; its full hit
    // record is built once by finish_best_triangle
    unsigned best_tri;
    float best_u, best_v;
    // Julia march
    unsigned jidx, jstep;
    V3 jp, jd;
    // current node of the ray tree
    V3 n_point, n_normal, n_dir;
    V2 n_ph;
    float n_cos;
    unsigned n_obj;
    float acc[3], matcol[3];
    unsigned li;
    // sample
    unsigned sp;             // frames on the stack == depth of the current node
    unsigned resident;       // TopInLds: lowest frame held in LDS | what the frames below it hold in HBM
    float wavelength;
    int hue;                 // -1 without dispersion
    unsigned long long out_slot;
    unsigned hot;            // trace_julia_kernel: a march of this sample has been long (DevJob::hot_steps)
};

template <bool STATS>
RT_DEV void bump(Counters& n, int slot, unsigned long long by = 1) {
    if (STATS) n.v[slot] += by;
}

// Developer build (-DRT_PHASE_TIMING, tools/phase_split.py): section clocks of the Julia kernel's waves without the per-lane
// counters of the STATS build (which double its time and so distort the shares): a few wave-uniform accumulators,
// added to the launch's counter words when the wave ends.
#ifdef RT_PHASE_TIMING
constexpr bool kPhaseTiming = true;
#else
constexpr bool kPhaseTiming = false;
#endif
struct WaveClocks {
    static constexpr int kSlots = 12;
    unsigned long long acc[kSlots];
    int slot_of[kSlots];
    long long mark;
    RT_DEV void begin(const int* slots, int n_slots) {
        for (int i = 0; i < kSlots; i++) {
            acc[i] = 0ull;
            slot_of[i] = i < n_slots ? slots[i] : -1;
        }
        mark = static_cast<long long>(__builtin_readcyclecounter());
    }
    RT_DEV void charge(int i) {
        const long long now = static_cast<long long>(__builtin_readcyclecounter());
        acc[i] += static_cast<unsigned long long>(now - mark);
        mark = now;
    }
    RT_DEV void add(int i, unsigned long long by) { acc[i] += by; }
    RT_DEV void flush(unsigned long long* out) const {
        if ((threadIdx.x & 63) != 0) return;
        for (int i = 0; i < kSlots; i++)
            if (slot_of[i] >= 0 && acc[i] != 0ull) atomicAdd(&out[slot_of[i]], acc[i]);
    }
};

// A candidate hit of scene object `obj` reaches the per-query acceptance rule
// (src/raytracer.cpp:128-141 for Q_TRACE, :222-226 for Q_SHADOW, src/photonmap.cpp:68-74 for Q_PHOTON).
// Where the record of an accepted hit goes.  Kept in the lane's registers it stays live over the whole object loop;
// both render kernels have an LDS home for it instead (word offsets and stride in floats):
//   trace_kernel        the lane's parking column: the slots of n_point, n_normal, n_dir.xy and n_ph (park_node), which
//                       are dead while a Q_TRACE query is in flight (a shadow query never records a hit) and are
//                       exactly where the resolve phase wants the winner's record;
//   trace_julia_kernel  the union words of the ray's context record, where ctx_store would put it anyway.
// base == nullptr keeps the record in L.best (photon pass, Julia hits judged by the shader wave itself).
struct HitSink {
    float* base;
    unsigned stride;
    unsigned char point, normal, uv, phx, phy;
};
constexpr HitSink kHitInRegisters = {nullptr, 0u, 0, 0, 0, 0, 0};

RT_DEV void accept_hit(Lane& L, const HitRec& h, unsigned obj, float transmittance, float refr_norm, const HitSink& sink = kHitInRegisters) {
    if (L.qkind == Q_SHADOW) {
        if (dot(L.qd, h.normal) < 0.0f) {   // only front faces cast shadows
            L.blocked = 1u;
            L.qactive = 0u;
        }
        return;
    }
    float cosv = 0.0f;
    if (L.qkind == Q_TRACE) {
        cosv = clampf(dot(L.qd, h.normal), -1.0f, 1.0f);
        if (cosv >= 0.0f && (transmittance == 0.0f || refr_norm == 0.0f)) return;   // unseen back face
    }
    L.found = 1u;
    if (sink.base != nullptr) {
        float* const b = sink.base;
        const unsigned st = sink.stride;
        b[(sink.point + 0u) * st] = h.point.x;
        b[(sink.point + 1u) * st] = h.point.y;
        b[(sink.point + 2u) * st] = h.point.z;
        b[(sink.normal + 0u) * st] = h.normal.x;
        b[(sink.normal + 1u) * st] = h.normal.y;
        b[(sink.normal + 2u) * st] = h.normal.z;
        b[(sink.uv + 0u) * st] = h.uv.x;
        b[(sink.uv + 1u) * st] = h.uv.y;
        b[sink.phx * st] = h.ph.x;
        b[sink.phy * st] = h.ph.y;
        L.best.dist = h.dist;
    } else {
        L.best = h;
    }
    L.best_obj = obj;
    L.best_cos = cosv;
    L.best_tri = kNoTriangle;
    L.qmax = h.dist;
}

// The same rule for a triangle candidate (u, v, dist) whose interpolated normal sum is nv; only the
// facing decision is taken here, the hit record is deferred.
RT_DEV void accept_triangle(Lane& L, unsigned obj, unsigned tri, float u, float v, float dist, V3 nv, float transmittance, float refr_norm) {
    if (L.qkind == Q_SHADOW) {
        if (facing(L.qd, nv) < 0.0f) {
            L.blocked = 1u;
            L.qactive = 0u;
        }
        return;
    }
    if (L.qkind == Q_TRACE && (transmittance == 0.0f || refr_norm == 0.0f) && facing(L.qd, nv) >= 0.0f) return;
    L.found = 1u;
    L.best_obj = obj;
    L.best_tri = tri;
    L.best_u = u;
    L.best_v = v;
    L.best.dist = dist;
    L.qmax = dist;
}

// Completes a deferred triangle winner (rest of Triangle::intersect and the cosine of
// src/raytracer.cpp:131).
RT_DEV void finish_best_triangle(const ShadeTables& T, Lane& L) {
    if (L.found != 0u && L.best_tri != kNoTriangle) {
        tri_finish(T.tri_shade[L.best_tri], L.qo, L.qd, L.best_u, L.best_v, L.best.dist, L.best);
        L.best_cos = L.qkind == Q_TRACE ? clampf(dot(L.qd, L.best.normal), -1.0f, 1.0f) : 0.0f;
        L.best_tri = kNoTriangle;
    }
}

// Per-lane traversal of one run's BVH (built on the GPU by rt_bvh.hip).  Returns the winner the
// sequential scan of the run would return: nearest accepted hit, equal distances to the later object.
// One 64-byte record per step holds the boxes of both children: the nearer one is entered first, the other is
// pushed only if the ray meets it too.  Stack entries are child words (leaf bit = a triangle to test).
// (A "while-while" form, lanes holding a triangle waiting for the wave before the test, measured slower:
// 6.1 ms against 4.6 ms on the 104k-triangle mesh; the walk is latency-bound and the mixed step overlaps more loads.)
// 1/d per axis for the culling slab tests only (never for a value that reaches the picture): the hardware
// reciprocal (1 ulp) instead of three IEEE divisions; rcp(+-0) = +-inf like the division.
RT_DEV V3 cull_inverse(V3 d) { return {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z)}; }

struct SlabHit {
    bool overlap;
    float tnear;
};
RT_DEV float slab_slack(float4 lo, float4 hi) { return 1e-4f * ((hi.x - lo.x) + (hi.y - lo.y) + (hi.z - lo.z)); }
RT_DEV SlabHit slab_test(float4 lo, float4 hi, V3 o, V3 inv, float qmax, float t_slack) {
    // fminf/fmaxf drop the NaNs of 0*inf so that an axis-parallel ray lying in a face plane is kept; the far side
    // is widened by 8 ulp (robust traversal with the 1-ulp reciprocal of cull_inverse)
    const float x0 = (lo.x - o.x) * inv.x, x1 = (hi.x - o.x) * inv.x;
    const float y0 = (lo.y - o.y) * inv.y, y1 = (hi.y - o.y) * inv.y;
    const float z0 = (lo.z - o.z) * inv.z, z1 = (hi.z - o.z) * inv.z;
    const float tnear = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
    const float tfar = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    // The distance the triangle test computes (src/objects.cpp:56-81) and the slab distances computed here
    // are different roundings of the same number: a box is flat when its triangles lie in an axis plane,
    // and the test may then put the hit a few ulp in front of the box.  The scan accepts t == maximum (an
    // exact tie goes to the later object), so a box is skipped only when it lies beyond the maximum by
    // more than that noise can be (1e-4 of the distance and of the box: ~10^3 ulp, no measurable cost).
    // (t_slack = slab_slack(lo, hi); the BVH records carry it in their spare words)
    SlabHit h;
    h.overlap = !(tnear > tfar * 1.000001f) && !(tfar < -t_slack) && !(tnear * 0.9999f - t_slack > qmax);
    h.tnear = tnear;
    return h;
}

// The first entries of a lane's traversal stack live in LDS ([entry][lane of the workgroup]): a pop is a dependent
// load in front of the next record's load, and LDS answers in ~64 cycles where scratch goes through the vector cache.
struct BvhStackLds {
    unsigned* lane_base;     // this lane's column
    unsigned entries;        // 0: everything in scratch (trace_julia_kernel, whose LDS is its ray pool)
    unsigned stride;         // lanes in the workgroup
    const DevBvhPair* nodes; // the records: sc.bvh_nodes, or the workgroup's LDS copy of a small scene's (the walk is a chain
                             // of dependent loads: LDS answers in half the time of a vector-L1 hit)
};

template <bool STATS>
RT_DEV void query_bvh(const DevScene& sc, Lane& L, unsigned act, unsigned bvh_root, unsigned first_object, unsigned first_prim, const BvhStackLds& fast, Counters& n) {
    if (act == 0u) return;
    const V3 inv = cull_inverse(L.qd);
    unsigned deep[64];
    unsigned sp = 0;
    unsigned cur = 0;
    for (;;) {
        if ((cur & kBvhLeafBit) != 0u) {
            bump<STATS>(n, C_TRIANGLE_TESTS);
            const unsigned tri = cur & ~kBvhLeafBit;
            const TriGeom g = load_tri(sc.tri_test + tri);
            const TriStage1 s1 = tri_stage1(g, L.qo, L.qd);
            float u, v, t;
            if (s1.maybe && tri_stage23(g, s1, L.qd, L.qmax, u, v, t)) {
                const unsigned obj = first_object + (tri - first_prim);
                // order-independent form of the scan's tie rule (t <= running maximum, later wins)
                if (L.qkind == Q_SHADOW || t < L.qmax || L.found == 0u || obj > L.best_obj) {
                    const V3 nv = tri_normal_sum(sc.tri_shade + tri, u, v);
                    const DevMaterial& m = sc.materials[sc.object_material[obj]];
                    accept_triangle(L, obj, tri, u, v, t, nv, m.transmittance, m.refr_norm);
                    if (L.qactive == 0u) break;
                }
            }
        } else {
            bump<STATS>(n, C_BVH_NODE_VISITS, 2);
            // (one flat load per 16 bytes: typed loads — ds_read_b128 from the workgroup's copy, global_load otherwise — behind a
            // uniform branch measured no faster on C2 and 6 % slower on the 104k-triangle mesh)
            const float4* raw = reinterpret_cast<const float4*>(fast.nodes + bvh_root + cur);
            const float4 lo0 = raw[0], hi0 = raw[1], lo1 = raw[2], hi1 = raw[3];
            const SlabHit h0 = slab_test(lo0, hi0, L.qo, inv, L.qmax, lo1.w), h1 = slab_test(lo1, hi1, L.qo, inv, L.qmax, hi1.w);
            const unsigned c0 = __float_as_uint(lo0.w), c1 = __float_as_uint(hi0.w);
            if (h0.overlap && h1.overlap) {
                const bool first0 = h0.tnear <= h1.tnear;
                const unsigned later = first0 ? c1 : c0;
                if (sp < fast.entries) fast.lane_base[sp * fast.stride] = later;
                else deep[sp - fast.entries] = later;
                sp++;
                cur = first0 ? c0 : c1;
                continue;
            }
            if (h0.overlap || h1.overlap) {
                cur = h0.overlap ? c0 : c1;
                continue;
            }
        }
        if (sp == 0u) break;
        sp--;
        cur = sp < fast.entries ? fast.lane_base[sp * fast.stride] : deep[sp - fast.entries];
    }
}

// The wave-uniform object loop: every lane with L.qactive tests its ray against every non-Julia
// object in scene order.  `runs`, the geometry tables and the material flags are read at uniform
// addresses.
template <bool STATS>
RT_DEV void query_objects(const DevScene& sc, Lane& L, bool in_phase, const BvhStackLds& bvh_stack, Counters& n, const HitSink& sink = kHitInRegisters) {
    unsigned act = in_phase ? L.qactive : 0u;   // 32-bit lane value on purpose (see Lane)
    const auto runs = to_constant(sc.runs);
    const auto tris = to_constant(sc.tri_test);
    const auto shades = to_constant(sc.tri_shade);
    const auto spheres = to_constant(sc.spheres);
    const auto obj_mat = to_constant(sc.object_material);
    const auto mats = to_constant(sc.materials);
    const auto chunks = to_constant(sc.tri_chunks);
    const auto sbounds = to_constant(sc.sphere_bounds);
    const float oo = dot(L.qo, L.qo);
    const V3 inv = cull_inverse(L.qd);
    for (unsigned r = 0; r < sc.n_runs; r++) {
        const int kind = runs[r].kind;
        const unsigned first_object = runs[r].first_object, first_prim = runs[r].first_prim, count = runs[r].count;
        const long long t_run = STATS ? clock64() : 0;
        if (kind == RUN_TRIANGLES) {
            // RT_TRI_GROUP triangles per trip: their records arrive in SGPRs together and their
            // division-free first stages are independent instruction streams; the (rare) survivors
            // are finished one after the other in scene order against the running maximum.
            for (unsigned i = 0; i < count; i += RT_TRI_GROUP) {
                if (!__any(act != 0u)) break;
                {
                    // skip the chunk when no open query of the wave can touch its bounds (slab test as in
                    // query_bvh: NaNs of 0*inf dropped by fminf/fmaxf, far side widened)
                    const auto cb = chunks + runs[r].first_chunk + i / RT_TRI_GROUP;
                    const float x0 = (cb->lo[0] - L.qo.x) * inv.x, x1 = (cb->hi[0] - L.qo.x) * inv.x;
                    const float y0 = (cb->lo[1] - L.qo.y) * inv.y, y1 = (cb->hi[1] - L.qo.y) * inv.y;
                    const float z0 = (cb->lo[2] - L.qo.z) * inv.z, z1 = (cb->hi[2] - L.qo.z) * inv.z;
                    const float tnear = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
                    const float tfar = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
                    // (t_slack: see query_bvh — the triangle test's distance and the slab's are different roundings)
                    const float t_slack = cb->slack;
                    const bool overlap = !(tnear > tfar * 1.000001f) && !(tfar < -t_slack) && !(tnear * 0.9999f - t_slack > L.qmax);
                    if (STATS && (threadIdx.x & 63) == 0) n.v[D_CHUNK_TESTS]++;
                    if (!__any(act != 0u && overlap)) {
                        if (act != 0u) bump<STATS>(n, C_TRIANGLE_TESTS, count - i < RT_TRI_GROUP ? count - i : RT_TRI_GROUP);
                        continue;
                    }
                    if (STATS) {
                        if ((threadIdx.x & 63) == 0) n.v[D_CHUNK_PASSED]++;
                        if (act != 0u && overlap) n.v[D_CHUNK_LANES]++;
                    }
                }
                TriGeom g[RT_TRI_GROUP];
                TriStage1 s1[RT_TRI_GROUP];
#pragma unroll
                for (int k = 0; k < RT_TRI_GROUP; k++) {
                    const unsigned idx = i + k < count ? i + k : count - 1;
                    g[k] = load_tri(tris + first_prim + idx);
                }
                bool any_maybe = false;
#pragma unroll
                for (int k = 0; k < RT_TRI_GROUP; k++) {
                    s1[k] = tri_stage1(g[k], L.qo, L.qd);
                    any_maybe = any_maybe || (s1[k].maybe && i + k < count);
                }
                if (STATS) {
                    // the oracle's count: tests made while the query is still open (a shadow ray stops
                    // at its first blocker), in order
                    if (act != 0u && !any_maybe) bump<STATS>(n, C_TRIANGLE_TESTS, count - i < RT_TRI_GROUP ? count - i : RT_TRI_GROUP);
                }
                if (act != 0u && any_maybe) {
#pragma unroll
                    for (int k = 0; k < RT_TRI_GROUP; k++) {
                        if (i + k < count && act != 0u) {
                            bump<STATS>(n, C_TRIANGLE_TESTS);
                            float u, v, t;
                            if (s1[k].maybe && tri_stage23(g[k], s1[k], L.qd, L.qmax, u, v, t)) {
                                const V3 nv = tri_normal_sum(shades + first_prim + i + k, u, v);
                                const unsigned m = obj_mat[first_object + i + k];
                                accept_triangle(L, first_object + i + k, first_prim + i + k, u, v, t, nv, mats[m].transmittance, mats[m].refr_norm);
                                act = L.qactive;
                            }
                        }
                    }
                }
            }
        } else if (kind == RUN_TRIANGLE_BVH) {
            query_bvh<STATS>(sc, L, act, runs[r].bvh_root, first_object, first_prim, bvh_stack, n);
            act = in_phase ? L.qactive : 0u;
        } else if (kind == RUN_SPHERES) {
            for (unsigned i = 0; i < count; i++) {
                if (!__any(act != 0u)) break;
                {
                    // skip the sphere when every open query of the wave provably misses it: the ray passes
                    // outside the bounding sphere, starts outside and points away, or the sphere lies
                    // beyond the query's maximum (slack: DevSphereBound)
                    const auto sb = sbounds + first_prim + i;
                    const V3 vc = v3(sb->center[0], sb->center[1], sb->center[2]) - L.qo;
                    const float cc = dot(vc, vc), bb = dot(vc, L.qd);
                    const float slack = sb->slack_scale * (oo + sb->magnitude2);
                    const float lim = sb->radius2 + slack;
                    const bool miss = (cc - bb * bb > lim) || (cc > lim && (bb < 0.0f || bb - sb->radius > L.qmax * 1.001f + slack));
                    if (STATS && (threadIdx.x & 63) == 0) n.v[D_SPHERE_CULLS]++;
                    if (!__any(act != 0u && !miss)) {
                        if (act != 0u) bump<STATS>(n, C_SPHERE_TESTS);
                        continue;
                    }
                    if (STATS && (threadIdx.x & 63) == 0) n.v[D_SPHERE_PASSED]++;
                }
                if (act != 0u) {
                    bump<STATS>(n, C_SPHERE_TESTS);
                    HitRec h;
                    if (sphere_intersect(spheres + first_prim + i, L.qo, L.qd, L.qmax, h)) {
                        const unsigned m = obj_mat[first_object + i];
                        accept_hit(L, h, first_object + i, mats[m].transmittance, mats[m].refr_norm, sink);
                        act = L.qactive;
                    }
                }
            }
        }
        if (STATS && (threadIdx.x & 63) == 0) n.v[kind != RUN_SPHERES ? D_TRI_CYCLES : D_SPHERE_CYCLES] += static_cast<unsigned long long>(clock64() - t_run);
    }
}

// Julia hits are judged after the uniform loop; the rule below is the order-independent form of the
// reference's sequential one (nearest wins, equal distances go to the later scene object).
//
// With one exception, which is not order-independent in the reference either.  Sphere::intersect does not compare its
// distance with the maximum: it carries the maximum into object space — the LENGTH of o - w2o(o_w + d_w * max) — and
// compares the ray parameter with that (src/objects.cpp:33-38).  Far from the origin o_w + d_w * max lands on a grid
// point (at 5 000 units the grid is 0.0005 wide), the length is that noisy, and a sphere tested AFTER a set that has
// just been hit nearer is accepted all the same, or rejected at the very same distance (tools/fuzz_parity.py, seed 5477:
// two pixels).  So when the winner so far is a sphere that comes after this set in the scene, the reference's own question
// is put: does that sphere still report a hit with the set's distance as its maximum?  (The loop tested it with a
// maximum that did not know of the set's hit.)
RT_DEV const DevSphere* sphere_of_object(const DevScene& sc, unsigned obj) {
    for (unsigned r = 0; r < sc.n_runs; r++) {
        const DevRun& run = sc.runs[r];
        if (obj >= run.first_object && obj - run.first_object < run.count) return run.kind == RUN_SPHERES ? sc.spheres + run.first_prim + (obj - run.first_object) : nullptr;
    }
    return nullptr;
}
// (out of line: rare, and inlined it costs the shader waves of trace_julia_kernel registers they spill)
__device__ __attribute__((noinline)) bool sphere_hit_within(const DevSphere* sphere, V3 o, V3 d, float maxd) {
    HitRec again;
    return sphere_intersect(sphere, o, d, maxd, again);
}
RT_DEV void accept_julia_hit(const DevScene& sc, Lane& L, const HitRec& h, const DevJulia& j, const DevMaterial& m) {
    if (L.qkind != Q_SHADOW) {
        bool nearer = h.dist < L.qmax || (h.dist == L.qmax && (!L.found || j.object > L.best_obj));
        if (L.found != 0u && L.best_obj > j.object && L.best_tri == kNoTriangle) {
            // (an unseen back face of the set changes nothing in the reference's loop: accept_hit below drops it either way)
            if (const DevSphere* sphere = sphere_of_object(sc, L.best_obj)) nearer = !sphere_hit_within(sphere, L.qo, L.qd, h.dist);
        }
        if (!nearer) return;
    }
    accept_hit(L, h, j.object, m.transmittance, m.refr_norm);
}

RT_DEV void begin_query(Lane& L, V3 o, V3 d, float maxd, int kind) {
    L.qo = o;
    L.qd = d;
    L.qmax = maxd;
    L.qkind = kind;
    L.qactive = 1u;
    L.blocked = 0u;
    L.found = 0u;
    L.best_tri = kNoTriangle;
    L.state = S_QUERY;
}

// ---- frame stack in HBM ---------------------------------------------------------------------------
struct FrameRef {
    float* base;
    unsigned long long stride;   // lanes in the grid
    RT_DEV float& at(unsigned frame, int field) const { return base[(static_cast<unsigned long long>(frame) * kFrameFields + field) * stride]; }
};

// Where a lane's DFS frames live.  resolve_lane only sees this interface: frame f of the lane, field c.
//   TopInLds   (trace_kernel, photon-free lockstep lanes): the newest frame is kept in the workgroup's LDS array
//              [field][lane] (in registers it was the first thing the compiler spilled to scratch); HBM holds the
//              frames below it, [frame][field][grid lane] so that a wave's accesses coalesce.
//   FramesInHbm (trace_julia_kernel's ray pool): a context is handled by whichever wave pops it, so every frame is a
//              48-byte record of its own in HBM, [context][frame][field]: a lane touches one or two cache lines per push/pop.
struct TopInLds {
    float* top;              // this lane's column of the first LDS slot: field c at top[c * top_stride]
    float* top2;             // ... of the second slot (unused with one slot)
    unsigned top_stride;
    unsigned slot_mask;      // slots - 1: 0 or 1
    FrameRef below;
    unsigned* resident;      // low bits: the lane's lowest frame that is in LDS (frames lowest .. lowest + slots - 1 are); above them the frame codes, see below
    RT_DEV float* slot(unsigned f) const { return (f & slot_mask) != 0u ? top2 : top; }
    RT_DEV float get(unsigned f, int c) const { return slot(f)[c * top_stride]; }
    RT_DEV void set(unsigned f, int c, float v) const { slot(f)[c * top_stride] = v; }
    // With two slots a ray tree that goes up and down between two neighbouring depths (the common case near its
    // leaves) touches no HBM at all; a frame moves out only when the tree goes two deeper than the oldest resident
    // frame and comes back only when the tree returns below it.
    //
    // What moves is what the frame still needs.  The pending reflection ray (fields 4-9) is dead unless the child in
    // flight is the refraction with the reflection still to come (flags == 2), and the flags themselves (0, 1 or 2) fit
    // two bits: the word *resident keeps, above the frame number, a two-bit code for each of the first kCodedFrames
    // frames — 3 = pushed and never moved out, else the flags it had when it last moved out.  A frame that moves out
    // a second time (flags 1 after 2: its reflection child has gone deeper) finds its Fresnel weight in HBM already.
    // So a node with two children costs 10 + 3 words out and 10 + 4 back instead of 11 each way, twice (C3: 4.55 GB
    // of traffic per frame with whole records).  Frames from kCodedFrames up move whole, flags and all.
    static constexpr unsigned kFrameNumberMask = 127u, kCodeShift = 7u, kCodedFrames = 12u;
    RT_DEV void before_push(unsigned f) const {      // frame f becomes the top
        unsigned r = *resident;
        const unsigned lowest = r & kFrameNumberMask;
        if (f > lowest + slot_mask) {
            const unsigned out = lowest;
            const float* from = slot(out);
            const unsigned flags = __float_as_uint(from[10 * top_stride]);
            const bool coded = out < kCodedFrames;
            const unsigned shift = kCodeShift + 2u * (coded ? out : 0u);
            const unsigned before = coded ? (r >> shift) & 3u : 3u;
#pragma unroll
            for (int c = 0; c < 3; c++) below.at(out, c) = from[c * top_stride];
            if (!(before == 2u && flags == 1u)) below.at(out, 3) = from[3 * top_stride];
            if (!coded || flags == 2u) {
#pragma unroll
                for (int c = 4; c < 10; c++) below.at(out, c) = from[c * top_stride];
            }
            if (coded) r = (r & ~(3u << shift)) | (flags << shift);
            else below.at(out, 10) = from[10 * top_stride];
            r += 1u;
        }
        if (f < kCodedFrames) r |= 3u << (kCodeShift + 2u * f);
        *resident = r;
    }
    RT_DEV void after_pop(unsigned f) const {        // frame f is gone: frame f-1 is the top again
        const unsigned r = *resident;
        if (f > 0 && f - 1 < (r & kFrameNumberMask)) {
            const unsigned k = f - 1;
            float* to = slot(k);
            const bool coded = k < kCodedFrames;
            const unsigned code = coded ? (r >> (kCodeShift + 2u * k)) & 3u : 2u;
#pragma unroll
            for (int c = 0; c < 4; c++) to[c * top_stride] = below.at(k, c);
            if (code == 2u) {
#pragma unroll
                for (int c = 4; c < 10; c++) to[c * top_stride] = below.at(k, c);
            }
            to[10 * top_stride] = coded ? __uint_as_float(code) : below.at(k, 10);
            *resident = (r & ~kFrameNumberMask) | k;
        }
    }
};
constexpr int kFrameRecordWords = 12;
struct FramesInHbm {
    float* rec;              // this context's records
    RT_DEV float get(unsigned f, int c) const { return rec[f * kFrameRecordWords + c]; }
    RT_DEV void set(unsigned f, int c, float v) const { rec[f * kFrameRecordWords + c] = v; }
    RT_DEV void before_push(unsigned) const {}
    RT_DEV void after_pop(unsigned) const {}
};

// ---- sample queue ---------------------------------------------------------------------------------
// Lanes with `need` set get consecutive sample ids.  Most ids are owned outright: wave w of W owns the
// 64-id blocks w, w + W, w + 2W ... (interleaved: a contiguous range per wave would hand whole image
// regions, cheap or expensive, to single waves) and touches no shared state while they last; the rest
// of the id space is handed out through one global counter to even out the end of the launch.
// That counter is a device-scope atomic on a single address: it completes memory-side at about one per
// 20 ns for the whole chip (measured: the photon pass took 8.9 ms with 360 k reservations and 3.4 ms
// with 51 k), so reservations are also made in chunks where the units are short and uniform.  The
// trace kernels reserve exactly what the idle lanes need: ids parked in a slow wave cost more at the
// end of a frame than the atomics do.
struct WaveIds {
    unsigned long long next, end;   // wave-uniform: the ids in hand
    unsigned wave, n_waves;         // this wave's place in the interleaved static part
    unsigned static_done, static_blocks;
    bool dry;                       // the dynamic part is exhausted
};
RT_DEV WaveIds wave_ids_begin(unsigned long long gid, unsigned n_waves, unsigned static_blocks) {
    return {0ull, 0ull, static_cast<unsigned>(gid >> 6), n_waves, 0u, static_blocks, false};
}
RT_DEV bool fetch_ids(unsigned long long* head, unsigned long long dyn_base, unsigned long long limit, unsigned chunk, WaveIds& w, bool need, unsigned long long& id) {
    const unsigned long long mask = __ballot(need);
    if (mask == 0ull) return false;
    const int lane = threadIdx.x & 63;
    const unsigned rank = __popcll(mask & ((1ull << lane) - 1ull));
    const unsigned cnt = __popcll(mask);
    unsigned served = 0;
    bool got = false;
    while (served < cnt) {
        while (w.next >= w.end && w.static_done < w.static_blocks) {
            const unsigned long long b = (static_cast<unsigned long long>(w.static_done) * w.n_waves + w.wave) * 64ull;
            w.static_done++;
            if (b < limit) {
                w.next = b;
                w.end = b + 64ull < limit ? b + 64ull : limit;
            }
        }
        if (w.next >= w.end) {
            if (w.dry) break;   // the queue ran dry earlier
            const unsigned long long want = chunk > cnt - served ? chunk : cnt - served;
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(head, want) + dyn_base;
            base = __shfl(base, 0);
            w.next = base < limit ? base : limit;
            w.end = base + want < limit ? base + want : limit;
            if (w.next >= w.end) {
                w.dry = true;
                break;
            }
        }
        const unsigned long long avail = w.end - w.next;
        const unsigned take = avail < cnt - served ? static_cast<unsigned>(avail) : cnt - served;
        if (need && rank >= served && rank < served + take) {
            id = w.next + (rank - served);
            got = true;
        }
        w.next += take;
        served += take;
    }
    return got;
}

// ---- long chains first (trace_julia_kernel, DevJob::sched) ------------------------------------------------------------
// A sample is a chain of dependent queries, and a Julia march a chain of dependent steps of ~4 us each: on C4 the
// longest sample makes 5 100 steps in 24 marches, 21 ms of latency whatever the machine's throughput, and 450 of the
// frame's 9 000 tiles hold a sample of more than 2 000 steps.  Handed out in image order, the last of them start when
// the frame is nearly done, and the frame ends a chain's length later (measured: 30.5 ms at 960 x 600, of which 18.5 ms
// is throughput; a frame of a sixty-fourth of the size takes 13.2 ms).  So, where a pixel has several samples:
//   * ids are handed out in blocks of DevJob::group_tiles tiles' worth of one sample, numbered sample-major — the first
//     sample of every group of tiles, then the second ... — one at a time through the launch's counter;
//   * a marcher wave that takes step number DevJob::hot_steps of a march of a FIRST sample declares the sample's tile
//     hot: the tile's other blocks go on the hot list, and a wave that needs a block looks there first.  Cheap tiles
//     (background, the plane) thus queue behind the tiles whose samples are long, which start within a millisecond or
//     two of the first sample that gave them away.
// A block with sample number >= 1 is served by whoever sets its claim bit first, through either way; results do not
// depend on the order (every sample has its own slot in the sample buffer).
RT_DEV unsigned sched_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RT_DEV bool sched_claim(const DevJob& job, unsigned block) {
    uint32_t* const bits = job.sched + kSchedHeaderWords + job.n_groups;
    const unsigned bit = 1u << (block & 31u);
    return (atomicOr(bits + (block >> 5), bit) & bit) == 0u;
}
// Hot list of level k (a march reached hot_steps << k steps): the longer the march, the longer (as a rule) the sample,
// and the sooner the tile's other samples should start; a wave looks at the highest level first.
RT_DEV uint32_t* sched_list(const DevJob& job, unsigned level) {
    const unsigned n_blocks = job.n_groups * job.samples_per_pixel;
    return job.sched + kSchedHeaderWords + job.n_groups + (n_blocks + 31u) / 32u + level * job.n_groups * (job.samples_per_pixel - 1u);
}
// One lane: the next block for its wave, or ~0 when every block has been handed out.
RT_DEV unsigned sched_next_block(const DevJob& job) {
    const unsigned n_blocks = job.n_groups * job.samples_per_pixel;
    for (;;) {
        // first samples before everything else: they are what finds the long chains (and they hold some themselves)
        if (__hip_atomic_load(job.queue_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < job.n_groups) {
            const unsigned long long q = atomicAdd(job.queue_head, 1ull);
            if (q >= n_blocks) return ~0u;
            const unsigned b = static_cast<unsigned>(q);
            if (b < job.n_groups || sched_claim(job, b)) return b;
        }
        for (int level = static_cast<int>(kSchedLevels) - 1; level >= 0; level--) {
            uint32_t* const ctl = job.sched + 2 * level;          // [0] entries pushed, [1] entries claimed
            const uint32_t* const list = sched_list(job, static_cast<unsigned>(level));
            unsigned head = sched_load(ctl + 1);
            const unsigned tail = sched_load(ctl + 0);
            while (head < tail) {
                const unsigned seen = atomicCAS(ctl + 1, head, head + 1u);
                if (seen != head) {
                    head = seen;
                    continue;
                }
                // entry `head` is this wave's; its writer reserved it a moment ago (if it is late the block stays with the queue)
                unsigned b = ~0u;
                for (int spin = 0; spin < 4096 && (b = sched_load(list + head)) == ~0u; spin++) __builtin_amdgcn_s_sleep(1);
                if (b != ~0u && sched_claim(job, b)) return b;
                head++;
            }
        }
        const unsigned long long q = atomicAdd(job.queue_head, 1ull);
        if (q >= n_blocks) return ~0u;
        const unsigned b = static_cast<unsigned>(q);
        if (b < job.n_groups || sched_claim(job, b)) return b;      // first samples have no other way
    }
}
// fetch_ids for launches with DevJob::sched.  The block in hand belongs to the WORKGROUP (`hand`, a word of its LDS:
// next id << 24 | ids left, kHandBusy while one wave is away fetching the next block, kHandDry once there is none): a
// context dies only when no id is left anywhere, whichever shader wave it asks — with a hand per wave the contexts could
// all die through the wave whose hand is empty while the other wave's still holds ids.
constexpr unsigned long long kHandBusy = ~0ull, kHandDry = ~0ull - 1ull;
constexpr unsigned long long kStuckTicksEarly = 100000000ull;   // = kStuckTicks (1 s of the 100 MHz clock)
RT_DEV bool fetch_ids_scheduled(const DevJob& job, unsigned long long* hand, unsigned* abort_flag, bool need, unsigned long long& id) {
    const unsigned long long mask = __ballot(need);
    if (mask == 0ull) return false;
    const int lane = threadIdx.x & 63;
    const unsigned rank = __popcll(mask & ((1ull << lane) - 1ull));
    const unsigned cnt = __popcll(mask);
    unsigned served = 0;
    bool got = false;
    while (served < cnt) {
        unsigned long long first = 0;
        unsigned take = 0;
        if (lane == 0) {
            unsigned long long t0 = 0;
            for (;;) {
                unsigned long long h = __hip_atomic_load(hand, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (h == kHandDry) break;
                if (h == kHandBusy) {
                    // (the other wave is fetching: a few microseconds; bounded like every wait of this kernel)
                    __builtin_amdgcn_s_sleep(2);
                    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                    if (t0 == 0) t0 = now;
                    if (now - t0 > kStuckTicksEarly) {
                        __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        break;
                    }
                    continue;
                }
                const unsigned left = static_cast<unsigned>(h & 0xffffffull);
                if (left == 0u) {
                    if (!__hip_atomic_compare_exchange_strong(hand, &h, kHandBusy, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) continue;
                    const unsigned b = sched_next_block(job);
                    const unsigned long long per_block = 64ull * job.group_tiles;
                    __hip_atomic_store(hand, b == ~0u ? kHandDry : ((static_cast<unsigned long long>(b) * per_block) << 24) | per_block, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    continue;
                }
                const unsigned want = cnt - served;
                const unsigned n = left < want ? left : want;
                const unsigned long long next = h >> 24;
                if (__hip_atomic_compare_exchange_strong(hand, &h, ((next + n) << 24) | (left - n), __ATOMIC_ACQ_REL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                    first = next;
                    take = n;
                    break;
                }
            }
        }
        first = __shfl(first, 0);
        take = __shfl(take, 0);
        if (take == 0u) break;      // dry (or given up)
        if (need && rank >= served && rank < served + take) {
            id = first + (rank - served);
            got = true;
        }
        served += take;
    }
    return got;
}
// A march of the sample in out_slot has just taken step number DevJob::hot_steps << level (marcher waves).
RT_DEV void sched_mark_hot(const DevScene& sc, const DevJob& job, unsigned long long out_slot, unsigned level) {
    if (out_slot >= job.n_pixels) return;                        // not a first sample
    const unsigned pixel = static_cast<unsigned>(out_slot);
    const unsigned yc = pixel / sc.width, x = pixel % sc.width;
    const unsigned tile = ((yc >> 3) * job.tiles_x + (x >> 3)) / job.group_tiles;       // its group of tiles, that is
    if (atomicMax(job.sched + kSchedHeaderWords + tile, level + 1u) >= level + 1u) return;   // on this list, or a higher one, already
    const unsigned k = job.samples_per_pixel - 1u;
    uint32_t* const list = sched_list(job, level);
    const unsigned at = atomicAdd(job.sched + 2 * level, k);
    for (unsigned s = 1; s <= k; s++) __hip_atomic_store(list + at + (s - 1u), s * job.n_groups + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the level whose threshold `jstep` has just reached, or -1
RT_DEV int sched_level_reached(const DevJob& job, unsigned jstep) {
    if (job.hot_steps == 0u) return -1;
    for (unsigned k = 0; k < kSchedLevels; k++)
        if (jstep == job.hot_steps << k) return static_cast<int>(k);
    return -1;
}

RT_DEV unsigned hash_u32(unsigned x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

RT_DEV bool start_sample_at(const DevScene& sc, const DevJob& job, Lane& L, unsigned x, unsigned y, unsigned yc, unsigned s, unsigned hue, unsigned sub, unsigned sx, unsigned sy);
// Decodes a sample id and builds its camera ray (src/raytracer.cpp:56-93).  Returns false for the
// padding lanes of edge tiles.
RT_DEV bool start_sample(const DevScene& sc, const DevJob& job, unsigned long long id, Lane& L) {
    const unsigned p = static_cast<unsigned>(id & 63ull);
    const unsigned long long q = id >> 6;
    // blocks of 64 ids: tile-major (every sample of a tile, then the next tile), or sample-major under DevJob::sched
    unsigned s, tile;
    if (job.sched == nullptr && job.fast_decode != 0u) {
        // the launch's divisors as multiplications (rt_divide.h)
        tile = divmod_u32(static_cast<unsigned>(q), job.div_samples, s);
        unsigned tx, by;
        const unsigned ty = divmod_u32(tile, job.div_tiles_x, tx);
        const unsigned x = tx * 8 + (p & 7), yc = ty * 8 + (p >> 3);
        if (x >= sc.width || yc >= job.compact_rows) return false;
        const unsigned band = divmod_u32(yc, job.div_band_rows, by);
        const unsigned y = (job.first_band + band * job.band_stride) * job.band_rows + by;
        const unsigned hue = sc.dispersion ? s & 7u : 0u, sub = sc.dispersion ? s >> 3 : s;
        unsigned sx;
        const unsigned sy = divmod_u32(sub, job.div_supersampling, sx);
        return start_sample_at(sc, job, L, x, y, yc, s, hue, sub, sx, sy);
    }
    if (job.sched != nullptr) {
        // block = (sample, group of group_tiles consecutive tiles): sample-major
        const unsigned long long blk = q / job.group_tiles;
        s = static_cast<unsigned>(blk / job.n_groups);
        tile = static_cast<unsigned>(blk % job.n_groups) * job.group_tiles + static_cast<unsigned>(q % job.group_tiles);
        if (tile >= job.n_tiles) return false;     // padding of the last group
    } else {
        s = static_cast<unsigned>(q % job.samples_per_pixel);
        tile = static_cast<unsigned>(q / job.samples_per_pixel);
    }
    const unsigned x = (tile % job.tiles_x) * 8 + (p & 7), yc = (tile / job.tiles_x) * 8 + (p >> 3);
    if (x >= sc.width || yc >= job.compact_rows) return false;
    const unsigned y = (job.first_band + (yc / job.band_rows) * job.band_stride) * job.band_rows + yc % job.band_rows;
    const unsigned hues = sc.dispersion ? 8u : 1u;
    const unsigned hue = s % hues, sub = s / hues;
    const unsigned sx = sub % sc.supersampling, sy = sub / sc.supersampling;
    return start_sample_at(sc, job, L, x, y, yc, s, hue, sub, sx, sy);
}

// ... the camera ray of sample s (= hue, sub-pixel (sx, sy)) of pixel (x, y), row yc of the compact output
RT_DEV bool start_sample_at(const DevScene& sc, const DevJob& job, Lane& L, unsigned x, unsigned y, unsigned yc, unsigned s, unsigned hue, unsigned sub, unsigned sx, unsigned sy) {
    const float sdx = sc.sub_disp[sx], sdy = sc.sub_disp[sy];
    const V3 on_plane = {sc.ray_x[x] + sdx * sc.pixel_size[0], sc.ray_y[y] + sdy * sc.pixel_size[1], -1.0f};
    const V3 on_focus = xform_point(sc.cam, on_plane * sc.focus_distance);
    // lens jitter: the reference draws from a std::minstd_rand seeded by std::random_device per worker
    // (not reproducible); here the draw is a hash of (pixel, sub-pixel, seed).  With
    // lens_size == 0 the product is zero either way.
    const unsigned h1 = hash_u32((y * sc.width + x) * 64u + sub + job.dof_seed * 0x9e3779b9u), h2 = hash_u32(h1 ^ 0x68bc21ebu);
    const float r1 = static_cast<float>(h1 >> 8) * (2.0f / 16777216.0f) - 1.0f, r2 = static_cast<float>(h2 >> 8) * (2.0f / 16777216.0f) - 1.0f;
    const float inv_n = 1.0f / static_cast<float>(sc.supersampling);
    const float lx = (sdx + r1 * inv_n) * sc.lens_size, ly = (sdy + r2 * inv_n) * sc.lens_size;
    V3 origin = xform_point(sc.cam, v3(0.0f, 0.0f, 0.0f));
    origin = {origin.x + lx, origin.y + ly, origin.z};
    L.sp = 0;
    L.resident = 0;
    L.hot = 0u;
    L.hue = sc.dispersion ? static_cast<int>(hue) : -1;
    L.wavelength = sc.dispersion ? (static_cast<float>(hue) * 45.0f) / 180.0f - 1.0f : 0.0f;
    L.out_slot = static_cast<unsigned long long>(s) * job.n_pixels + static_cast<unsigned long long>(yc) * sc.width + x;
    begin_query(L, origin, normalized(on_focus - origin), kInfinite, Q_TRACE);
    return true;
}

// Color::scaleOut (src/types.h:274-282): u8(clamp(v * gain, 0, 1) * 255.0f), truncating; NaN -> 0 like the x86 cast.
RT_DEV uchar4 scale_out(float4 c, float gain) {
    auto q = [gain](float v) -> unsigned char {
        const float s = clampf(v * gain, 0.0f, 1.0f) * 255.0f;
        return static_cast<unsigned char>(s != s ? 0.0f : s);
    };
    return make_uchar4(q(c.x), q(c.y), q(c.z), q(c.w));
}

// Stores a finished primary sample (src/raytracer.cpp:100-110).
RT_DEV void store_sample(const DevScene& sc, const DevJob& job, const Lane& L, Col value) {
    Col contrib = value;
    if (L.hue >= 0) {
        const Col hc = {sc.hue[L.hue][0], sc.hue[L.hue][1], sc.hue[L.hue][2], sc.hue[L.hue][3]};
        contrib = (value * hc) / 4.0f;
    }
    if (job.samples_per_pixel == 1) {
        const Col zero = {0.0f, 0.0f, 0.0f, 0.0f};
        const Col px = zero + (zero + contrib) * sc.inv_samples;
        if (job.out8 != nullptr) job.out8[L.out_slot] = scale_out(make_float4(px.r, px.g, px.b, px.a), job.out8_gain);
        else job.out[L.out_slot] = make_float4(px.r, px.g, px.b, px.a);
    } else {
        job.samples[L.out_slot] = make_float4(contrib.r, contrib.g, contrib.b, contrib.a);
    }
}

// Shadow query towards light L.li from the current node (src/raytracer.cpp:214-220).
RT_DEV void begin_shadow(const ShadeTables& T, Lane& L) {
    const DevLight& light = T.lights[L.li];
    const V3 lpos = v3(light.position[0], light.position[1], light.position[2]);
    const V3 dir = normalized(light.type == 0 ? lpos * -1.0f : lpos - L.n_point);
    const V3 origin = L.n_point + L.n_normal * kEpsilon;
    const float ldist = light.type == 0 ? kInfinite : length(lpos - origin);
    begin_query(L, origin, dir, ldist, Q_SHADOW);
}

// Section timer of the counters build: the first active lane of the wave charges the cycles since
// the previous mark to `slot`.
template <bool STATS>
RT_DEV void mark(Counters& n, long long& t, int slot) {
    if (STATS) {
        const long long now = clock64();
        const unsigned long long m = __ballot(1);
        if ((threadIdx.x & 63) == static_cast<unsigned>(__ffsll(static_cast<long long>(m)) - 1)) n.v[slot] += static_cast<unsigned long long>(now - t);
        t = now;
    }
}

// pow(double, double) of the specular term (src/raytracer.cpp:233; Y8), ~250 instructions that few lit points need
// (resolve_lane): out of line, so that its registers and its code stay out of the resolve phase's way.
__device__ __attribute__((noinline)) float specular_pow(float base, float exponent) {
    return static_cast<float>(pow(static_cast<double>(base), static_cast<double>(exponent)));
}

// First light at or after `li` whose shadow ray has to be traced: lights that can only add +-0 at this point are passed
// over (resolve_lane explains).  Out of line: inlined, its few live values cost the Julia kernel 70 more spilled VGPRs.
// The light direction, Lambert term, reflected direction and specular base are formed exactly as begin_shadow and the
// shading code form them (src/raytracer.cpp:214-233), so "zero" here is the zero they would multiply by.
template <bool STATS>
__device__ __attribute__((noinline)) unsigned skip_dark_lights(const DevLight* lights, unsigned n_lights, unsigned li, V3 point, V3 normal, V3 view_dir,
                                                               float kd, float ks, float exponent, bool elide, Counters& n) {
    if (!(fabsf(point.x) + fabsf(point.y) + fabsf(point.z) < __builtin_inff() && fabsf(normal.x) + fabsf(normal.y) + fabsf(normal.z) < __builtin_inff())) return li;
    while (li < n_lights) {
        const DevLight& light = lights[li];
        const V3 lpos = v3(light.position[0], light.position[1], light.position[2]);
        const V3 to_light = light.type == 0 ? lpos * -1.0f : lpos - point;
        const float dd = dot(to_light, to_light);
        if (!(dd > 0.0f && dd < __builtin_inff())) break;
        const V3 dir = normalized(to_light);
        const float ndotl = dot(dir, normal);
        if (!(kd == 0.0f || ndotl <= 0.0f)) break;                  // max(ndotl, 0) * kd != 0: lit
        if (ks != 0.0f) {
            if (!(exponent > 0.0f)) break;                          // pow(0, 0) = 1
            const V3 refl = normalized(normal * ndotl * 2.0f - dir);
            const float rv = dot(refl, view_dir * -1.0f);
            if (!(rv <= 0.0f)) break;                               // max(rv, 0) != 0 (or NaN): a highlight is possible
        }
        bump<STATS>(n, C_SHADOW_ELIDED);
        if (!elide) break;
        li++;
    }
    return li;
}

// ---- resolve phase: advance one lane's ray tree by one query result ---------------------------------
template <bool STATS, typename Frames>
RT_DEV void resolve_lane(const DevScene& sc, const ShadeTables& T, const DevJob& job, const Frames& frames, Lane& L, Counters& n,
                         bool& sample_done, Col& sample_value) {
    bool lights_done = false;
    bool returning = false;
    float val[3] = {0.0f, 0.0f, 0.0f};
    float val_a = 0.0f;
    long long t_sec = STATS ? clock64() : 0;

    if (L.qkind == Q_TRACE) {
        finish_best_triangle(T, L);
        bump<STATS>(n, L.sp == 0 ? C_PRIMARY : C_SECONDARY);
        if (!L.found) {
            val[0] = sc.background[0];
            val[1] = sc.background[1];
            val[2] = sc.background[2];
            val_a = sc.background[3];
            returning = true;
        } else {
            bump<STATS>(n, C_SHADED_HITS);
            L.n_point = L.best.point;
            L.n_normal = L.best.normal;
            L.n_dir = L.qd;
            L.n_cos = L.best_cos;
            L.n_obj = L.best_obj;
            L.n_ph = L.best.ph;
            L.acc[0] = L.acc[1] = L.acc[2] = 0.0f;
            L.li = 0;
            if (L.n_cos < 0.0f) {
                const DevMaterial& m = T.materials[T.object_material[L.n_obj]];
                Col mc = {m.color[0], m.color[1], m.color[2], m.color[3]};
                if (m.texture >= 0) {
                    bump<STATS>(n, C_TEXTURE_LOOKUPS);
                    mc = texture_bilinear(T.textures[m.texture], sc.texels, L.best.uv);
                }
                L.matcol[0] = mc.r;
                L.matcol[1] = mc.g;
                L.matcol[2] = mc.b;
                // rad += ambient * materialColor * ka   (src/raytracer.cpp:212)
                L.acc[0] = 0.0f + sc.ambient[0] * mc.r * m.ka;
                L.acc[1] = 0.0f + sc.ambient[1] * mc.g * m.ka;
                L.acc[2] = 0.0f + sc.ambient[2] * mc.b * m.ka;
            } else {
                lights_done = true;   // back side of a transparent object: no Phong, no photons
            }
        }
    } else {   // Q_SHADOW result for light L.li (src/raytracer.cpp:227-235)
        bump<STATS>(n, C_SHADOW);
        bump<STATS>(n, C_PHONG_LIGHTS);
        if (!L.blocked) {
            const DevMaterial& m = T.materials[T.object_material[L.n_obj]];
            const DevLight& light = T.lights[L.li];
            const float ndotl = dot(L.qd, L.n_normal);
            const float lambert = ndotl < 0.0f ? 0.0f : ndotl;   // std::max(x, 0.0f)
            const V3 refl = normalized(L.n_normal * ndotl * 2.0f - L.qd);
            const float rv = dot(refl, L.n_dir * -1.0f);
            const float sbase = rv < 0.0f ? 0.0f : rv;
            float diffuse[3];
            for (int c = 0; c < 3; c++) diffuse[c] = light.power[c] * L.matcol[c] * lambert * m.kd;
            // pow(double, double) is ~250 instructions; most lit points are far from the highlight, where
            // its value cannot change the sum below.  With ub >= pow(sbase, exponent) (hardware log2/exp2,
            // 1 % and 2^-6 in the exponent to spare) the term is dropped when it is below a quarter ulp of
            // the diffuse term it is added to in every channel (then fl(diffuse + specular) == diffuse),
            // or when the float result is exactly zero; everything else takes the reference's expression.
            float spow = 0.0f;
            bool exact = true;
            if (m.exponent > 0.0f && m.exponent < 1e6f) {
                if (sbase == 0.0f) {
                    exact = false;                                   // pow(+0, y > 0) = +0
                } else if (sbase < 1.0f) {
                    const float lg = m.exponent * __builtin_amdgcn_logf(sbase) + 0.015625f;   // log2, rounded up
                    if (lg < -160.0f) {
                        exact = false;                               // < 2^-150: rounds to +0 as a float
                    } else {
                        const float ub = __builtin_amdgcn_exp2f(lg) * 1.01f;
                        bool negligible = true;
                        for (int c = 0; c < 3; c++) {
                            const float s_ub = fabsf(light.power[c] * m.ks) * ub;
                            negligible = negligible && (light.power[c] * m.ks == 0.0f || s_ub < fabsf(diffuse[c]) * 0x1p-26f);
                        }
                        // the products power*spow*ks of the dropped term must not matter either way: they
                        // are below 2^-26 of a finite diffuse term, so they are finite and the sum is diffuse
                        if (negligible) exact = false;
                    }
                }
            }
            if (exact) spow = specular_pow(sbase, m.exponent);
            for (int c = 0; c < 3; c++) {
                const float specular = light.power[c] * spow * m.ks;
                L.acc[c] = L.acc[c] + (diffuse[c] + specular);
            }
        }
        L.li++;
    }

    mark<STATS>(n, t_sec, L.qkind == Q_TRACE ? D_R_NODE_CYCLES : D_R_LIGHT_CYCLES);
    if (!returning && !lights_done) {
        // Lights that can only add +-0.  The reference casts a shadow ray per light and, if it is not blocked, adds
        // power*colour*max(n.l, 0)*kd + power*pow(max(r.v, 0), exponent)*ks (src/raytracer.cpp:214-235).  Where the light
        // is behind the surface (n.l <= 0) or kd == 0, and the mirrored light direction points away from the viewer
        // (r.v <= 0, exponent > 0: pow(0, e) = 0) or ks == 0 — the back of every object as seen from a light, all of a
        // mirror or of clear glass — that is +-0 onto a sum that is +0 or non-zero (it starts as 0.0f + ambient term),
        // i.e. nothing, blocked or not, provided no factor is infinite or NaN: DevMaterial::phong_tame,
        // DevScene::lights_benign and skip_dark_lights' checks (finite point and normal, a light direction of finite
        // non-zero length; then n.l, the reflected direction and pow's base are finite and pow <= e^0.25) see to that.
        // Such rays are not traced.  A counters launch traces them (its counts are those of the reference's formulation)
        // and reports how many the plain launch skips.  (The test in front of the call is a filter only: the sign of n.l
        // before the direction is normalised; the function decides on the values the shading code would use.)
        if (L.li < sc.n_lights && sc.lights_benign != 0u) {
            const DevMaterial& m = T.materials[T.object_material[L.n_obj]];
            if (m.phong_tame != 0u) {
                bool maybe = m.kd == 0.0f;
                if (!maybe) {
                    const DevLight& light = T.lights[L.li];
                    const V3 lpos = v3(light.position[0], light.position[1], light.position[2]);
                    maybe = dot(light.type == 0 ? lpos * -1.0f : lpos - L.n_point, L.n_normal) <= 0.0f;
                }
                if (maybe) L.li = skip_dark_lights<STATS>(T.lights, sc.n_lights, L.li, L.n_point, L.n_normal, L.n_dir, m.kd, m.ks, m.exponent, job.elide_unlit != 0u, n);
            }
        }
        if (L.li < sc.n_lights) {
            begin_shadow(T, L);
            mark<STATS>(n, t_sec, D_R_SHADOWSETUP_CYCLES);
            return;
        }
        // rad += object.getPhoton(photonCoordinate)   (src/raytracer.cpp:146)
        const int slot = T.object_photon_slot[L.n_obj];
        if (slot >= 0 && sc.photon_atlas != nullptr) {
            const unsigned texel = photon_texel(L.n_ph, sc.photon_size);
            const float4 ph = sc.photon_atlas[static_cast<size_t>(slot) * sc.photon_size * sc.photon_size + texel];
            L.acc[0] += ph.x;
            L.acc[1] += ph.y;
            L.acc[2] += ph.z;
        }
        lights_done = true;
    }

    if (!returning) {
        // Fresnel split (src/raytracer.cpp:151-159)
        const DevMaterial& m = T.materials[T.object_material[L.n_obj]];
        bool descended = false;
        if ((m.transmittance != 0.0f || m.reflectance != 0.0f) && m.refr_norm > 0.0f) {
            bump<STATS>(n, C_FRESNEL);
            const float kr = fresnel_kr(m.refr_re, m.refr_im, m.dispersion, L.n_cos, L.wavelength);
            const bool want_refr = m.transmittance != 0.0f && kr < 1.0f;
            const bool want_refl = m.reflectance != 0.0f && kr > 0.0f;
            // a child at depth sp+1 > max_bounces returns zero radiance: nothing to add
            if (L.sp + 1 <= sc.max_bounces && (want_refr || want_refl)) {
                V3 ro, rd, mo, md;
                const bool refr_ok = want_refr && refraction_ray(L.n_dir, L.n_point, L.n_normal, m.refr_re, m.dispersion, L.n_cos, L.wavelength, ro, rd);
                if (want_refl) reflection_ray(L.n_dir, L.n_point, L.n_normal, L.n_cos, mo, md);
                if (refr_ok || want_refl) {
                    const unsigned f = L.sp;
                    frames.before_push(f);
                    frames.set(f, 0, L.acc[0]);
                    frames.set(f, 1, L.acc[1]);
                    frames.set(f, 2, L.acc[2]);
                    frames.set(f, 3, kr);
                    frames.set(f, 4, mo.x);
                    frames.set(f, 5, mo.y);
                    frames.set(f, 6, mo.z);
                    frames.set(f, 7, md.x);
                    frames.set(f, 8, md.y);
                    frames.set(f, 9, md.z);
                    // flags: bit0 = the child now in flight is the reflection, bit1 = reflection still to do
                    frames.set(f, 10, __uint_as_float(refr_ok ? (want_refl ? 2u : 0u) : 1u));
                    L.sp = f + 1;
                    if (refr_ok) begin_query(L, ro, rd, kInfinite, Q_TRACE); else begin_query(L, mo, md, kInfinite, Q_TRACE);
                    descended = true;
                }
            }
        }
        mark<STATS>(n, t_sec, D_R_FRESNEL_CYCLES);
        if (descended) return;
        val[0] = L.acc[0];
        val[1] = L.acc[1];
        val[2] = L.acc[2];
        val_a = 1.0f;   // withoutAlpha
    }

    // return `val` to the parent frames (src/raytracer.cpp:153-160)
    for (;;) {
        if (L.sp == 0) {
            sample_done = true;
            sample_value = {val[0], val[1], val[2], val_a};
            L.state = S_IDLE;
            mark<STATS>(n, t_sec, D_R_RETURN_CYCLES);
            return;
        }
        const unsigned f = L.sp - 1;
        const unsigned flags = __float_as_uint(frames.get(f, 10));
        const float kr = frames.get(f, 3);
        const float w = (flags & 1u) ? kr : 1.0f - kr;
        float acc[3];
        for (int c = 0; c < 3; c++) acc[c] = frames.get(f, c) + val[c] * w;
        if (flags & 2u) {
            // refraction child done, reflection child next
            for (int c = 0; c < 3; c++) frames.set(f, c, acc[c]);
            frames.set(f, 10, __uint_as_float(1u));
            begin_query(L, v3(frames.get(f, 4), frames.get(f, 5), frames.get(f, 6)), v3(frames.get(f, 7), frames.get(f, 8), frames.get(f, 9)), kInfinite, Q_TRACE);
            mark<STATS>(n, t_sec, D_R_RETURN_CYCLES);
            return;
        }
        frames.after_pop(f);
        L.sp = f;
        val[0] = acc[0];
        val[1] = acc[1];
        val[2] = acc[2];
        val_a = 1.0f;
    }
}

template <bool STATS>
RT_DEV void flush_counters(const Counters& n, unsigned long long* out) {
    if (!STATS) return;
    for (int i = 0; i < C_COUNT; i++) {
        unsigned long long v = n.v[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&out[i], v);
    }
}

// What a lane knows about the node it is shading and about its sample is not touched while its query runs.  Kept
// in registers it is what the compiler spills around the object loop (71 VGPRs of scratch at the 168-register
// budget: ~6 GB of HBM writes per C3 frame, since the scratch of 3072 waves does not stay in L2); trace_kernel
// parks it in LDS at the end of every resolve phase and takes it back at the start of the next one, for all lanes
// alike (uniform control flow), so that the registers are dead in between.
constexpr int kParkWords = 26;
RT_DEV void park_node(float* col, const Lane& L) {
    const float w[kParkWords] = {L.n_point.x, L.n_point.y, L.n_point.z, L.n_normal.x, L.n_normal.y, L.n_normal.z, L.n_dir.x, L.n_dir.y, L.n_dir.z,
                                 L.n_ph.x, L.n_ph.y, L.n_cos, __uint_as_float(L.n_obj), L.acc[0], L.acc[1], L.acc[2], L.matcol[0], L.matcol[1], L.matcol[2],
                                 __uint_as_float(L.li), __uint_as_float(L.sp), L.wavelength, __int_as_float(L.hue),
                                 __uint_as_float(static_cast<unsigned>(L.out_slot)), __uint_as_float(static_cast<unsigned>(L.out_slot >> 32)), __uint_as_float(L.resident)};
#pragma unroll
    for (int i = 0; i < kParkWords; i++) col[i * 256] = w[i];
}
RT_DEV void unpark_node(const float* col, Lane& L) {
    float w[kParkWords];
#pragma unroll
    for (int i = 0; i < kParkWords; i++) w[i] = col[i * 256];
    L.n_point = v3(w[0], w[1], w[2]);
    L.n_normal = v3(w[3], w[4], w[5]);
    L.n_dir = v3(w[6], w[7], w[8]);
    L.n_ph = {w[9], w[10]};
    L.n_cos = w[11];
    L.n_obj = __float_as_uint(w[12]);
    L.acc[0] = w[13];
    L.acc[1] = w[14];
    L.acc[2] = w[15];
    L.matcol[0] = w[16];
    L.matcol[1] = w[17];
    L.matcol[2] = w[18];
    L.li = __float_as_uint(w[19]);
    L.sp = __float_as_uint(w[20]);
    L.wavelength = w[21];
    L.hue = __float_as_int(w[22]);
    L.out_slot = static_cast<unsigned long long>(__float_as_uint(w[23])) | (static_cast<unsigned long long>(__float_as_uint(w[24])) << 32);
    L.resident = __float_as_uint(w[25]);
}

// ---- the render kernel (scenes without Julia sets) ----------------------------------------------------
// DENSE: 4 waves per SIMD at 128 VGPRs (20 of them spilled) instead of 3 at 168 (none): the waves wait a third of their
// time, and where a frame has the samples to fill a fourth wave per SIMD and its workgroups fit four to a CU in LDS,
// the extra wave hides more than the spills cost (C3 10.0 -> 9.2 ms; C2 does not fit, C5 is too small: rt_api.hip).
template <bool STATS, bool DENSE>
__global__ __launch_bounds__(256, DENSE ? 4 : RT_TRACE_WAVES_PER_SIMD) void trace_kernel(const DevScene sc_global, const DevJob job) {
    // With a small scene the shading tables are read from an LDS copy of the shade blob (latency
    // ~64 cycles instead of an L2 round trip per dependent load); the wave-uniform geometry reads
    // keep using the HBM tables (scalar loads).
    extern __shared__ __align__(16) unsigned char lds_blob[];
    const DevScene& sc = sc_global;
    const unsigned char* table_base = sc.shade_blob;
    if (sc.shade_blob_in_lds) {
        for (unsigned i = threadIdx.x * 16u; i < sc.shade_blob_bytes; i += 256u * 16u) {
            *reinterpret_cast<uint4*>(lds_blob + i) = *reinterpret_cast<const uint4*>(sc.shade_blob + i);
        }
        __syncthreads();
        table_base = lds_blob;
    }
    const ShadeTables T = tables_of(sc, table_base);
    unsigned char* const lds_after_blob = lds_blob + (sc.shade_blob_in_lds ? ((sc.shade_blob_bytes + 15u) & ~15u) : 0u);
    // dynamic LDS: [shade blob copy][BVH stack columns][second frame slot][BVH records]
    const DevBvhPair* bvh_nodes = sc.bvh_nodes;
    if (sc.bvh_records_in_lds != 0u) {
        uint4* const dst = reinterpret_cast<uint4*>(lds_after_blob + sc.bvh_lds_stack_entries * 256u * sizeof(unsigned) +
                                                    (sc.frame_slots_in_lds == 2u ? kFrameFields * 256u * sizeof(float) : 0u));
        const uint4* const src = reinterpret_cast<const uint4*>(sc.bvh_nodes);
        for (unsigned i = threadIdx.x; i < sc.bvh_records_in_lds * 4u; i += 256u) dst[i] = src[i];
        __syncthreads();
        bvh_nodes = reinterpret_cast<const DevBvhPair*>(dst);
    }
    const BvhStackLds bvh_stack = {reinterpret_cast<unsigned*>(lds_after_blob) + threadIdx.x, sc.bvh_lds_stack_entries, 256u, bvh_nodes};
    Counters n;
    if (STATS) for (int i = 0; i < C_COUNT; i++) n.v[i] = 0;
    const unsigned long long grid_lanes = static_cast<unsigned long long>(gridDim.x) * blockDim.x;
    const unsigned long long gid = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    const FrameRef stack = {job.stack + gid, grid_lanes};

    Lane L;
    __shared__ float top_store[kFrameFields * 256];
    __shared__ float park_store[kParkWords * 256];
    float* const park = park_store + threadIdx.x;
    // a second slot per lane behind the blob copy and the BVH stack columns, when the host found room for it
    float* const second_slot = reinterpret_cast<float*>(lds_blob + (sc.shade_blob_in_lds ? ((sc.shade_blob_bytes + 15u) & ~15u) : 0u) +
                                                        sc.bvh_lds_stack_entries * 256u * sizeof(unsigned)) + threadIdx.x;
    const bool two_slots = sc.frame_slots_in_lds == 2u;
    const TopInLds frames = {top_store + threadIdx.x, second_slot, 256u, two_slots ? 1u : 0u, stack, &L.resident};
    L.state = S_RESOLVE;       // "resolve" of a lane without a sample = fetch one
    L.qkind = -1;
    L.sp = 0;
    L.resident = 0;
    L.jidx = 0;
    L.jstep = 0;
    L.n_point = L.n_normal = L.n_dir = v3(0.0f, 0.0f, 0.0f);
    L.n_ph = {0.0f, 0.0f};
    L.n_cos = 0.0f;
    L.n_obj = L.li = 0u;
    L.acc[0] = L.acc[1] = L.acc[2] = L.matcol[0] = L.matcol[1] = L.matcol[2] = 0.0f;
    L.wavelength = 0.0f;
    L.hue = -1;
    L.out_slot = 0ull;
    park_node(park, L);
    bool fresh = true;         // lane holds no sample yet
    bool want = false;         // lane is idle and waits for the next refill
    WaveIds wave_ids = wave_ids_begin(gid, job.n_waves, job.static_blocks);
#ifdef RT_PHASE_TIMING
    unsigned long long phase_clocks[4] = {0, 0, 0, 0};
#endif

    for (;;) {
        // every lane alternates query -> resolve in lockstep
        const bool any_query = __any(L.state == S_QUERY);
        const bool any_resolve = __any(L.state == S_RESOLVE);
        if (!any_query && !any_resolve) break;
        const int phase = any_query ? S_QUERY : S_RESOLVE;

        long long t_phase = 0, t_fetch_outer = 0;
#ifdef RT_PHASE_TIMING      // developer build: phase clocks without the per-lane counters of the STATS build
        if (!STATS) t_phase = __builtin_readcyclecounter();
#endif
        if (STATS) {
            t_phase = clock64();
            const int base = phase == S_QUERY ? D_QUERY_PHASES : D_RESOLVE_PHASES;
            if ((threadIdx.x & 63) == 0) n.v[base]++;
            if (L.state == phase) n.v[base + 1]++;
        }
        if (phase == S_QUERY) {
            const bool mine = L.state == S_QUERY;
            const HitSink sink = {park, 256u, 0, 3, 6, 9, 10};
            query_objects<STATS>(sc, L, mine, bvh_stack, n, sink);
            if (mine) L.state = S_RESOLVE;
        } else {   // S_RESOLVE
            const bool mine = L.state == S_RESOLVE;
            bool done = false;
            Col value = {0.0f, 0.0f, 0.0f, 0.0f};
            unpark_node(park, L);
            // a sphere winner's record was written into the parking slots (accept_hit, HitSink); a triangle winner's is built by
            // finish_best_triangle.  Unconditional on purpose: the registers must not carry a value across the query.
            L.best.point = L.n_point;
            L.best.normal = L.n_normal;
            L.best.uv = {L.n_dir.x, L.n_dir.y};
            L.best.ph = L.n_ph;
            if (STATS) {
                const int n_trace = __popcll(__ballot(mine && !fresh && L.qkind == Q_TRACE)), n_shadow = __popcll(__ballot(mine && !fresh && L.qkind == Q_SHADOW));
                if ((threadIdx.x & 63) == 0 && n_trace > 0 && n_shadow > 0) {
                    n.v[D_RESOLVE_MIXED]++;
                    n.v[D_RESOLVE_MINORITY_LANES] += static_cast<unsigned long long>(n_trace < n_shadow ? n_trace : n_shadow);
                }
            }
            if (mine && !fresh) resolve_lane<STATS>(sc, T, job, frames, L, n, done, value);
            t_fetch_outer = STATS ? clock64() : 0;
            if (mine && done) store_sample(sc, job, L, value);
            if (mine && (fresh || done)) {
                L.state = S_IDLE;
                want = true;
            }
            // refill from the queue once RT_REFILL_MIN lanes are idle, or when nothing else is left to do;
            // padding lanes of edge tiles simply ask again
            const int n_want = __popcll(__ballot(want)), n_live = __popcll(__ballot(L.state == S_QUERY));
            bool need = want && (n_want >= RT_REFILL_MIN || n_live == 0);
            if (need) want = false;
            while (__any(need)) {
                unsigned long long id = 0;
                const bool got = fetch_ids(job.queue_head, job.dyn_base, job.n_samples, job.id_chunk, wave_ids, need, id);
                if (need) {
                    if (!got) {
                        need = false;   // queue is exhausted: this lane stays idle
                    } else if (start_sample(sc, job, id, L)) {
                        need = false;
                        fresh = false;
                    }
                }
            }
            park_node(park, L);
        }
#ifdef RT_PHASE_TIMING
        if (!STATS) {
            const unsigned long long dt = static_cast<unsigned long long>(__builtin_readcyclecounter() - t_phase);
            phase_clocks[phase == S_QUERY ? 0 : 1] += dt;
            if (phase == S_QUERY) phase_clocks[3]++;
        }
#endif
        if (STATS && phase == S_RESOLVE && (threadIdx.x & 63) == 0) n.v[D_R_FETCH_CYCLES] += static_cast<unsigned long long>(clock64() - t_fetch_outer);
        if (STATS && (threadIdx.x & 63) == 0) {
            const int slot = phase == S_QUERY ? D_QUERY_CYCLES : D_RESOLVE_CYCLES;
            n.v[slot] += static_cast<unsigned long long>(clock64() - t_phase);
        }
    }
    flush_counters<STATS>(n, job.counters);
#ifdef RT_PHASE_TIMING
    if (!STATS && (threadIdx.x & 63) == 0) {
        atomicAdd(&job.counters[D_QUERY_CYCLES], phase_clocks[0]);
        atomicAdd(&job.counters[D_RESOLVE_CYCLES], phase_clocks[1]);
        atomicAdd(&job.counters[D_R_FETCH_CYCLES], phase_clocks[2]);
        atomicAdd(&job.counters[D_QUERY_PHASES], phase_clocks[3]);
    }
#endif
}

// ---- the render kernel for scenes with Julia sets: a ray pool with queues by kind -----------------------
// Marching a Julia set costs 10^2..10^6 times what the rest of a ray's life costs, and a ray's marches come
// back at unrelated times.  A wave that owned its rays in registers (round 1) therefore ran its shading code
// for whichever few lanes had a result (11 of 64 on C4).  Here nothing owns a ray:
//   * every ray tree in flight is a CONTEXT, a 160-byte record in the workgroup's LDS pool (kPoolCtx of them:
//     the current query, the Julia job and its result, the best hit so far or the shaded node, the sample id);
//     the frames of its DFS stack are 48-byte records in HBM;
//   * four rings of context ids in LDS connect the stages: READY_T (a trace query is complete or a context needs
//     a new sample), READY_S (a shadow query is complete), MARCH (Julia jobs), NORMAL (converged marches);
//   * SHADER waves pop up to 64 contexts of ONE kind, load them into lanes, and run collect -> resolve -> next
//     query at full width (all lanes take the same branch of the shading code: wavefront compaction by kind);
//     the new query's Julia job goes on MARCH, a query that met no Julia set straight back on a READY ring;
//   * MARCHER waves pop jobs into all 64 lanes and march them (estimator iterations in blocks of four for the
//     wave, march steps together once enough lanes wait for one), refilling a lane the moment its job ends;
//     converged jobs are finished (6-tap normal, world-space hit) 64 at a time from the NORMAL ring.
// One workgroup per CU; queues, contexts and results never leave LDS.
constexpr int kPoolCtx = RT_JULIA_POOL;                       // contexts per workgroup
#ifndef RT_JULIA_HELPERS
#define RT_JULIA_HELPERS 64             // contexts set aside for the shadow rays that long samples send out in parallel (a multiple of 32; 0: none)
#endif
constexpr int kHelperCtx = RT_JULIA_HELPERS;                  // the last kHelperCtx contexts of the pool
constexpr int kSampleCtx = kPoolCtx - kHelperCtx;             // contexts that hold a sample's ray tree
static_assert(kHelperCtx % 32 == 0 && kHelperCtx < kPoolCtx, "helper contexts");
constexpr int kJuliaConstTable = 16;                          // Julia sets whose constants the workgroup keeps in LDS
constexpr int kCtxWords = 40;
constexpr unsigned kJuliaQueueCap = 1024u;                    // a context is on at most one ring
static_assert(kPoolCtx % 64 == 0 && kPoolCtx <= static_cast<int>(kJuliaQueueCap), "pool size");
static_assert((kJuliaQueueCap & (kJuliaQueueCap - 1)) == 0, "ring size");
#ifndef RT_JULIA_NORMAL_IDLE
#define RT_JULIA_NORMAL_IDLE 8             // a marcher wave with fewer jobs than this finishes a partial batch of converged jobs
#endif
#ifndef RT_JULIA_STEP_EIGHTHS
#define RT_JULIA_STEP_EIGHTHS 5            // march steps are taken once this many eighths of a wave's jobs have one due
#endif
#ifndef RT_JULIA_JOBS_AHEAD
#define RT_JULIA_JOBS_AHEAD 64u            // queued Julia jobs above which a shader wave may wait for a fuller batch
#endif
#ifndef RT_JULIA_SHADER_MIN_BATCH
#define RT_JULIA_SHADER_MIN_BATCH 48       // a shader wave waits for this many ready contexts of one kind while the marchers have work
#endif
// The watchdog gives a launch up on two kinds of evidence together: no progress in the workgroup for kStuckTicks of the
// 100 MHz wall clock (1 s) AND kStuckPolls looks at the heartbeat by THIS wave in that time, none of them after a gap.
// Wall time alone is not evidence: a workgroup that the hardware scheduler switches out (another process's grid on
// the same GPU, a debugger, a preempted queue) sleeps with its clock running, and the first wave restored would
// find a second gone by with the heartbeat unchanged before the working waves are back.  A poll that comes more than
// kPollGapTicks after the previous one (0.5 ms; a poll takes a microsecond) therefore starts the wait afresh.
constexpr unsigned long long kStuckTicks = 100000000ull;
constexpr unsigned kStuckPolls = 20000u;
constexpr unsigned long long kPollGapTicks = 50000ull;

enum CtxWord : int {
    CW_QO = 0, CW_QMAX = 3, CW_QD = 4, CW_FLAGS = 7,
    CW_JP = 8, CW_JDIST = 11, CW_JD = 12, CW_JSTATUS = 15,   // job in: position, direction (object space); out: world point, distance, normal, status
    CW_UNION = 16,                                           // 20 words: best hit so far (trace query) or the shaded node (shadow query)
    CW_SLOT_LO = 36, CW_SLOT_HI = 37, CW_MISC = 38, CW_JIDX = 39
};
enum CtxFlag : unsigned { CF_KIND_MASK = 3u, CF_QACTIVE = 4u, CF_BLOCKED = 8u, CF_FOUND = 16u, CF_WAITING = 32u, CF_FRESH = 64u,
                          CF_HELPER = 128u /* a shadow ray sent out on behalf of context CW_SLOT_LO & 0xffff, light CW_SLOT_LO >> 16 */,
                          CF_NEWQUERY = 256u /* ... that has not met the scene yet */ };
// The join word of a context (CW_JOIN, free in both uses of the union words): bit j = light j's shadow ray is with a helper,
// bit 8 + j = its answer is in, bit 16 + j = blocked, bit 24 + j = the context is parked until that answer comes.
constexpr int CW_JOIN = 16 + 19;
constexpr unsigned kMaxForkLights = 8u;
constexpr unsigned kCtxHotBit = 16u;      // in CW_MISC: Lane::hot
enum JobStatus : unsigned { JOB_PENDING = 0u, JOB_MISS = 1u, JOB_HIT = 2u };
enum RingId : int { R_MARCH = 0, R_NORMAL = 1, R_READY_T = 2, R_READY_S = 3, R_TAIL = 4, R_COUNT = RT_JULIA_TAIL_RING ? 5 : 4 };

// Entry i of a ring lives at slot[i & (cap-1)] as ring_word(i, context id): one 16-bit store publishes the id and the
// lap it belongs to (a slot holds the previous lap's word until its producer has written, so the lap number tells a
// consumer that came first to wait; 16-bit words: four rings of 1024 entries and 960 contexts fill the CU's 160 KB).
struct JuliaRing {
    unsigned short slot[kJuliaQueueCap];
};
static_assert(kJuliaQueueCap == 1024u, "ring_word packs a 10-bit context id under a 5-bit lap number");
RT_DEV unsigned short ring_word(unsigned i, unsigned id) { return static_cast<unsigned short>(0x8000u | (((i >> 10) & 31u) << 10) | id); }
RT_DEV bool ring_word_is(unsigned v, unsigned i) { return (v >> 10) == (32u | ((i >> 10) & 31u)); }
struct PoolShared {
    // running counts (never wrapped) of the rings: entries head..tail-1 of ring r are reserved, ctl[2r] = head, ctl[2r+1] = tail
    __attribute__((aligned(16))) unsigned ctl[12];
    unsigned dead;          // contexts that found the sample queue empty
    unsigned abort;
    unsigned heartbeat;     // bumped by every wave that made progress (watchdog)
    unsigned pad;
    unsigned helper_free[kHelperCtx / 32 + 1];   // bit b of word w: helper context kSampleCtx + 32 w + b is free
    __attribute__((aligned(8))) unsigned long long hand;   // fetch_ids_scheduled: the workgroup's block of sample ids
    __attribute__((aligned(16))) float jconst[kJuliaConstTable][8];   // c (4 floats) and the cut plane of the scene's first Julia sets
    JuliaRing ring[R_COUNT];
    __attribute__((aligned(16))) float ctx[kPoolCtx][kCtxWords];
};
static_assert(sizeof(PoolShared) <= kJuliaStaticLdsBytes, "kJuliaStaticLdsBytes");
static_assert(2 * R_COUNT <= 12, "ctl");

RT_DEV unsigned lds_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
RT_DEV void lds_store(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
RT_DEV unsigned lds_load(const unsigned short* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
RT_DEV void lds_store(unsigned short* p, unsigned short v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
RT_DEV void lds_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
RT_DEV void lds_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }

// Lanes with `want` set append their context id to the ring (one LDS atomic per wave).  Everything the lane wrote
// to the context before the call is visible to whoever pops the entry.
RT_DEV void ring_push(JuliaRing& q, unsigned* tail_word, bool want, unsigned id) {
    const unsigned long long mask = __ballot(want);
    if (mask == 0ull) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll(static_cast<long long>(mask)) - 1;
    unsigned base = 0;
    if (lane == leader) base = __hip_atomic_fetch_add(tail_word, static_cast<unsigned>(__popcll(mask)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    base = __shfl(base, leader);
    if (want) {
        const unsigned i = base + __popcll(mask & ((1ull << lane) - 1ull));
        lds_release();
        lds_store(&q.slot[i & (kJuliaQueueCap - 1)], ring_word(i, id));
    }
}
// Reserves up to `want_n` entries for this wave; returns how many and the index of the first.
RT_DEV unsigned ring_pop(unsigned* head_word, const unsigned* tail_word, unsigned want_n, unsigned& first) {
    unsigned got = 0, h = 0;
    if ((threadIdx.x & 63) == 0 && want_n > 0) {
        for (int attempt = 0; attempt < 64; attempt++) {
            h = lds_load(head_word);
            const unsigned t = lds_load(tail_word);
            const unsigned avail = t - h;
            if (avail == 0u || avail > kJuliaQueueCap) break;
            const unsigned take = avail < want_n ? avail : want_n;
            unsigned expect = h;
            if (__hip_atomic_compare_exchange_strong(head_word, &expect, h + take, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                got = take;
                break;
            }
        }
    }
    got = __shfl(got, 0);
    first = __shfl(h, 0);
    return got;
}
// Context id of reserved entry i (waits for the producer's write; bounded by the 100 MHz clock).
RT_DEV unsigned ring_entry(JuliaRing& q, unsigned i, unsigned* abort_flag) {
    const unsigned short* w = &q.slot[i & (kJuliaQueueCap - 1)];
    unsigned v = lds_load(w);
    if (!ring_word_is(v, i)) {
        // (the producer reserved the slot and is about to write it: normally a few hundred cycles)
        unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), last = t0;
        unsigned polls = 0;
        while (!ring_word_is(v = lds_load(w), i)) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (now - last > kPollGapTicks) {      // this wave was away: so, probably, was the producer
                t0 = now;
                polls = 0;
            }
            last = now;
            polls++;
            if (now - t0 > kStuckTicks && polls > kStuckPolls) {
                lds_store(abort_flag, 1u);
                return 0u;
            }
        }
    }
    lds_acquire();
    return v & 0x3ffu;
}

// A wave that has nothing to do sleeps and watches the workgroup's heartbeat: the launch is given up only when no
// wave of the workgroup has made progress for kStuckTicks (a march of 10 240 steps of 10 000 iterations is slow,
// not stuck: its wave bumps the heartbeat with every block of march steps).
struct IdleWatch {
    unsigned seen;
    unsigned long long since;
    bool armed;
    unsigned long long last = 0ull;   // time of the previous poll
    unsigned polls = 0u;              // polls since `since` that saw the same heartbeat, without a gap between them
};
RT_DEV void idle_wait(PoolShared& sh, IdleWatch& w) {
    __builtin_amdgcn_s_sleep(8);
    const unsigned hb = lds_load(&sh.heartbeat);
    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
    if (!w.armed || hb != w.seen || now - w.last > kPollGapTicks) {
        // progress, or this wave was away (then so was the workgroup: see kStuckPolls): the wait starts afresh
        w.armed = true;
        w.seen = hb;
        w.since = now;
        w.polls = 0u;
    } else if (++w.polls > kStuckPolls && now - w.since > kStuckTicks) {
        lds_store(&sh.abort, 1u);
    }
    w.last = now;
}
RT_DEV void beat(PoolShared& sh) {
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&sh.heartbeat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

typedef float F4 __attribute__((ext_vector_type(4)));
RT_DEV F4 ctx_get(const float* c, int w) { return *reinterpret_cast<const F4*>(c + w); }
RT_DEV void ctx_put(float* c, int w, F4 v) { *reinterpret_cast<F4*>(c + w) = v; }

// Context -> lane.  `kind` is the ring's kind: the union words hold the best hit of a trace query or the node of a
// shadow query.
RT_DEV void ctx_load(const DevScene& sc, const float* c, Lane& L, bool& fresh, unsigned& jstatus, HitRec& jhit, unsigned& flags) {
    const F4 a = ctx_get(c, CW_QO), b = ctx_get(c, CW_QD), jr0 = ctx_get(c, CW_JP), jr1 = ctx_get(c, CW_JD), t = ctx_get(c, CW_SLOT_LO);
    flags = __float_as_uint(b.w);
    L.qo = v3(a.x, a.y, a.z);
    L.qmax = a.w;
    L.qd = v3(b.x, b.y, b.z);
    L.qkind = static_cast<int>(flags & CF_KIND_MASK);
    L.qactive = (flags & CF_QACTIVE) ? 1u : 0u;
    L.blocked = (flags & CF_BLOCKED) ? 1u : 0u;
    L.found = (flags & CF_FOUND) ? 1u : 0u;
    L.state = (flags & CF_WAITING) ? S_WAIT : S_RESOLVE;
    fresh = (flags & CF_FRESH) != 0u;
    jhit.point = v3(jr0.x, jr0.y, jr0.z);
    jhit.dist = jr0.w;
    jhit.normal = v3(jr1.x, jr1.y, jr1.z);
    jhit.uv = {0.0f, 0.0f};
    jhit.ph = {0.0f, 0.0f};
    jstatus = __float_as_uint(jr1.w);
    L.out_slot = static_cast<unsigned long long>(__float_as_uint(t.x)) | (static_cast<unsigned long long>(__float_as_uint(t.y)) << 32);
    const unsigned misc = __float_as_uint(t.z);
    L.hue = static_cast<int>(misc & 15u) - 1;
    L.hot = (misc & kCtxHotBit) != 0u ? 1u : 0u;
    L.sp = (misc >> 8) & 255u;
    L.li = misc >> 16;
    L.wavelength = L.hue >= 0 ? (static_cast<float>(L.hue) * 45.0f) / 180.0f - 1.0f : 0.0f;   // as start_sample
    L.jidx = __float_as_uint(t.w);
    const F4 u0 = ctx_get(c, CW_UNION), u1 = ctx_get(c, CW_UNION + 4), u2 = ctx_get(c, CW_UNION + 8), u3 = ctx_get(c, CW_UNION + 12), u4 = ctx_get(c, CW_UNION + 16);
    if (L.qkind == Q_SHADOW) {
        L.n_point = v3(u0.x, u0.y, u0.z);
        L.n_cos = u0.w;
        L.n_normal = v3(u1.x, u1.y, u1.z);
        L.n_obj = __float_as_uint(u1.w);
        L.n_dir = v3(u2.x, u2.y, u2.z);
        L.n_ph = {u2.w, u3.w};
        L.matcol[0] = u3.x; L.matcol[1] = u3.y; L.matcol[2] = u3.z;
        L.acc[0] = u4.x; L.acc[1] = u4.y; L.acc[2] = u4.z;
        L.best_tri = kNoTriangle;
    } else {
        L.best.dist = u0.x;
        L.best_obj = __float_as_uint(u0.y);
        L.best_cos = u0.z;
        L.best_tri = __float_as_uint(u0.w);
        L.best_u = u1.x;
        L.best_v = u1.y;
        L.best.uv = {u1.z, u1.w};
        L.best.point = v3(u2.x, u2.y, u2.z);
        L.best.normal = v3(u3.x, u3.y, u3.z);
        L.best.ph = {u2.w, u3.w};
    }
    (void)sc;
}
// Lane -> context in two parts, so that what the object loop does not touch is out of the registers while it runs.
// ctx_store_cold (before the loop): the sample's words, the shaded node of a shadow query, and the best record of a
// trace query that does not run the loop in this batch (`keeps_best`: it moves on to the next Julia set).
RT_DEV void ctx_store_cold(float* c, const Lane& L, bool keeps_best) {
    const unsigned misc = static_cast<unsigned>(L.hue + 1) | (L.hot != 0u ? kCtxHotBit : 0u) | (L.sp << 8) | (L.li << 16);
    c[CW_SLOT_LO] = __uint_as_float(static_cast<unsigned>(L.out_slot));
    c[CW_SLOT_HI] = __uint_as_float(static_cast<unsigned>(L.out_slot >> 32));
    c[CW_MISC] = __uint_as_float(misc);
    if (L.qkind == Q_SHADOW) {
        ctx_put(c, CW_UNION, F4{L.n_point.x, L.n_point.y, L.n_point.z, L.n_cos});
        ctx_put(c, CW_UNION + 4, F4{L.n_normal.x, L.n_normal.y, L.n_normal.z, __uint_as_float(L.n_obj)});
        ctx_put(c, CW_UNION + 8, F4{L.n_dir.x, L.n_dir.y, L.n_dir.z, L.n_ph.x});
        ctx_put(c, CW_UNION + 12, F4{L.matcol[0], L.matcol[1], L.matcol[2], L.n_ph.y});
        c[CW_UNION + 16] = L.acc[0];       // (not the fourth word: CW_JOIN, which helpers write)
        c[CW_UNION + 17] = L.acc[1];
        c[CW_UNION + 18] = L.acc[2];
    } else if (keeps_best) {
        c[CW_UNION + 6] = L.best.uv.x;
        c[CW_UNION + 7] = L.best.uv.y;
        ctx_put(c, CW_UNION + 8, F4{L.best.point.x, L.best.point.y, L.best.point.z, L.best.ph.x});
        ctx_put(c, CW_UNION + 12, F4{L.best.normal.x, L.best.normal.y, L.best.normal.z, L.best.ph.y});
    }
}
// The HitSink of a context: accept_hit writes the record of a trace query's winner straight into the union words.
RT_DEV HitSink ctx_hit_sink(float* c) { return {c, 1u, CW_UNION + 8, CW_UNION + 12, CW_UNION + 6, CW_UNION + 11, CW_UNION + 15}; }
// ctx_store_hot (after the loop): the query and its state (waiting: a Julia job goes with it).
RT_DEV void ctx_store_hot(float* c, const Lane& L, bool waiting, bool helper = false) {
    unsigned flags = static_cast<unsigned>(L.qkind) & CF_KIND_MASK;
    if (helper) flags |= CF_HELPER;
    if (L.qactive) flags |= CF_QACTIVE;
    if (L.blocked) flags |= CF_BLOCKED;
    if (L.found) flags |= CF_FOUND;
    if (waiting) flags |= CF_WAITING;
    ctx_put(c, CW_QO, F4{L.qo.x, L.qo.y, L.qo.z, L.qmax});
    ctx_put(c, CW_QD, F4{L.qd.x, L.qd.y, L.qd.z, __uint_as_float(flags)});
    if (waiting) {
        ctx_put(c, CW_JP, F4{L.jp.x, L.jp.y, L.jp.z, 0.0f});
        ctx_put(c, CW_JD, F4{L.jd.x, L.jd.y, L.jd.z, __uint_as_float(JOB_PENDING)});
    }
    c[CW_JIDX] = __uint_as_float(L.jidx);
    if (L.qkind != Q_SHADOW) {
        ctx_put(c, CW_UNION, F4{L.best.dist, __uint_as_float(L.best_obj), L.best_cos, __uint_as_float(L.best_tri)});
        c[CW_UNION + 4] = L.best_u;
        c[CW_UNION + 5] = L.best_v;
    }
}

// ---- marcher waves ------------------------------------------------------------------------------------------
template <bool STATS>
RT_DEV void julia_marcher(const DevScene& sc, const DevJob& job, PoolShared& sh, Counters& n, unsigned long long* counter_words) {
    const int lane = threadIdx.x & 63;
    bool has = false, busy = false;
    unsigned owner = 0, jidx = 0, jstep = 0, it = 0, okind = 0, prev_it = 0;
    V3 p = {0, 0, 0}, d = {0, 0, 0};
    JuliaPk s = julia_pk_begin({0, 0, 0, 0});
    PkF2 cp = {0, 0}, cq = {0, 0};
    float cut = 0.0f;
    IdleWatch watch = {0u, 0ull, false, 0ull, 0u};
    long long t_mark = STATS ? clock64() : 0;
    constexpr bool TIMED = kPhaseTiming && !STATS;
    WaveClocks wc;
    enum { W_MARCH, W_TAIL, W_REFILL, W_IDLE, W_NORMAL, W_PHASES, W_LANES, W_BLOCKS, W_TAIL_LANES, W_NORMAL_PHASES, W_NORMAL_LANES, W_BUSY_LANES };
    if (TIMED) {
        const int slots[] = {D_MARCH_CYCLES, D_M_TAIL_CYCLES, D_M_REFILL_CYCLES, D_M_IDLE_CYCLES, D_NORMAL_CYCLES, D_MARCH_PHASES, D_MARCH_LANES,
                             D_DE_WAVE_ITERS, D_M_TAIL_LANES, D_NORMAL_PHASES, D_NORMAL_LANES, D_MARCH_HAS_SUM};
        wc.begin(slots, 12);
    }
    auto charge = [&](int slot) {
        if (STATS) {
            const long long now = clock64();
            if (lane == 0) n.v[slot] += static_cast<unsigned long long>(now - t_mark);
            t_mark = now;
        } else if (TIMED) {
            wc.charge(slot == D_MARCH_CYCLES ? W_MARCH : slot == D_M_TAIL_CYCLES ? W_TAIL : slot == D_M_REFILL_CYCLES ? W_REFILL : slot == D_M_IDLE_CYCLES ? W_IDLE : W_NORMAL);
        }
    };
    unsigned round = 0;
    for (;;) {
        // The rings are looked at when enough lanes are idle to make the LDS round trips worth it (or
        // every eighth round, so that no job waits long).
        const int n_has0 = __popcll(__ballot(has));
        round++;
        if (n_has0 <= 56 || (round & 7u) == 0u) {
            typedef unsigned U4 __attribute__((ext_vector_type(4)));
            const U4 ctl = *reinterpret_cast<volatile U4*>(sh.ctl);                  // march head, tail, normal head, tail
            const U4 flags = *reinterpret_cast<volatile U4*>(&sh.dead);              // dead, abort, heartbeat
            if (flags.y != 0u) break;
            const unsigned n_conv = ctl.w - ctl.z, n_jobs = ctl.y - ctl.x;
            if (STATS && lane == 0) {
                n.v[D_MARCH_POLLS]++;
                n.v[D_MARCH_HAS_SUM] += static_cast<unsigned long long>(n_has0);
            }
            // ---- finish converged jobs: a full batch at once, a partial one when there is little else to do
            if (n_conv >= 64u || (n_conv > 0u && n_has0 < RT_JULIA_NORMAL_IDLE && n_jobs == 0u)) {
                unsigned first = 0;
                const unsigned got = ring_pop(&sh.ctl[2 * R_NORMAL], &sh.ctl[2 * R_NORMAL + 1], 64u, first);
                bool to_t = false, to_s = false;
                unsigned o = 0;
                if (static_cast<unsigned>(lane) < got) {
                    o = ring_entry(sh.ring[R_NORMAL], first + lane, &sh.abort);
                    float* c = sh.ctx[o];
                    const F4 q0 = ctx_get(c, CW_QO), q1 = ctx_get(c, CW_QD), j0 = ctx_get(c, CW_JP), j1 = ctx_get(c, CW_JD);
                    const DevJulia& j = sc.julias[__float_as_uint(c[CW_JIDX])];
                    bump<STATS>(n, C_JULIA_NORMALS);
                    unsigned long long iters = 0;
                    HitRec h;
                    // the distance limits of src/objects.cpp:236 are applied by the shader wave (it knows the query)
                    const bool hit = julia_finish<STATS>(j, v3(q0.x, q0.y, q0.z), v3(j0.x, j0.y, j0.z), v3(j1.x, j1.y, j1.z), kInfinite, h, iters);
                    bump<STATS>(n, C_JULIA_DE_ITERATIONS, iters);
                    if (hit) {
                        ctx_put(c, CW_JP, F4{h.point.x, h.point.y, h.point.z, h.dist});
                        ctx_put(c, CW_JD, F4{h.normal.x, h.normal.y, h.normal.z, __uint_as_float(JOB_HIT)});
                    } else {
                        c[CW_JSTATUS] = __uint_as_float(JOB_MISS);
                    }
                    const bool shadow = (__float_as_uint(q1.w) & CF_KIND_MASK) == static_cast<unsigned>(Q_SHADOW);
                    to_s = shadow;
                    to_t = !shadow;
                }
                ring_push(sh.ring[R_READY_T], &sh.ctl[2 * R_READY_T + 1], to_t, o);
                ring_push(sh.ring[R_READY_S], &sh.ctl[2 * R_READY_S + 1], to_s, o);
                beat(sh);
                if (STATS && lane == 0) n.v[D_NORMAL_PHASES]++;
                if (STATS && static_cast<unsigned>(lane) < got) n.v[D_NORMAL_LANES]++;
                if (TIMED) {
                    wc.add(W_NORMAL_PHASES, 1);
                    wc.add(W_NORMAL_LANES, got);
                }
                charge(D_NORMAL_CYCLES);
                continue;
            }
            // ---- refill idle lanes from the march ring
            const unsigned long long need = __ballot(!has);
            if (need != 0ull && n_jobs != 0u) {
                unsigned first = 0;
                const unsigned got = ring_pop(&sh.ctl[2 * R_MARCH], &sh.ctl[2 * R_MARCH + 1], static_cast<unsigned>(__popcll(need)), first);
                const unsigned rank = __popcll(need & ((1ull << lane) - 1ull));
                if (!has && rank < got) {
                    owner = ring_entry(sh.ring[R_MARCH], first + rank, &sh.abort);
                    const float* c = sh.ctx[owner];
                    const F4 j0 = ctx_get(c, CW_JP), j1 = ctx_get(c, CW_JD);
                    jidx = __float_as_uint(c[CW_JIDX]);
                    okind = __float_as_uint(c[CW_FLAGS]) & CF_KIND_MASK;
                    p = v3(j0.x, j0.y, j0.z);
                    d = v3(j1.x, j1.y, j1.z);
                    if (jidx < static_cast<unsigned>(kJuliaConstTable)) {
                        const F4 c4 = *reinterpret_cast<const F4*>(sh.jconst[jidx]);
                        cp = {c4.x, c4.y};
                        cq = {c4.z, c4.w};
                        cut = sh.jconst[jidx][4];
                    } else {
                        const DevJulia& j = sc.julias[jidx];
                        cp = {j.c[0], j.c[1]};
                        cq = {j.c[2], j.c[3]};
                        cut = j.cut_plane;
                    }
                    jstep = 0;
                    s = julia_pk_begin({p.x, p.y, p.z, cut});
                    it = 0;
                    has = true;
                    busy = true;
                }
            }
            const int n_now = __popcll(__ballot(has));
            if (n_now == 0) {
                if (flags.x >= static_cast<unsigned>(kSampleCtx)) break;   // every sample context is dead: no job can come any more
                idle_wait(sh, watch);
                charge(D_M_IDLE_CYCLES);
                continue;
            }
        }
        const int n_has = __popcll(__ballot(has));
        watch.armed = false;
        charge(D_M_REFILL_CYCLES);
        if (STATS && lane == 0) n.v[D_MARCH_PHASES]++;
        if (STATS && has) n.v[D_MARCH_LANES]++;
        if (TIMED) {
            wc.add(W_PHASES, 1);
            wc.add(W_LANES, static_cast<unsigned>(n_has));
        }
        // ---- estimator iterations, four at a time, until enough lanes wait for their march step
        for (;;) {
            if (has && busy) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    bump<STATS>(n, C_JULIA_DE_ITERATIONS);
                    julia_pk_iterate_only(s, cp, cq);
                    // (the counters build stops where the oracle's count stops: at the overflow of d2 itself)
                    if (s.m2 > 1e10f || (STATS && s.d2 == __builtin_inff())) break;
                }
                it++;   // blocks of four
                // exact early-out once d2 has overflowed, looked at once per block: further iterations
                // cannot change the value the march uses
                if (s.m2 > 1e10f || s.d2 == __builtin_inff() || it >= kJuliaDistanceIterations / 4) busy = false;
            }
            if (STATS && lane == 0) n.v[D_DE_WAVE_ITERS] += 256ull;
            const int n_busy = __popcll(__ballot(has && busy));
            if (TIMED) {
                wc.add(W_BLOCKS, 1);                                   // estimator blocks of four passes ...
                wc.add(W_BUSY_LANES, static_cast<unsigned>(n_busy));   // ... and the lanes still busy after each
            }
            if (n_busy == 0 || (n_has - n_busy) * 8 >= n_has * RT_JULIA_STEP_EIGHTHS) break;
        }
        charge(D_MARCH_CYCLES);
        if (STATS && lane == 0) n.v[D_M_TAIL_BLOCKS]++;
        if (STATS && has && !busy) n.v[D_M_TAIL_LANES]++;
        if (TIMED) wc.add(W_TAIL_LANES, static_cast<unsigned>(__popcll(__ballot(has && !busy))));
        // ---- march steps of the lanes whose estimate is complete (src/objects.cpp:206-221)
        bool converged = false, miss_t = false, miss_s = false;
        if (has && !busy) {
            bump<STATS>(n, C_JULIA_STEPS);
            if (STATS) {
                const int b = it <= 1u ? 0 : it <= 2u ? 1 : it <= 4u ? 2 : it <= 8u ? 3 : it <= 16u ? 4 : it <= 32u ? 5 : 6;
                n.v[D_STEP_BLOCKS_1 + b]++;
                n.v[D_STEP_BLOCK_SUM] += it;
                if (jstep > 0u && (prev_it <= 4u) != (it <= 4u)) n.v[D_STEP_CLASS_SWITCH]++;
                prev_it = it;
            }
            const int r = julia_advance(julia_distance_tail_march(s.m2, s.d2), p, d, jstep);
            const int hot_level = r == 0 ? sched_level_reached(job, jstep) : -1;
            if (hot_level >= 0) {
                // a long march: the sample is hot from here on (julia_shader sends its shadow rays out in parallel), and
                // so is its tile (the tile's other samples jump the queue)
                float* c = sh.ctx[owner];
                if ((__float_as_uint(c[CW_FLAGS]) & CF_HELPER) == 0u) {
                    c[CW_MISC] = __uint_as_float(__float_as_uint(c[CW_MISC]) | kCtxHotBit);
                    if (job.sched != nullptr)
                        sched_mark_hot(sc, job, static_cast<unsigned long long>(__float_as_uint(c[CW_SLOT_LO])) | (static_cast<unsigned long long>(__float_as_uint(c[CW_SLOT_HI])) << 32),
                                       static_cast<unsigned>(hot_level));
                }
            }
            if (r == 0) {
                s = julia_pk_begin({p.x, p.y, p.z, cut});
                it = 0;
                busy = true;
            } else {
                has = false;
                float* c = sh.ctx[owner];
                if (r == 1) {
                    converged = true;
                    c[CW_JP] = p.x;
                    c[CW_JP + 1] = p.y;
                    c[CW_JP + 2] = p.z;
                } else {
                    c[CW_JSTATUS] = __uint_as_float(JOB_MISS);
                    miss_s = okind == static_cast<unsigned>(Q_SHADOW);
                    miss_t = !miss_s;
                }
            }
        }
        ring_push(sh.ring[R_NORMAL], &sh.ctl[2 * R_NORMAL + 1], converged, owner);
        ring_push(sh.ring[R_READY_T], &sh.ctl[2 * R_READY_T + 1], miss_t, owner);
        ring_push(sh.ring[R_READY_S], &sh.ctl[2 * R_READY_S + 1], miss_s, owner);
        beat(sh);
        charge(D_M_TAIL_CYCLES);
    }
    if (TIMED) wc.flush(counter_words);
}

// ---- marcher waves with two jobs per lane (RT_JULIA_DUAL, the default) ---------------------------------------------
// julia_marcher above takes the march steps of a wave's lanes together once 5/8 of them have an estimate complete:
// measured on C4, 35 of the 51 lanes that hold a job run the estimator and 37 lanes take a step, because estimates of
// one wave end one to sixteen blocks apart and a lane that is done idles until the wave's step phase.  Here a lane
// holds up to TWO jobs but only ONE estimator state:
//   E  the job being estimated (registers: the packed quaternion state);
//   Q  a second job that is either PENDING (its estimate is complete: (m2, d2) kept in two registers, its march step
//      due) or READY (its position after the step, three registers: the estimate can start at once).
// A lane whose estimate completes parks it in Q and goes on with the job that was READY there, so it idles only when
// both of its jobs wait for a step; the wave takes the steps of all PENDING jobs together once RT_JULIA_DUAL_PENDING
// lanes have one (or RT_JULIA_DUAL_STALLED_EIGHTHS of the lanes that hold a job have nothing else to do).  What a step needs besides (m2, d2)
// — position, direction, step count — stays in the job's context record in LDS (CW_JP, CW_JD; the step count in
// CW_JDIST, which is free while the job marches) and is read and written in the step phase only.
#ifndef RT_JULIA_DUAL_PENDING
#define RT_JULIA_DUAL_PENDING 48
#endif
#ifndef RT_JULIA_DUAL_STALLED_EIGHTHS
#define RT_JULIA_DUAL_STALLED_EIGHTHS 4  // ... or this many eighths of the lanes that hold a job wait for it
#endif
#ifndef RT_JULIA_DUAL_REFILL
#define RT_JULIA_DUAL_REFILL 8          // empty Q slots that make a wave look at the MARCH ring
#endif
template <bool STATS>
RT_DEV void julia_marcher_dual(const DevScene& sc, const DevJob& job, PoolShared& sh, Counters& n, unsigned long long* counter_words) {
    const int lane = threadIdx.x & 63;
    enum : unsigned { E_NONE = 0u, E_RUN = 1u, E_DONE = 2u, Q_EMPTY = 0u, Q_READY = 1u, Q_PENDING = 2u };
    unsigned e_state = E_NONE, q_state = Q_EMPTY;
    unsigned owner_e = 0, owner_q = 0;          // context id | Julia set index << 16
    unsigned it = 0, q_it = 0, cjidx = ~0u;
    float qa = 0.0f, qb = 0.0f, qc = 0.0f;      // PENDING: m2, d2;  READY: the position
    JuliaPk s = julia_pk_begin({0, 0, 0, 0});
    PkF2 cp = {0, 0}, cq = {0, 0};
    float cut = 0.0f;
    IdleWatch watch = {0u, 0ull, false, 0ull, 0u};
    long long t_mark = STATS ? clock64() : 0;
    constexpr bool TIMED = kPhaseTiming && !STATS;
    WaveClocks wc;
    enum { W_MARCH, W_TAIL, W_REFILL, W_IDLE, W_NORMAL, W_PHASES, W_LANES, W_BLOCKS, W_TAIL_LANES, W_NORMAL_PHASES, W_NORMAL_LANES, W_BUSY_LANES };
    if (TIMED) {
        const int slots[] = {D_MARCH_CYCLES, D_M_TAIL_CYCLES, D_M_REFILL_CYCLES, D_M_IDLE_CYCLES, D_NORMAL_CYCLES, D_MARCH_PHASES, D_MARCH_LANES,
                             D_DE_WAVE_ITERS, D_M_TAIL_LANES, D_NORMAL_PHASES, D_NORMAL_LANES, D_MARCH_HAS_SUM};
        wc.begin(slots, 12);
    }
    auto charge = [&](int slot) {
        if (STATS) {
            const long long now = clock64();
            if (lane == 0) n.v[slot] += static_cast<unsigned long long>(now - t_mark);
            t_mark = now;
        } else if (TIMED) {
            wc.charge(slot == D_MARCH_CYCLES ? W_MARCH : slot == D_M_TAIL_CYCLES ? W_TAIL : slot == D_M_REFILL_CYCLES ? W_REFILL : slot == D_M_IDLE_CYCLES ? W_IDLE : W_NORMAL);
        }
    };
    auto load_constants = [&](unsigned jidx) {
        if (jidx < static_cast<unsigned>(kJuliaConstTable)) {
            const F4 c4 = *reinterpret_cast<const F4*>(sh.jconst[jidx]);
            cp = {c4.x, c4.y};
            cq = {c4.z, c4.w};
            cut = sh.jconst[jidx][4];
        } else {
            const DevJulia& j = sc.julias[jidx];
            cp = {j.c[0], j.c[1]};
            cq = {j.c[2], j.c[3]};
            cut = j.cut_plane;
        }
        cjidx = jidx;
    };
    unsigned round = 0;
    for (;;) {
        const int n_empty = __popcll(__ballot(q_state == Q_EMPTY));
        round++;
        if (n_empty >= RT_JULIA_DUAL_REFILL || (round & 7u) == 0u) {
            typedef unsigned U4 __attribute__((ext_vector_type(4)));
            const U4 ctl = *reinterpret_cast<volatile U4*>(sh.ctl);                  // march head, tail, normal head, tail
            const U4 flags = *reinterpret_cast<volatile U4*>(&sh.dead);              // dead, abort, heartbeat
            if (flags.y != 0u) break;
            const unsigned n_conv = ctl.w - ctl.z, n_jobs = ctl.y - ctl.x;
            const int n_held = __popcll(__ballot(e_state != E_NONE)) + (64 - n_empty);
            if (STATS && lane == 0) {
                n.v[D_MARCH_POLLS]++;
                n.v[D_MARCH_HAS_SUM] += static_cast<unsigned long long>(n_held);
            }
            // ---- finish converged jobs: a full batch at once, a partial one when there is little else to do
            if (n_conv >= 64u || (n_conv > 0u && n_held < RT_JULIA_NORMAL_IDLE && n_jobs == 0u)) {
                unsigned first = 0;
                const unsigned got = ring_pop(&sh.ctl[2 * R_NORMAL], &sh.ctl[2 * R_NORMAL + 1], 64u, first);
                bool to_t = false, to_s = false;
                unsigned o = 0;
                if (static_cast<unsigned>(lane) < got) {
                    o = ring_entry(sh.ring[R_NORMAL], first + lane, &sh.abort);
                    float* c = sh.ctx[o];
                    const F4 q0 = ctx_get(c, CW_QO), q1 = ctx_get(c, CW_QD), j0 = ctx_get(c, CW_JP), j1 = ctx_get(c, CW_JD);
                    const DevJulia& j = sc.julias[__float_as_uint(c[CW_JIDX])];
                    bump<STATS>(n, C_JULIA_NORMALS);
                    unsigned long long iters = 0;
                    HitRec h;
                    // the distance limits of src/objects.cpp:236 are applied by the shader wave (it knows the query)
                    const bool hit = julia_finish<STATS>(j, v3(q0.x, q0.y, q0.z), v3(j0.x, j0.y, j0.z), v3(j1.x, j1.y, j1.z), kInfinite, h, iters);
                    bump<STATS>(n, C_JULIA_DE_ITERATIONS, iters);
                    if (hit) {
                        ctx_put(c, CW_JP, F4{h.point.x, h.point.y, h.point.z, h.dist});
                        ctx_put(c, CW_JD, F4{h.normal.x, h.normal.y, h.normal.z, __uint_as_float(JOB_HIT)});
                    } else {
                        c[CW_JSTATUS] = __uint_as_float(JOB_MISS);
                    }
                    const bool shadow = (__float_as_uint(q1.w) & CF_KIND_MASK) == static_cast<unsigned>(Q_SHADOW);
                    to_s = shadow;
                    to_t = !shadow;
                }
                ring_push(sh.ring[R_READY_T], &sh.ctl[2 * R_READY_T + 1], to_t, o);
                ring_push(sh.ring[R_READY_S], &sh.ctl[2 * R_READY_S + 1], to_s, o);
                beat(sh);
                if (STATS && lane == 0) n.v[D_NORMAL_PHASES]++;
                if (STATS && static_cast<unsigned>(lane) < got) n.v[D_NORMAL_LANES]++;
                if (TIMED) {
                    wc.add(W_NORMAL_PHASES, 1);
                    wc.add(W_NORMAL_LANES, got);
                }
                charge(D_NORMAL_CYCLES);
                continue;
            }
            // ---- new jobs from the march ring into empty Q slots: READY at once (the shader wave left the entry position)
            const unsigned long long need = __ballot(q_state == Q_EMPTY);
            if (need != 0ull && n_jobs != 0u) {
                unsigned first = 0;
                const unsigned got = ring_pop(&sh.ctl[2 * R_MARCH], &sh.ctl[2 * R_MARCH + 1], static_cast<unsigned>(__popcll(need)), first);
                const unsigned rank = __popcll(need & ((1ull << lane) - 1ull));
                if (q_state == Q_EMPTY && rank < got) {
                    const unsigned o = ring_entry(sh.ring[R_MARCH], first + rank, &sh.abort);
                    const float* c = sh.ctx[o];
                    const F4 j0 = ctx_get(c, CW_JP);
                    owner_q = o | (__float_as_uint(c[CW_JIDX]) << 16);
                    qa = j0.x;
                    qb = j0.y;
                    qc = j0.z;
                    q_state = Q_READY;
                }
            }
            if (!__any(e_state != E_NONE || q_state != Q_EMPTY)) {
                if (flags.x >= static_cast<unsigned>(kSampleCtx)) break;   // every sample context is dead: no job can come any more
                idle_wait(sh, watch);
                charge(D_M_IDLE_CYCLES);
                continue;
            }
        }
        watch.armed = false;
        // ---- a complete estimate moves to Q (PENDING) unless Q holds one already; a READY job becomes the lane's estimate
        {
            const bool start = e_state != E_RUN && q_state == Q_READY;
            const bool park = e_state == E_DONE && q_state != Q_PENDING;
            const unsigned was_e = owner_e, was_it = it;
            const float m2 = s.m2, d2 = s.d2;
            if (__any(start)) {
                if (__any(start && (owner_q >> 16) != cjidx)) {
                    if (start && (owner_q >> 16) != cjidx) load_constants(owner_q >> 16);
                }
                if (start) {
                    owner_e = owner_q;
                    s = julia_pk_begin({qa, qb, qc, cut});
                    it = 0;
                    e_state = E_RUN;
                    q_state = Q_EMPTY;
                }
            }
            if (park) {
                owner_q = was_e;
                qa = m2;
                qb = d2;
                q_it = was_it;
                q_state = Q_PENDING;
                if (!start) e_state = E_NONE;
            }
        }
        const int n_run = __popcll(__ballot(e_state == E_RUN)), n_pend = __popcll(__ballot(q_state == Q_PENDING));
        const int n_stalled = __popcll(__ballot(e_state != E_RUN && q_state == Q_PENDING));   // lanes that wait for the step phase
        charge(D_M_REFILL_CYCLES);
        if (n_pend >= RT_JULIA_DUAL_PENDING || (n_pend > 0 && n_stalled * 8 >= (n_run + n_stalled) * RT_JULIA_DUAL_STALLED_EIGHTHS)) {
            // ---- march steps of the PENDING jobs (src/objects.cpp:206-221)
            if (STATS && lane == 0) n.v[D_M_TAIL_BLOCKS]++;
            if (STATS && q_state == Q_PENDING) n.v[D_M_TAIL_LANES]++;
            if (TIMED) {
                wc.add(W_PHASES, 1);
                wc.add(W_TAIL_LANES, static_cast<unsigned>(n_pend));
            }
            if (STATS && lane == 0) n.v[D_MARCH_PHASES]++;
            bool converged = false, miss_t = false, miss_s = false;
            const unsigned o = owner_q & 0xffffu;
            if (q_state == Q_PENDING) {
                bump<STATS>(n, C_JULIA_STEPS);
                if (STATS) {
                    const int b = q_it <= 1u ? 0 : q_it <= 2u ? 1 : q_it <= 4u ? 2 : q_it <= 8u ? 3 : q_it <= 16u ? 4 : q_it <= 32u ? 5 : 6;
                    n.v[D_STEP_BLOCKS_1 + b]++;
                    n.v[D_STEP_BLOCK_SUM] += q_it;
                }
                float* c = sh.ctx[o];
                const F4 j0 = ctx_get(c, CW_JP), j1 = ctx_get(c, CW_JD);
                V3 p = v3(j0.x, j0.y, j0.z);
                unsigned jstep = __float_as_uint(j0.w);
                const int r = julia_advance(julia_distance_tail_march(qa, qb), p, v3(j1.x, j1.y, j1.z), jstep);
                const int hot_level = r == 0 ? sched_level_reached(job, jstep) : -1;
                if (hot_level >= 0 && (__float_as_uint(c[CW_FLAGS]) & CF_HELPER) == 0u) {
                    c[CW_MISC] = __uint_as_float(__float_as_uint(c[CW_MISC]) | kCtxHotBit);
                    if (job.sched != nullptr)
                        sched_mark_hot(sc, job, static_cast<unsigned long long>(__float_as_uint(c[CW_SLOT_LO])) | (static_cast<unsigned long long>(__float_as_uint(c[CW_SLOT_HI])) << 32),
                                       static_cast<unsigned>(hot_level));
                }
                if (r == 0) {
                    ctx_put(c, CW_JP, F4{p.x, p.y, p.z, __uint_as_float(jstep)});
                    qa = p.x;
                    qb = p.y;
                    qc = p.z;
                    q_state = Q_READY;
                } else {
                    q_state = Q_EMPTY;
                    if (r == 1) {
                        converged = true;
                        c[CW_JP] = p.x;
                        c[CW_JP + 1] = p.y;
                        c[CW_JP + 2] = p.z;
                    } else {
                        c[CW_JSTATUS] = __uint_as_float(JOB_MISS);
                        miss_s = (__float_as_uint(c[CW_FLAGS]) & CF_KIND_MASK) == static_cast<unsigned>(Q_SHADOW);
                        miss_t = !miss_s;
                    }
                }
            }
            ring_push(sh.ring[R_NORMAL], &sh.ctl[2 * R_NORMAL + 1], converged, o);
            ring_push(sh.ring[R_READY_T], &sh.ctl[2 * R_READY_T + 1], miss_t, o);
            ring_push(sh.ring[R_READY_S], &sh.ctl[2 * R_READY_S + 1], miss_s, o);
            beat(sh);
            charge(D_M_TAIL_CYCLES);
            continue;
        }
        if (n_run == 0) continue;     // (nothing to estimate and no step due: only READY jobs of lanes that start them next round)
        // ---- one estimator block of four passes
        if (e_state == E_RUN) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                bump<STATS>(n, C_JULIA_DE_ITERATIONS);
                julia_pk_iterate_only(s, cp, cq);
                // (the counters build stops where the oracle's count stops: at the overflow of d2 itself)
                if (s.m2 > 1e10f || (STATS && s.d2 == __builtin_inff())) break;
            }
            it++;   // blocks of four
            // exact early-out once d2 has overflowed, looked at once per block: further iterations
            // cannot change the value the march uses
            if (s.m2 > 1e10f || s.d2 == __builtin_inff() || it >= kJuliaDistanceIterations / 4) e_state = E_DONE;
        }
        if (STATS && lane == 0) n.v[D_DE_WAVE_ITERS] += 256ull;
        if (STATS && e_state != E_NONE) n.v[D_MARCH_LANES]++;
        if (TIMED) {
            wc.add(W_BLOCKS, 1);
            wc.add(W_BUSY_LANES, static_cast<unsigned>(n_run));
            wc.add(W_LANES, static_cast<unsigned>(__popcll(__ballot(e_state != E_NONE || q_state != Q_EMPTY))));
        }
        charge(D_MARCH_CYCLES);
    }
    if (TIMED) wc.flush(counter_words);
}

#if RT_JULIA_TAIL_RING
// ---- marcher waves, march steps pooled through the TAIL ring (RT_JULIA_TAIL_RING) -----------------------------------
// julia_marcher above takes the march steps of its own lanes, so a lane whose estimate is complete idles until enough
// of its wave's lanes are (measured on C4: 37 of 64 lanes per step phase, a third of the estimator's lane slots
// empty).  Here a marcher lane owns nothing but an estimate: a finished one leaves (m2, d2) in its context and the
// context on the TAIL ring, and the lane takes the next job at once; any marcher wave that sees 64 entries on TAIL
// takes their march steps at full width and sends each context back to MARCH (the position advanced, the step count
// in the flags word), on to NORMAL (converged) or to a READY ring (missed).
constexpr int CW_D2 = CW_UNION + 19;       // free in both uses of the union words
constexpr unsigned kStepShift = 8u;        // march step index in the flags word above the CtxFlag bits
#ifndef RT_JULIA_PARTIAL_TAIL_HAS
#define RT_JULIA_PARTIAL_TAIL_HAS 48      // a partial batch of march steps is taken by a wave with at most this many estimates in hand, when MARCH is empty
#endif
#ifndef RT_JULIA_RETIRE_EIGHTHS
#define RT_JULIA_RETIRE_EIGHTHS 2          // finished estimates are retired once this many eighths of a wave's lanes hold one
#endif
template <bool STATS>
RT_DEV void julia_marcher_pooled(const DevScene& sc, PoolShared& sh, Counters& n) {
    const int lane = threadIdx.x & 63;
    bool has = false, busy = false;
    unsigned owner = 0, it = 0;
    JuliaPk s = julia_pk_begin({0, 0, 0, 0});
    PkF2 cp = {0, 0}, cq = {0, 0};
    IdleWatch watch = {0u, 0ull, false, 0ull, 0u};
    long long t_mark = STATS ? clock64() : 0;
    auto charge = [&](int slot) {
        if (STATS) {
            const long long now = clock64();
            if (lane == 0) n.v[slot] += static_cast<unsigned long long>(now - t_mark);
            t_mark = now;
        }
    };
    for (;;) {
        typedef unsigned U4 __attribute__((ext_vector_type(4)));
        const U4 ctl = *reinterpret_cast<volatile U4*>(sh.ctl);                  // march head, tail, normal head, tail
        const U4 ctl_t = *reinterpret_cast<volatile U4*>(sh.ctl + 2 * R_TAIL);   // tail head, tail
        const U4 flags = *reinterpret_cast<volatile U4*>(&sh.dead);              // dead, abort, heartbeat
        if (flags.y != 0u) break;
        const int n_has0 = __popcll(__ballot(has));
        const unsigned n_conv = ctl.w - ctl.z, n_jobs = ctl.y - ctl.x, n_tail = ctl_t.y - ctl_t.x;
        if (STATS && lane == 0) {
            n.v[D_MARCH_POLLS]++;
            n.v[D_MARCH_HAS_SUM] += static_cast<unsigned long long>(n_has0);
        }
        // ---- finish converged jobs: a full batch at once, a partial one when there is little else to do
        if (n_conv >= 64u || (n_conv > 0u && n_has0 < RT_JULIA_NORMAL_IDLE && n_jobs == 0u && n_tail == 0u)) {
            unsigned first = 0;
            const unsigned got = ring_pop(&sh.ctl[2 * R_NORMAL], &sh.ctl[2 * R_NORMAL + 1], 64u, first);
            bool to_t = false, to_s = false;
            unsigned o = 0;
            if (static_cast<unsigned>(lane) < got) {
                o = ring_entry(sh.ring[R_NORMAL], first + lane, &sh.abort);
                float* c = sh.ctx[o];
                const F4 q0 = ctx_get(c, CW_QO), q1 = ctx_get(c, CW_QD), j0 = ctx_get(c, CW_JP), j1 = ctx_get(c, CW_JD);
                const DevJulia& j = sc.julias[__float_as_uint(c[CW_JIDX])];
                bump<STATS>(n, C_JULIA_NORMALS);
                unsigned long long iters = 0;
                HitRec h;
                // the distance limits of src/objects.cpp:236 are applied by the shader wave (it knows the query)
                const bool hit = julia_finish<STATS>(j, v3(q0.x, q0.y, q0.z), v3(j0.x, j0.y, j0.z), v3(j1.x, j1.y, j1.z), kInfinite, h, iters);
                bump<STATS>(n, C_JULIA_DE_ITERATIONS, iters);
                if (hit) {
                    ctx_put(c, CW_JP, F4{h.point.x, h.point.y, h.point.z, h.dist});
                    ctx_put(c, CW_JD, F4{h.normal.x, h.normal.y, h.normal.z, __uint_as_float(JOB_HIT)});
                } else {
                    c[CW_JSTATUS] = __uint_as_float(JOB_MISS);
                }
                const bool shadow = (__float_as_uint(q1.w) & CF_KIND_MASK) == static_cast<unsigned>(Q_SHADOW);
                to_s = shadow;
                to_t = !shadow;
            }
            ring_push(sh.ring[R_READY_T], &sh.ctl[2 * R_READY_T + 1], to_t, o);
            ring_push(sh.ring[R_READY_S], &sh.ctl[2 * R_READY_S + 1], to_s, o);
            beat(sh);
            if (STATS && lane == 0) n.v[D_NORMAL_PHASES]++;
            if (STATS && static_cast<unsigned>(lane) < got) n.v[D_NORMAL_LANES]++;
            charge(D_NORMAL_CYCLES);
            continue;
        }
        // ---- march steps (src/objects.cpp:206-221) of a full batch of complete estimates; of a partial one when the
        // MARCH ring is empty and this wave has room (whatever is on TAIL is then what the idle lanes wait for)
        if (n_tail >= 64u || (n_tail > 0u && n_jobs == 0u && n_has0 <= RT_JULIA_PARTIAL_TAIL_HAS)) {
            unsigned first = 0;
            const unsigned got = ring_pop(&sh.ctl[2 * R_TAIL], &sh.ctl[2 * R_TAIL + 1], 64u, first);
            bool again = false, converged = false, miss_t = false, miss_s = false;
            unsigned o = 0;
            if (static_cast<unsigned>(lane) < got) {
                o = ring_entry(sh.ring[R_TAIL], first + lane, &sh.abort);
                float* c = sh.ctx[o];
                const F4 j0 = ctx_get(c, CW_JP), j1 = ctx_get(c, CW_JD);      // position and m2; direction and status
                const float d2 = c[CW_D2];
                const unsigned fl = __float_as_uint(c[CW_FLAGS]);
                unsigned jstep = fl >> kStepShift;
                V3 p = v3(j0.x, j0.y, j0.z);
                bump<STATS>(n, C_JULIA_STEPS);
                const int r = julia_advance(julia_distance_tail_march(j0.w, d2), p, v3(j1.x, j1.y, j1.z), jstep);
                if (r == 2) {
                    c[CW_JSTATUS] = __uint_as_float(JOB_MISS);
                    miss_s = (fl & CF_KIND_MASK) == static_cast<unsigned>(Q_SHADOW);
                    miss_t = !miss_s;
                } else {
                    c[CW_JP] = p.x;
                    c[CW_JP + 1] = p.y;
                    c[CW_JP + 2] = p.z;
                    if (r == 0) c[CW_FLAGS] = __uint_as_float((fl & ((1u << kStepShift) - 1u)) | (jstep << kStepShift));
                    again = r == 0;
                    converged = r == 1;
                }
            }
            ring_push(sh.ring[R_MARCH], &sh.ctl[2 * R_MARCH + 1], again, o);
            ring_push(sh.ring[R_NORMAL], &sh.ctl[2 * R_NORMAL + 1], converged, o);
            ring_push(sh.ring[R_READY_T], &sh.ctl[2 * R_READY_T + 1], miss_t, o);
            ring_push(sh.ring[R_READY_S], &sh.ctl[2 * R_READY_S + 1], miss_s, o);
            beat(sh);
            if (STATS && lane == 0) n.v[D_M_TAIL_BLOCKS]++;
            if (STATS && static_cast<unsigned>(lane) < got) n.v[D_M_TAIL_LANES]++;
            charge(D_M_TAIL_CYCLES);
            continue;
        }
        // ---- refill idle lanes from the march ring
        const unsigned long long need = __ballot(!has);
        if (need != 0ull && n_jobs != 0u) {
            unsigned first = 0;
            const unsigned got = ring_pop(&sh.ctl[2 * R_MARCH], &sh.ctl[2 * R_MARCH + 1], static_cast<unsigned>(__popcll(need)), first);
            const unsigned rank = __popcll(need & ((1ull << lane) - 1ull));
            if (!has && rank < got) {
                owner = ring_entry(sh.ring[R_MARCH], first + rank, &sh.abort);
                const float* c = sh.ctx[owner];
                const F4 j0 = ctx_get(c, CW_JP);
                const unsigned jidx = __float_as_uint(c[CW_JIDX]);
                float cut;
                if (jidx < static_cast<unsigned>(kJuliaConstTable)) {
                    const F4 c4 = *reinterpret_cast<const F4*>(sh.jconst[jidx]);
                    cp = {c4.x, c4.y};
                    cq = {c4.z, c4.w};
                    cut = sh.jconst[jidx][4];
                } else {
                    const DevJulia& j = sc.julias[jidx];
                    cp = {j.c[0], j.c[1]};
                    cq = {j.c[2], j.c[3]};
                    cut = j.cut_plane;
                }
                s = julia_pk_begin({j0.x, j0.y, j0.z, cut});
                it = 0;
                has = true;
                busy = true;
            }
        }
        const int n_has = __popcll(__ballot(has));
        if (n_has == 0) {
            if (flags.x >= static_cast<unsigned>(kSampleCtx)) break;   // every sample context is dead: no job can come any more
            idle_wait(sh, watch);
            charge(D_M_IDLE_CYCLES);
            continue;
        }
        watch.armed = false;
        charge(D_M_REFILL_CYCLES);
        if (STATS && lane == 0) n.v[D_MARCH_PHASES]++;
        if (STATS && has) n.v[D_MARCH_LANES]++;
        // ---- estimator iterations, four at a time, until enough lanes hold a complete estimate (or, with lanes to
        // spare, for four blocks: new jobs may have come)
        for (int blocks = 1;; blocks++) {
            if (busy) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    bump<STATS>(n, C_JULIA_DE_ITERATIONS);
                    julia_pk_iterate_only(s, cp, cq);
                    // (the counters build stops where the oracle's count stops: at the overflow of d2 itself)
                    if (s.m2 > 1e10f || (STATS && s.d2 == __builtin_inff())) break;
                }
                it++;   // blocks of four
                // exact early-out once d2 has overflowed, looked at once per block: further iterations
                // cannot change the value the march uses
                if (s.m2 > 1e10f || s.d2 == __builtin_inff() || it >= kJuliaDistanceIterations / 4) busy = false;
            }
            if (STATS && lane == 0) n.v[D_DE_WAVE_ITERS] += 256ull;
            const int n_busy = __popcll(__ballot(busy));
            if (n_busy == 0 || (n_has - n_busy) * 8 >= 64 * RT_JULIA_RETIRE_EIGHTHS || (n_has < 56 && blocks >= 4)) break;
        }
        charge(D_MARCH_CYCLES);
        // ---- retire the complete estimates
        const bool retire = has && !busy;
        if (retire) {
            float* c = sh.ctx[owner];
            c[CW_JDIST] = s.m2;
            c[CW_D2] = s.d2;
            has = false;
            if (STATS) {
                const int b = it <= 1u ? 0 : it <= 2u ? 1 : it <= 4u ? 2 : it <= 8u ? 3 : it <= 16u ? 4 : it <= 32u ? 5 : 6;
                n.v[D_STEP_BLOCKS_1 + b]++;
                n.v[D_STEP_BLOCK_SUM] += it;
            }
        }
        ring_push(sh.ring[R_TAIL], &sh.ctl[2 * R_TAIL + 1], retire, owner);
        beat(sh);
    }
}
#endif

// ---- shader waves -------------------------------------------------------------------------------------------
template <bool STATS>
RT_DEV void julia_shader(const DevScene& sc, const ShadeTables& T, const DevJob& job, PoolShared& sh, Counters& n, WaveIds& wave_ids) {
    const int lane = threadIdx.x & 63;
    IdleWatch watch = {0u, 0ull, false, 0ull, 0u};
    unsigned patience = 0;
    long long t_mark = STATS ? clock64() : 0;
    constexpr bool TIMED = kPhaseTiming && !STATS;
    WaveClocks wc;
    enum { W_RESOLVE, W_QUERY, W_WAIT, W_RPHASES, W_RLANES, W_QPHASES, W_QLANES };
    if (TIMED) {
        const int slots[] = {D_RESOLVE_CYCLES, D_QUERY_CYCLES, D_T_WAIT_CYCLES, D_RESOLVE_PHASES, D_RESOLVE_LANES, D_QUERY_PHASES, D_QUERY_LANES};
        wc.begin(slots, 7);
    }
    auto charge = [&](int slot) {
        if (STATS) {
            const long long now = clock64();
            if (lane == 0) n.v[slot] += static_cast<unsigned long long>(now - t_mark);
            t_mark = now;
        } else if (TIMED) {
            wc.charge(slot == D_RESOLVE_CYCLES ? W_RESOLVE : slot == D_QUERY_CYCLES ? W_QUERY : W_WAIT);
        }
    };
    float* const wg_frames = job.stack + static_cast<size_t>(blockIdx.x) * kPoolCtx * (sc.max_bounces + 1) * kFrameRecordWords;
    for (;;) {
        typedef unsigned U4 __attribute__((ext_vector_type(4)));
        const U4 ctl_m = *reinterpret_cast<volatile U4*>(sh.ctl);             // march head, tail, normal head, tail
        const U4 ctl_r = *reinterpret_cast<volatile U4*>(sh.ctl + 4);         // ready_t head, tail, ready_s head, tail
        const U4 flags = *reinterpret_cast<volatile U4*>(&sh.dead);           // dead, abort, heartbeat
        if (flags.y != 0u || flags.x >= static_cast<unsigned>(kSampleCtx)) break;
        const unsigned n_t = ctl_r.y - ctl_r.x, n_s = ctl_r.w - ctl_r.z, n_jobs = ctl_m.y - ctl_m.x;
        if (STATS && lane == 0) {
            n.v[D_RING_SAMPLES]++;
            n.v[D_RING_MARCH_SUM] += n_jobs;
            n.v[D_RING_NORMAL_SUM] += ctl_m.w - ctl_m.z;
            n.v[D_RING_READY_T_SUM] += n_t;
            n.v[D_RING_READY_S_SUM] += n_s;
        }
        const int r = n_t >= n_s ? R_READY_T : R_READY_S;
        const unsigned n_ready = n_t >= n_s ? n_t : n_s;
        // a full batch at once; a partial one when the marchers are about to run out of jobs, or after a while
        if (n_ready == 0u || (n_ready < static_cast<unsigned>(RT_JULIA_SHADER_MIN_BATCH) && n_jobs >= RT_JULIA_JOBS_AHEAD && patience < 16u)) {
            if (n_ready == 0u) idle_wait(sh, watch); else __builtin_amdgcn_s_sleep(4);
            patience++;
            charge(D_T_WAIT_CYCLES);
            continue;
        }
        unsigned first = 0;
        const unsigned got = ring_pop(&sh.ctl[2 * r], &sh.ctl[2 * r + 1], 64u, first);
        if (got == 0u) continue;
        patience = 0;
        watch.armed = false;
#if RT_JULIA_SHADER_PRIO
        __builtin_amdgcn_s_setprio(RT_JULIA_SHADER_PRIO);     // a batch in hand: the marchers on this SIMD wait for what it produces
#endif
        const bool mine = static_cast<unsigned>(lane) < got;
        Lane L;
        L.state = S_IDLE;
        L.qkind = -1;
        L.sp = 0;
        L.jidx = 0;
        L.jstep = 0;
        L.qactive = 0u;
        L.hot = 0u;
        L.li = 0;
        unsigned id = 0, jstatus = JOB_PENDING, cflags = 0u;
        bool fresh = false;
        HitRec jhit = {};
        if (mine) {
            id = ring_entry(sh.ring[r], first + lane, &sh.abort);
            ctx_load(sc, sh.ctx[id], L, fresh, jstatus, jhit, cflags);
        }
        // a helper context: one shadow ray of another context's node (see "shadow rays in parallel" below)
        const bool helper = mine && (cflags & CF_HELPER) != 0u;
        if (helper && (cflags & CF_NEWQUERY) != 0u) L.state = S_QUERY;
        if (STATS) {
            if (lane == 0) n.v[D_RESOLVE_PHASES]++;
            if (mine) n.v[D_RESOLVE_LANES]++;
        }
        if (TIMED) {
            wc.add(W_RPHASES, 1);
            wc.add(W_RLANES, got);
        }
        // ---- a finished Julia job: Julia::intersect's last checks and the acceptance rule; then the scene's next set
        bool push_job = false;
        if (mine && L.state == S_WAIT) {
            const DevJulia& j = sc.julias[L.jidx];
            if (jstatus == JOB_HIT) {
                // Julia::intersect ignores max_distance until this final check (src/objects.cpp:236)
                const float limit = L.qkind == Q_SHADOW ? L.qmax : kInfinite;
                if (!(jhit.dist > limit)) accept_julia_hit(sc, L, jhit, j, T.materials[j.material]);
            }
            L.state = S_RESOLVE;
            if (L.qactive) {
                for (unsigned k = L.jidx + 1; k < sc.n_julias; k++) {
                    bump<STATS>(n, C_JULIA_QUERIES);
                    if (julia_enter(sc.julias[k], L.qo, L.qd, L.jp, L.jd)) {
                        L.jidx = k;
                        L.state = S_WAIT;
                        push_job = true;
                        break;
                    }
                }
            }
        }
        // ---- resolve: one step of the ray tree; a finished sample is stored and the context takes the next one.
        // Shadow rays in parallel: a sample is a chain of dependent queries, and the longest chains set the frame's
        // time (fetch_ids_scheduled explains).  A HOT sample — one of its marches has reached DevJob::hot_steps — therefore
        // sends the shadow rays of a node's lights out together: the first one is its own query as always, each further
        // lit light's ray goes to a free HELPER context, which takes it through the same pipeline and leaves the answer
        // (blocked or not) in the node's join word.  The node still adds its lights in the reference's order: when it
        // comes to a light whose ray a helper has, it takes the answer if it is there (and goes on at once) or parks
        // until the helper wakes it.  Lights are sent out only if resolve_lane will ask for them (same elision rule),
        // so every helper's answer is consumed before the node moves on.
        bool done = false;
        Col value = {0.0f, 0.0f, 0.0f, 0.0f};
        const FramesInHbm frames = {wg_frames + static_cast<size_t>(id) * (sc.max_bounces + 1) * kFrameRecordWords};
        float* const my_ctx = sh.ctx[mine ? id : 0u];
        unsigned* const my_join = reinterpret_cast<unsigned*>(my_ctx + CW_JOIN);
        bool again = mine && L.state == S_RESOLVE && !fresh && !helper;
        bool first_pass = true;
        while (__any(again)) {
            bool fork = false;
            if (again) {
                again = false;
                const bool was_trace = L.qkind == Q_TRACE;
                resolve_lane<STATS>(sc, T, job, frames, L, n, done, value);
                if (L.state == S_QUERY && L.qkind == Q_SHADOW) {
                    if (was_trace) {
                        fork = !STATS && kHelperCtx > 0 && L.hot != 0u && L.li + 1u < sc.n_lights;
                        lds_store(my_join, 0u);        // (no helper holds a ray of this context: the last node consumed them all)
                    } else if (L.li < kMaxForkLights) {
                        const unsigned jw = lds_load(my_join);
                        if ((jw >> L.li) & 1u) {
                            if ((jw >> (8u + L.li)) & 1u) {
                                L.blocked = (jw >> (16u + L.li)) & 1u;
                                L.qactive = 0u;
                                L.state = S_RESOLVE;
                                again = true;
                            } else {
                                L.state = S_PARK;
                            }
                        }
                    }
                }
            }
            if (first_pass && __any(fork)) {
                // (only a trace result can open a node, so every fork of this batch happens in the first pass)
                const DevMaterial& m = T.materials[T.object_material[fork ? L.n_obj : 0u]];
                const bool may_elide = sc.lights_benign != 0u && job.elide_unlit != 0u && fork && m.phong_tame != 0u;
                const unsigned last = sc.n_lights < kMaxForkLights ? sc.n_lights : kMaxForkLights;
                for (unsigned j = 1; j < last; j++) {
                    bool want = fork && j > L.li;
                    if (want && may_elide) want = skip_dark_lights<false>(T.lights, j + 1u, j, L.n_point, L.n_normal, L.n_dir, m.kd, m.ks, m.exponent, true, n) == j;
                    unsigned h = ~0u;
                    if (want) {
                        // a free helper context (bit set = free), any word
                        for (unsigned t = 0; t < 2u * (kHelperCtx / 32) && h == ~0u; t++) {
                            const unsigned w = (static_cast<unsigned>(lane) + t) % (kHelperCtx / 32 > 0 ? kHelperCtx / 32 : 1);
                            const unsigned v = lds_load(&sh.helper_free[w]);
                            if (v == 0u) continue;
                            const unsigned bit = v & (0u - v);
                            const unsigned old = __hip_atomic_fetch_and(&sh.helper_free[w], ~bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if ((old & bit) != 0u) h = static_cast<unsigned>(kSampleCtx) + w * 32u + static_cast<unsigned>(__ffs(static_cast<int>(bit)) - 1);
                        }
                    }
                    if (h != ~0u) {
                        Lane ray = L;
                        ray.li = j;
                        begin_shadow(T, ray);
                        float* hc = sh.ctx[h];
                        ctx_put(hc, CW_QO, F4{ray.qo.x, ray.qo.y, ray.qo.z, ray.qmax});
                        ctx_put(hc, CW_QD, F4{ray.qd.x, ray.qd.y, ray.qd.z, __uint_as_float(static_cast<unsigned>(Q_SHADOW) | CF_QACTIVE | CF_HELPER | CF_NEWQUERY)});
                        hc[CW_JSTATUS] = __uint_as_float(JOB_PENDING);
                        hc[CW_SLOT_LO] = __uint_as_float(id | (j << 16));
                        hc[CW_SLOT_HI] = __uint_as_float(0u);
                        hc[CW_MISC] = __uint_as_float(0u);
                        hc[CW_JIDX] = __uint_as_float(0u);
                        // the join word says so before the helper can be seen by anyone (ring_push releases)
                        __hip_atomic_fetch_or(my_join, 1u << j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    ring_push(sh.ring[R_READY_S], &sh.ctl[2 * R_READY_S + 1], h != ~0u, h);
                }
            }
            first_pass = false;
        }
        if (mine && done) store_sample(sc, job, L, value);
        bool need = mine && (fresh || done);
        bool died = false;
        while (__any(need)) {
            unsigned long long sid = 0;
            const bool got_id = job.sched != nullptr ? fetch_ids_scheduled(job, &sh.hand, &sh.abort, need, sid)
                                                     : fetch_ids(job.queue_head, job.dyn_base, job.n_samples, job.id_chunk, wave_ids, need, sid);
            if (need) {
                if (!got_id) {
                    need = false;       // the queue is exhausted: this context is done for the launch
                    died = true;
                    L.state = S_IDLE;
                } else if (start_sample(sc, job, sid, L)) {
                    need = false;
                }
            }
        }
        charge(D_RESOLVE_CYCLES);
        // a helper whose query is complete leaves its answer with the node and is free again
        const bool deliver = helper && L.state == S_RESOLVE;
        bool wake = false;
        unsigned parent = 0;
        if (deliver) {
            const unsigned who = __float_as_uint(my_ctx[CW_SLOT_LO]);
            parent = who & 0xffffu;
            const unsigned j = who >> 16;
            float* pc = sh.ctx[parent];
            const unsigned bits = (1u << (8u + j)) | (L.blocked != 0u ? 1u << (16u + j) : 0u);
            const unsigned old = __hip_atomic_fetch_or(reinterpret_cast<unsigned*>(pc + CW_JOIN), bits, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            wake = ((old >> (24u + j)) & 1u) != 0u;
            // (a parked node is nobody's but the helper's that wakes it: its flags word takes the answer)
            if (wake && L.blocked != 0u) pc[CW_FLAGS] = __uint_as_float(__float_as_uint(pc[CW_FLAGS]) | CF_BLOCKED);
            const unsigned hid = id - static_cast<unsigned>(kSampleCtx);
            __hip_atomic_fetch_or(&sh.helper_free[hid >> 5], 1u << (hid & 31u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        const bool alive = mine && !died && !deliver;
        if (alive) ctx_store_cold(my_ctx, L, L.state != S_QUERY);
        // ---- the new query against everything that is not a Julia set, then the first set it enters
        if (__any(L.state == S_QUERY)) {
            const bool q = L.state == S_QUERY;
            if (STATS) {
                if (lane == 0) n.v[D_QUERY_PHASES]++;
                if (q) n.v[D_QUERY_LANES]++;
            }
            if (TIMED) {
                wc.add(W_QPHASES, 1);
                wc.add(W_QLANES, static_cast<unsigned>(__popcll(__ballot(q))));
            }
            const BvhStackLds no_lds_stack = {nullptr, 0u, 0u, sc.bvh_nodes};
            query_objects<STATS>(sc, L, q, no_lds_stack, n, ctx_hit_sink(my_ctx));
            if (q) {
                L.state = S_RESOLVE;
                if (L.qactive) {
                    for (unsigned k = 0; k < sc.n_julias; k++) {
                        bump<STATS>(n, C_JULIA_QUERIES);
                        if (julia_enter(sc.julias[k], L.qo, L.qd, L.jp, L.jd)) {
                            L.jidx = k;
                            L.state = S_WAIT;
                            push_job = true;
                            break;
                        }
                    }
                }
            }
        }
        // ---- back into the pool, and on to the next ring
        if (alive) ctx_store_hot(my_ctx, L, push_job, helper);
        // a node that waits for a helper: parked (on no ring) unless the answer has come in the meantime
        const bool park = alive && L.state == S_PARK;
        bool unparked = false;
        if (park) {
            lds_release();
            const unsigned old = __hip_atomic_fetch_or(my_join, 1u << (24u + L.li), __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((old >> (8u + L.li)) & 1u) {
                if ((old >> (16u + L.li)) & 1u) my_ctx[CW_FLAGS] = __uint_as_float(__float_as_uint(my_ctx[CW_FLAGS]) | CF_BLOCKED);
                unparked = true;
            }
        }
        ring_push(sh.ring[R_MARCH], &sh.ctl[2 * R_MARCH + 1], alive && push_job, id);
        ring_push(sh.ring[R_READY_T], &sh.ctl[2 * R_READY_T + 1], alive && !push_job && L.qkind != Q_SHADOW, id);
        ring_push(sh.ring[R_READY_S], &sh.ctl[2 * R_READY_S + 1], (alive && !push_job && L.qkind == Q_SHADOW && !park) || unparked, id);
        ring_push(sh.ring[R_READY_S], &sh.ctl[2 * R_READY_S + 1], wake, parent);
        const unsigned n_died = static_cast<unsigned>(__popcll(__ballot(died)));
        if (n_died != 0u && lane == 0) __hip_atomic_fetch_add(&sh.dead, n_died, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        beat(sh);
#if RT_JULIA_SHADER_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        charge(D_QUERY_CYCLES);
    }
    if (TIMED) wc.flush(job.counters);
}

template <bool STATS>
__global__ __launch_bounds__(kJuliaWaves * 64, kJuliaWaves / 4) void trace_julia_kernel(const DevScene sc_global, const DevJob job) {
    __shared__ PoolShared sh;
    const DevScene& sc = sc_global;
    for (unsigned i = threadIdx.x; i < kJuliaQueueCap; i += blockDim.x) {
        for (int r = 0; r < R_COUNT; r++) sh.ring[r].slot[i] = 0u;
        // every context starts on READY_T, asking for its first sample
        if (i < static_cast<unsigned>(kSampleCtx)) {
            sh.ring[R_READY_T].slot[i] = ring_word(i, i);
            sh.ctx[i][CW_FLAGS] = __uint_as_float(CF_FRESH);
        }
        if (i < static_cast<unsigned>(kPoolCtx)) sh.ctx[i][CW_JOIN] = __uint_as_float(0u);
        if (i <= static_cast<unsigned>(kHelperCtx / 32)) sh.helper_free[i] = i < static_cast<unsigned>(kHelperCtx / 32) ? ~0u : 0u;
        if (i < static_cast<unsigned>(kJuliaConstTable) && i < sc.n_julias) {
            for (int k = 0; k < 4; k++) sh.jconst[i][k] = sc.julias[i].c[k];
            sh.jconst[i][4] = sc.julias[i].cut_plane;
        }
    }
    if (threadIdx.x == 0) {
        for (int w = 0; w < 12; w++) sh.ctl[w] = 0u;
        sh.ctl[2 * R_READY_T + 1] = static_cast<unsigned>(kSampleCtx);
        sh.dead = 0u;
        sh.abort = (job.debug_abort != 0u && blockIdx.x == 0) ? 1u : 0u;   // RT_OPT_DEBUG_FORCE_ABORT
        sh.heartbeat = 0u;
        sh.pad = 0u;
        sh.hand = 0ull;
    }
    __syncthreads();
    const ShadeTables T = tables_of(sc, sc.shade_blob);
    Counters n;
    if (STATS) for (int i = 0; i < C_COUNT; i++) n.v[i] = 0;
    const int wave = threadIdx.x >> 6;
    if (wave < kJuliaMarcherWaves) {
#if RT_JULIA_TAIL_RING
        julia_marcher_pooled<STATS>(sc, sh, n);
#elif RT_JULIA_DUAL
        julia_marcher_dual<STATS>(sc, job, sh, n, job.counters);
#else
        julia_marcher<STATS>(sc, job, sh, n, job.counters);
#endif
    } else {
        const unsigned shader_wave = blockIdx.x * (kJuliaWaves - kJuliaMarcherWaves) + (wave - kJuliaMarcherWaves);
        WaveIds wave_ids = wave_ids_begin(static_cast<unsigned long long>(shader_wave) << 6, job.n_waves, job.static_blocks);
        julia_shader<STATS>(sc, T, job, sh, n, wave_ids);
    }
    if (lds_load(&sh.abort) != 0u && (threadIdx.x & 63) == 0) atomicExch(job.queue_head + 1, 1ull);
    flush_counters<STATS>(n, job.counters);
}

// ---- per-pixel sum of the samples, in the reference's order (src/raytracer.cpp:61-110) ------------
__global__ __launch_bounds__(256) void sum_samples_kernel(const float4* __restrict__ samples, float4* __restrict__ out,
                                                          unsigned long long n_pixels, unsigned samples_per_pixel, float inv_samples,
                                                          uchar4* __restrict__ out8, float gain) {
    const unsigned long long p = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (p >= n_pixels) return;
    Col rad = {0.0f, 0.0f, 0.0f, 0.0f};
    for (unsigned s = 0; s < samples_per_pixel; s++) {
        const float4 c = samples[static_cast<unsigned long long>(s) * n_pixels + p];
        rad = rad + Col{c.x, c.y, c.z, c.w};
    }
    const Col zero = {0.0f, 0.0f, 0.0f, 0.0f};
    const Col px = zero + rad * inv_samples;
    if (out8 != nullptr) out8[p] = scale_out(make_float4(px.r, px.g, px.b, px.a), gain);      // (the quantised picture only)
    else out[p] = make_float4(px.r, px.g, px.b, px.a);
}

// ---- photon roots that can leave nothing (DevScene::photon_cull_*) ----------------------------------------------
// Direction of root (theta index, phi index): src/photonmap.cpp:42-46 on the host's sin/cos tables.
RT_DEV V3 photon_root_direction(const DevPhotonJob& job, unsigned ti, unsigned pi) {
    const float st = job.sin_theta[ti], ct = job.cos_theta[ti], sp_ = job.sin_phi[pi], cp = job.cos_phi[pi];
    return normalized(v3(st * cp, st * sp_, ct));
}
// True when the root's ray misses the box around every object that refracts or reflects light rays.
RT_DEV bool photon_root_passes_by(const DevScene& sc, V3 origin, V3 dir) {
    const float4 lo = {sc.photon_cull_lo[0], sc.photon_cull_lo[1], sc.photon_cull_lo[2], 0.0f};
    const float4 hi = {sc.photon_cull_hi[0], sc.photon_cull_hi[1], sc.photon_cull_hi[2], 0.0f};
    return !slab_test(lo, hi, origin, cull_inverse(dir), kInfinite, slab_slack(lo, hi)).overlap;
}
// One thread per direction (light, phi, theta) of the pass: the directions whose ray may meet a specular object, in a
// compact list for photon_kernel (order within the list: as the waves arrive; a root keeps its own number).
__global__ __launch_bounds__(256) void photon_cull_kernel(const DevScene sc, const DevPhotonJob job, unsigned n_dirs, uint32_t* __restrict__ live_dirs,
                                                          unsigned long long* __restrict__ live_count) {
    const unsigned d = blockIdx.x * 256u + threadIdx.x;
    bool live = false;
    if (d < n_dirs) {
        const unsigned ti = d % job.n_theta, q = d / job.n_theta, pi = q % job.n_phi, li = q / job.n_phi;
        const DevLight& light = sc.lights[job.point_lights[li]];
        live = !photon_root_passes_by(sc, v3(light.position[0], light.position[1], light.position[2]), photon_root_direction(job, ti, pi));
    }
    const unsigned long long mask = __ballot(live);
    if (mask == 0ull) return;
    const unsigned lane = threadIdx.x & 63u;
    const int leader = __ffsll(static_cast<long long>(mask)) - 1;
    unsigned long long base = 0;
    if (static_cast<int>(lane) == leader) base = atomicAdd(live_count, static_cast<unsigned long long>(__popcll(mask)));
    base = __shfl(base, leader);
#ifdef RT_GUARD
    if (live && RT_GUARD_OK(job.guard_words, 8, base + __popcll(mask & ((1ull << lane) - 1ull)), job.guard_dirs)) live_dirs[base + __popcll(mask & ((1ull << lane) - 1ull))] = d;
#else
    if (live) live_dirs[base + __popcll(mask & ((1ull << lane) - 1ull))] = d;
#endif
}

// ---- photon pass (src/photonmap.cpp:20-146) ---------------------------------------------------------
// One lane = one root direction (light, phi, theta, hue).  The path tree is walked depth first with
// a stack of pending rays.  Deposits are RECORDED with their position in the reference's sequential pass
// (root id, number within the root) for rt_photon_order.hip to sum per texel in that order — or, with
// RT_OPT_PHOTON_ORDER = 0, added to the atlas with float atomics (texels equal to rounding only).
//
// The newest pending ray of a lane lives in LDS ([field][lane], like trace_kernel's frame top), older ones in the lane's
// stack column in HBM: a pop costs an LDS read, not a memory round trip in the middle of the lane's chain of queries.
template <bool HAS_JULIA, bool STATS>
__global__ __launch_bounds__(256, RT_PHOTON_WAVES_PER_SIMD) void photon_kernel(const DevScene sc, const DevPhotonJob job) {
    Counters n;
    if (STATS) for (int i = 0; i < C_COUNT; i++) n.v[i] = 0;
    const unsigned long long grid_lanes = static_cast<unsigned long long>(gridDim.x) * blockDim.x;
    const unsigned long long gid = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    const FrameRef stack = {job.stack + gid, grid_lanes};

    // shading tables from LDS when they fit (as in trace_kernel)
    extern __shared__ __align__(16) unsigned char lds_blob[];
    const unsigned char* table_base = sc.shade_blob;
    if (sc.shade_blob_in_lds) {
        for (unsigned i = threadIdx.x * 16u; i < sc.shade_blob_bytes; i += 256u * 16u) {
            *reinterpret_cast<uint4*>(lds_blob + i) = *reinterpret_cast<const uint4*>(sc.shade_blob + i);
        }
        __syncthreads();
        table_base = lds_blob;
    }
    const ShadeTables T = tables_of(sc, table_base);
    unsigned char* const lds_after_blob = lds_blob + (sc.shade_blob_in_lds ? ((sc.shade_blob_bytes + 15u) & ~15u) : 0u);
    // dynamic LDS: [shade blob copy][BVH stack columns][second frame slot][BVH records]
    const DevBvhPair* bvh_nodes = sc.bvh_nodes;
    if (sc.bvh_records_in_lds != 0u) {
        uint4* const dst = reinterpret_cast<uint4*>(lds_after_blob + sc.bvh_lds_stack_entries * 256u * sizeof(unsigned) +
                                                    (sc.frame_slots_in_lds == 2u ? kFrameFields * 256u * sizeof(float) : 0u));
        const uint4* const src = reinterpret_cast<const uint4*>(sc.bvh_nodes);
        for (unsigned i = threadIdx.x; i < sc.bvh_records_in_lds * 4u; i += 256u) dst[i] = src[i];
        __syncthreads();
        bvh_nodes = reinterpret_cast<const DevBvhPair*>(dst);
    }
    const BvhStackLds bvh_stack = {reinterpret_cast<unsigned*>(lds_after_blob) + threadIdx.x, sc.bvh_lds_stack_entries, 256u, bvh_nodes};
    Lane L;
    L.state = S_IDLE;
    L.sp = 0;
    float rad[4] = {0, 0, 0, 0};
    unsigned depth = 0;
    __shared__ float top_store[kFrameFields * 256];
    float* const top = top_store + threadIdx.x;      // the lane's newest pending ray: field c at top[c * 256]
    bool has_top = false;                // ... is there; L.sp counts the older ones in HBM
    bool need = true;
    // the ids to hand out: every root of the launch, or (RT_OPT_CULL_PHOTON_ROOTS) the hues of the directions that
    // photon_cull_kernel kept — their number is known on the device only, so the split of the id space is made here
    unsigned long long n_ids = job.n_roots, dyn_base = job.dyn_base;
    unsigned static_blocks = job.static_blocks;
    if (job.live_dirs != nullptr) {
        n_ids = static_cast<unsigned long long>(*job.live_count) * job.hues;
        static_blocks = static_cast<unsigned>(n_ids / 64ull * job.static_sixteenths / 16ull / job.n_waves);
        dyn_base = static_cast<unsigned long long>(job.n_waves) * static_blocks * 64ull;
    }
    WaveIds wave_ids = wave_ids_begin(gid, job.n_waves, static_blocks);
    unsigned long long root_id = 0;      // of the tree this lane walks, and the number of its deposits so far:
    unsigned root_deposits = 0;          // together the position of a deposit in the reference's sequential pass
    bool cullable = false;               // counters build: this root lies outside the cull box (traced all the same; must not meet a specular object)
    unsigned long long dep_next = 0, dep_end = 0;   // records this wave has reserved (wave-uniform)
    unsigned long long slot_owed = ~0ull;           // record whose number within its texel (slot_value) is still to be stored
    unsigned slot_value = 0;
    for (;;) {
        // refill idle lanes with new roots
        while (__any(need)) {
            unsigned long long id = 0;
            const bool got = fetch_ids(job.queue_head, dyn_base, n_ids, job.id_chunk, wave_ids, need, id);
            if (need) {
                need = false;
                if (got) {
                    // (light, phi, theta, hue) of root `id`; 64-bit division costs some 150 instructions a piece, and the
                    // wave pays for the decode in every round in which a single lane starts a root
                    unsigned hue, ti, pi, li;
                    if (job.live_dirs != nullptr) {
                        // ids count the hues of the LIVE directions (photon_cull_kernel); the root keeps its number in the full pass
                        const unsigned i32 = static_cast<unsigned>(id);
                        hue = i32 % job.hues;
#ifdef RT_GUARD
                        const unsigned dir_id = RT_GUARD_OK(job.guard_words, 3, i32 / job.hues, *job.live_count) ? job.live_dirs[i32 / job.hues] : 0u;
                        (void)RT_GUARD_OK(job.guard_words, 31, dir_id, job.guard_dirs);
#else
                        const unsigned dir_id = job.live_dirs[i32 / job.hues];
#endif
                        root_id = static_cast<unsigned long long>(dir_id) * job.hues + hue;
                        ti = dir_id % job.n_theta;
                        const unsigned q = dir_id / job.n_theta;
                        pi = q % job.n_phi;
                        li = q / job.n_phi;
                    } else if ((job.root_base + job.n_roots) >> 32 == 0) {
                        const unsigned i32 = static_cast<unsigned>(id + job.root_base);
                        root_id = i32;
                        hue = i32 % job.hues;
                        unsigned q = i32 / job.hues;
                        ti = q % job.n_theta;
                        q /= job.n_theta;
                        pi = q % job.n_phi;
                        li = q / job.n_phi;
                    } else {
                        id += job.root_base;
                        root_id = id;
                        hue = static_cast<unsigned>(id % job.hues);
                        unsigned long long q = id / job.hues;
                        ti = static_cast<unsigned>(q % job.n_theta);
                        q /= job.n_theta;
                        pi = static_cast<unsigned>(q % job.n_phi);
                        li = static_cast<unsigned>(q / job.n_phi);
                    }
                    root_deposits = 0;
#ifdef RT_GUARD
                    if (!RT_GUARD_OK(job.guard_words, 71, ti, job.n_theta)) ti = 0;
                    if (!RT_GUARD_OK(job.guard_words, 72, pi, job.n_phi)) pi = 0;
                    if (!RT_GUARD_OK(job.guard_words, 73, li, job.n_point_lights)) li = 0;
#endif
                    const DevLight& light = T.lights[job.point_lights[li]];
                    const V3 origin = v3(light.position[0], light.position[1], light.position[2]);
                    const V3 dir = photon_root_direction(job, ti, pi);
                    // the counters build traces every root and checks what photon_cull_kernel claims of those it would drop
                    cullable = STATS && sc.photon_cull_valid != 0u && photon_root_passes_by(sc, origin, dir);
                    if (cullable) bump<STATS>(n, C_PHOTON_ROOTS_CULLED);
                    if (sc.dispersion) {
                        // HSVtoRGB(h) * factor / 4
                        for (int c = 0; c < 4; c++) rad[c] = (sc.hue[hue][c] * sc.photon_factor) / 4.0f;
                        L.wavelength = (static_cast<float>(hue) * 45.0f) / 180.0f - 1.0f;
                    } else {
                        for (int c = 0; c < 4; c++) rad[c] = 1.0f * sc.photon_factor;
                        L.wavelength = 0.0f;
                    }
                    depth = 0;
                    L.sp = 0;
                    begin_query(L, origin, dir, kInfinite, Q_PHOTON);
                }
            }
        }
        if (!__any(L.state == S_QUERY)) break;

        const bool mine = L.state == S_QUERY;
        query_objects<STATS>(sc, L, mine, bvh_stack, n);
        bool deposit = false;
        unsigned long long deposit_texel = 0;   // texel of the whole atlas (receivers x size^2 can exceed 2^32)
        float deposit_rad[4] = {0, 0, 0, 0};
        if (mine) {
            bump<STATS>(n, C_PHOTON_SEGMENTS);
            if (HAS_JULIA) {
                for (unsigned k = 0; k < sc.n_julias; k++) {
                    const DevJulia& j = sc.julias[k];
                    HitRec h;
                    unsigned long long steps = 0, iters = 0;
                    bump<STATS>(n, C_JULIA_QUERIES);
                    if (julia_intersect_inline<STATS>(j, L.qo, L.qd, kInfinite, h, steps, iters)) accept_julia_hit(sc, L, h, j, T.materials[j.material]);
                    bump<STATS>(n, C_JULIA_STEPS, steps);
                    bump<STATS>(n, C_JULIA_DE_ITERATIONS, iters);
                }
            }
            finish_best_triangle(T, L);
            bool pop = true;
#ifdef RT_GUARD
            if (L.found && !RT_GUARD_OK(job.guard_words, 4, L.best_obj, sc.n_objects)) L.found = 0u;
#endif
            if (L.found) {
                const DevMaterial& m = T.materials[T.object_material[L.best_obj]];
                if (m.refr_norm <= 0.0f) {
                    // the light ray ends on a diffuse object (src/photonmap.cpp:79-83)
                    const int slot = T.object_photon_slot[L.best_obj];
                    if (depth > 0 && slot >= 0) {
                        bump<STATS>(n, C_PHOTON_DEPOSITS);
                        deposit = true;
                        deposit_texel = static_cast<unsigned long long>(slot) * sc.photon_size * sc.photon_size + photon_texel(L.best.ph, sc.photon_size);
                        for (int c = 0; c < 4; c++) deposit_rad[c] = rad[c];
                    }
                } else if (depth + 1 <= sc.max_bounces) {
                    if (STATS && cullable) n.v[D_PHOTON_CULL_VIOLATIONS]++;
                    const float cosv = clampf(dot(L.qd, L.best.normal), -1.0f, 1.0f);
                    bump<STATS>(n, C_FRESNEL);
                    const float kr = fresnel_kr(m.refr_re, m.refr_im, m.dispersion, cosv, L.wavelength);
                    V3 ro, rd, mo, md;
                    const bool refr_ok = kr < 1.0f && refraction_ray(L.qd, L.best.point, L.best.normal, m.refr_re, m.dispersion, cosv, L.wavelength, ro, rd);
                    reflection_ray(L.qd, L.best.point, L.best.normal, cosv, mo, md);
                    if (refr_ok) {
                        // reflection (always cast, src/photonmap.cpp:141-144) waits on the stack: in LDS, the ray that was there in HBM
                        if (has_top && RT_GUARD_OK(job.guard_words, 1, L.sp, job.guard_stack_frames)) {
                            const unsigned f = L.sp;
#pragma unroll
                            for (int c = 0; c < kFrameFields; c++) stack.at(f, c) = top[c * 256];
                            L.sp = f + 1;
                        }
                        top[0 * 256] = mo.x; top[1 * 256] = mo.y; top[2 * 256] = mo.z;
                        top[3 * 256] = md.x; top[4 * 256] = md.y; top[5 * 256] = md.z;
                        for (int c = 0; c < 4; c++) top[(6 + c) * 256] = rad[c] * kr;
                        top[10 * 256] = __uint_as_float(depth + 1);
                        has_top = true;
                        for (int c = 0; c < 4; c++) rad[c] = rad[c] * (1 - kr);
                        begin_query(L, ro, rd, kInfinite, Q_PHOTON);
                    } else {
                        for (int c = 0; c < 4; c++) rad[c] = rad[c] * kr;
                        begin_query(L, mo, md, kInfinite, Q_PHOTON);
                    }
                    depth = depth + 1;
                    pop = false;
                }
            }
            if (pop) {
                if (has_top) {
                    const V3 o = v3(top[0 * 256], top[1 * 256], top[2 * 256]);
                    const V3 d = v3(top[3 * 256], top[4 * 256], top[5 * 256]);
                    for (int c = 0; c < 4; c++) rad[c] = top[(6 + c) * 256];
                    depth = __float_as_uint(top[10 * 256]);
                    has_top = false;
                    begin_query(L, o, d, kInfinite, Q_PHOTON);
                } else if (L.sp > 0 && RT_GUARD_OK(job.guard_words, 2, L.sp - 1, job.guard_stack_frames)) {
                    const unsigned f = L.sp - 1;
                    const V3 o = v3(stack.at(f, 0), stack.at(f, 1), stack.at(f, 2));
                    const V3 d = v3(stack.at(f, 3), stack.at(f, 4), stack.at(f, 5));
                    for (int c = 0; c < 4; c++) rad[c] = stack.at(f, 6 + c);
                    depth = __float_as_uint(stack.at(f, 10));
                    L.sp = f;
                    begin_query(L, o, d, kInfinite, Q_PHOTON);
                } else {
                    L.state = S_IDLE;
                    need = true;
                }
            }
        }
        // Object::addPhoton (src/objects.cpp:244-256), with the wave converged again
#ifdef RT_GUARD
        if (deposit && !RT_GUARD_OK(job.guard_words, 5, deposit_texel, job.guard_texels)) deposit = false;
#endif
        const unsigned long long dep_mask = __ballot(deposit);
        if (dep_mask != 0) {
            bool add_now = deposit;
            if (job.dep_keys != nullptr) {
                const unsigned lane = threadIdx.x & 63u;
                const unsigned cnt = static_cast<unsigned>(__popcll(dep_mask));
                if (dep_next + cnt > dep_end) {   // the rest of the old chunk stays unused (its keys sort last)
                    unsigned long long base = 0;
                    const unsigned reserve = cnt > kDepositChunk ? cnt : kDepositChunk;
                    if (lane == static_cast<unsigned>(__ffsll(static_cast<long long>(dep_mask)) - 1)) base = atomicAdd(job.dep_words, static_cast<unsigned long long>(reserve));
                    base = __shfl(base, __ffsll(static_cast<long long>(dep_mask)) - 1);
                    dep_next = base;
                    dep_end = base + reserve;
                }
                const unsigned long long slot_index = dep_next + static_cast<unsigned>(__popcll(dep_mask & ((1ull << lane) - 1ull)));
                dep_next += cnt;
                if (deposit) {
                    if (root_deposits >> job.dep_seq_bits != 0u) {
                        atomicAdd(job.dep_words + 3, 1ull);   // more deposits from one root than the key has room for
                    } else if (slot_index < job.dep_capacity) {
                        job.dep_keys[slot_index] = (deposit_texel << (job.dep_root_bits + job.dep_seq_bits)) |
                                                   (root_id << job.dep_seq_bits) | root_deposits;
                        job.dep_vals[slot_index] = make_float4(deposit_rad[0], deposit_rad[1], deposit_rad[2], deposit_rad[3]);
                        if (job.dep_count != nullptr) {
                            // the count the atomic returns is stored at the lane's next deposit (or at the end): nothing waits for the round trip
                            if (slot_owed != ~0ull && RT_GUARD_OK(job.guard_words, 6, slot_owed, job.dep_capacity)) job.dep_slot[slot_owed] = slot_value;
                            slot_value = atomicAdd(job.dep_count + deposit_texel, 1u);
                            slot_owed = slot_index;
                        }
                        add_now = false;
                    } else {
                        atomicAdd(job.dep_words + 1, 1ull);   // buffer full: this deposit is added out of order (host grows the buffer)
                    }
                    root_deposits++;
                }
            }
            if (add_now) {
                float* t = reinterpret_cast<float*>(sc.photon_atlas + deposit_texel);
                atomicAdd(t + 0, deposit_rad[0]);
                atomicAdd(t + 1, deposit_rad[1]);
                atomicAdd(t + 2, deposit_rad[2]);
                atomicAdd(t + 3, deposit_rad[3]);
            }
        }
    }
    if (slot_owed != ~0ull && RT_GUARD_OK(job.guard_words, 61, slot_owed, job.dep_capacity)) job.dep_slot[slot_owed] = slot_value;
    flush_counters<STATS>(n, job.counters);
}

// ---- known-answer kernels: one function of the path per lane ----------------------------------------
// Object::intersect for (object, origin, direction, max) records; out = hit flag + 11 floats.
__global__ void kat_intersect_kernel(const DevScene sc, const unsigned* __restrict__ kind, const unsigned* __restrict__ prim,
                                     const float* __restrict__ rays, unsigned n, float* __restrict__ out) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = rays + static_cast<size_t>(i) * 7;
    const V3 o = v3(r[0], r[1], r[2]);
    const V3 d = normalized(v3(r[3], r[4], r[5]));   // the Ray constructor normalises
    const float maxd = r[6];
    HitRec h = {};
    bool hit = false;
    if (kind[i] == 0) {
        hit = sphere_intersect(sc.spheres + prim[i], o, d, maxd, h);
    } else if (kind[i] == 1) {
        float u, v, t;
        if (tri_test(load_tri(sc.tri_test + prim[i]), o, d, maxd, u, v, t)) {
            tri_finish(sc.tri_shade[prim[i]], o, d, u, v, t, h);
            hit = true;
        }
    } else {
        unsigned long long a = 0, b = 0;
        hit = julia_intersect_inline<false>(sc.julias[prim[i]], o, d, maxd, h, a, b);
    }
    float* w = out + static_cast<size_t>(i) * 12;
    w[0] = hit ? 1.0f : 0.0f;
    w[1] = h.dist;
    w[2] = h.point.x; w[3] = h.point.y; w[4] = h.point.z;
    w[5] = h.normal.x; w[6] = h.normal.y; w[7] = h.normal.z;
    w[8] = h.uv.x; w[9] = h.uv.y; w[10] = h.ph.x; w[11] = h.ph.y;
}

// calcFresnel for (re, im, dispersion, cos, wavelength) records
__global__ void kat_fresnel_kernel(const float* __restrict__ in, unsigned n, float* __restrict__ out) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = in + static_cast<size_t>(i) * 5;
    out[i] = fresnel_kr(r[0], r[1], r[2], r[3], r[4]);
}

// ---- Picture::mulAdd and Color::scaleOut (src/types.h:372-376, 274-282) -----------------------------
__global__ __launch_bounds__(256) void mul_add_kernel(float* __restrict__ acc, const float* __restrict__ src, float factor, unsigned long long n) {
    const unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) acc[i] = acc[i] + src[i] * factor;
}

__global__ __launch_bounds__(256) void quantize_kernel(const float4* __restrict__ rgba, uchar4* __restrict__ out, float gain, unsigned long long n) {
    const unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = scale_out(rgba[i], gain);
}

// explicit instantiations used by rt_api.hip
template __global__ void trace_kernel<false, false>(const DevScene, const DevJob);
template __global__ void trace_kernel<true, false>(const DevScene, const DevJob);
template __global__ void trace_kernel<false, true>(const DevScene, const DevJob);
template __global__ void trace_kernel<true, true>(const DevScene, const DevJob);
template __global__ void trace_julia_kernel<false>(const DevScene, const DevJob);
template __global__ void trace_julia_kernel<true>(const DevScene, const DevJob);
template __global__ void photon_kernel<false, false>(const DevScene, const DevPhotonJob);
template __global__ void photon_kernel<false, true>(const DevScene, const DevPhotonJob);
template __global__ void photon_kernel<true, false>(const DevScene, const DevPhotonJob);
template __global__ void photon_kernel<true, true>(const DevScene, const DevPhotonJob);

}  // namespace rtd
