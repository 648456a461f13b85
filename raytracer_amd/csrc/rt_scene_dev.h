// Device-resident scene layout (HBM) shared by rt_api.hip and the kernels.
//
// All of it is a few KB to a few MB: geometry tables are read with wave-uniform indices (scalar
// loads, SGPR operands) inside the object loops; shading tables, textures and the photon atlas are
// read per lane.  Tables are 16-byte aligned records so a uniform record is one or two s_load_dwordx8.
#pragma once

#include <stdint.h>
#include "rt_divide.h"

namespace rtd {

enum RunKind : int { RUN_SPHERES = 0, RUN_TRIANGLES = 1, RUN_TRIANGLE_BVH = 2 };

// A maximal run of consecutive scene objects of one kind (scene order is preserved across runs).
struct DevRun {
    int kind;
    uint32_t first_object;   // index into the per-object tables (= position in Scene::objects())
    uint32_t first_prim;     // index into the per-kind geometry table
    uint32_t count;
    uint32_t bvh_root;       // RUN_TRIANGLE_BVH: first node of this run's BVH
    uint32_t material;       // RUN_TRIANGLE_BVH: the run's single material
    uint32_t first_chunk;    // RUN_TRIANGLES: first entry of this run in tri_chunks
    uint32_t pad1;
};

// What the Möller–Trumbore test needs (edges precomputed with the reference's own subtraction,
// src/objects.cpp:56-57).  48 bytes.
struct DevTriTest {
    float p0[3], e1[3], e2[3];
    float pad[3];
};
// What shading a triangle hit needs.  64 bytes.
struct DevTriShade {
    float n0[3], n1[3], n2[3];
    float uv0[2], uv1[2], uv2[2];
    float pad;
};
// 160 bytes
struct DevSphere {
    float center[3];
    float radius;
    float w2o[12], o2w[12], o2wn[12];
};
// World-space bounding sphere of a DevSphere (an ellipsoid after its transform) with the slack that
// makes "outside" a proof that Sphere::intersect as written returns no hit (rt_api.hip).  32 bytes.
struct DevSphereBound {
    float center[3];
    float radius;            // grown by 1e-3
    float radius2;           // radius^2
    float slack_scale;       // slack = slack_scale * (|origin|^2 + magnitude2): bound on the float error of src/objects.cpp:12-31
    float magnitude2;        // |center|^2 + radius^2 + |object-space centre|^2 ...
    float pad;
};
// 192 bytes
struct DevJulia {
    float position[3];
    float scale;
    float c[4];
    float cut_plane;
    float inv_scale;         // 1.0f / scale as the reference computes it (src/objects.cpp:165)
    uint32_t object;         // position in Scene::objects()
    uint32_t material;
    float w2o[12], o2w[12], o2wn[12];
};
// 64 bytes
struct DevMaterial {
    float color[4];
    int texture;
    float ka, kd, ks, exponent;
    float reflectance, transmittance;
    float refr_re, refr_im, dispersion;
    float refr_norm;         // re*re + im*im (std::norm)
    uint32_t phong_tame;     // 1: colour, kd and ks finite and modest, exponent in [0, 1e6]: a light whose diffuse and specular factors are zero adds exactly +-0 here (resolve_lane)
};
struct DevTexture {
    uint32_t width, height;
    uint32_t first_texel;    // into the RGBA8 texel pool
    uint32_t pad;
};
struct DevLight {
    float position[3];
    int type;
    float power[4];
};
// Bounds of kTriChunk consecutive triangles of a RUN_TRIANGLES run (conservative for
// Triangle::intersect as written, like the BVH leaf boxes).  A chunk that no ray of the wave can
// touch is skipped as a whole; the chunks are still visited in scene order.  32 bytes.
constexpr uint32_t kTriChunk = 4;
struct DevTriChunk {
    float lo[3];
    float slack;             // distance slack of the cull: 1e-4 of the box's extents
    float hi[3];
    float pad1;
};
// Binary BVH node as the builder fits it (rt_bvh.hip): {lo, a, hi, b}; internal a/b = child node indices, leaf a = triangle, b = kBvhLeaf
struct DevBvhNode {
    float lo[3];
    uint32_t left_or_first;
    float hi[3];
    uint32_t count;
};
// What the traversal reads: one 64-byte record per INTERNAL node holding the boxes of both children, so that one
// cache line decides which of them to enter (and which first).  child: bit 31 set = leaf, low bits = triangle index
// into tri_test / tri_shade; else index of the child's record (relative to the run's first record).
constexpr uint32_t kBvhLeafBit = 0x80000000u;
struct DevBvhPair {
    float lo0[3];
    uint32_t child0;
    float hi0[3];
    uint32_t child1;
    float lo1[3];
    float slack0;     // distance slack of the slab test (rt_kernels.hip: slab_test) for child 0's box: 1e-4 of its extents' sum
    float hi1[3];
    float slack1;     // ... for child 1's
};

struct DevScene {
    // camera constants precomputed on the host with the reference's expressions (src/raytracer.cpp:17-29)
    float cam[12];
    float pixel_size[2];
    float focus_distance, lens_size;
    const float* ray_x;          // [width]  rayX as accumulated along a row (src/raytracer.cpp:59,111)
    const float* ray_y;          // [height]
    const float* sub_disp;       // [supersampling] 2(i+1)/(n+1)-1
    uint32_t width, height, supersampling, max_bounces;
    float inv_samples;           // 1.0f/(n*n)
    int dispersion;              // 0/1
    float hue[8][4];             // rgb + 1
    float background[4], ambient[4];

    uint32_t n_lights, n_objects, n_runs, n_julias;
    const DevLight* lights;
    const DevRun* runs;
    const DevTriTest* tri_test;
    const DevTriShade* tri_shade;
    const DevTriChunk* tri_chunks;
    const DevSphere* spheres;
    const DevSphereBound* sphere_bounds;
    const DevJulia* julias;
    const uint32_t* object_material;   // [n_objects]
    const int32_t* object_photon_slot; // [n_objects] slot in the photon atlas or -1
    const DevMaterial* materials;
    const DevTexture* textures;
    const uint32_t* texels;            // RGBA8 as in rt_scene_desc (R in the low byte)
    const DevBvhPair* bvh_nodes;       // n-1 records per BVH run of n triangles (root = record 0 of the run)
    const uint32_t* bvh_prims;         // triangle indices (into tri_test) in leaf order

    // lights, materials, textures, tri_shade and the two per-object tables live in ONE allocation
    // ("shade blob"); when it is small enough the trace kernel copies it into LDS and the per-lane
    // shading reads go there instead of L2 (the pointers above are rebased inside the kernel)
    const unsigned char* shade_blob;
    uint32_t shade_blob_bytes;
    uint32_t shade_blob_in_lds;        // 0/1, decided by the host from the blob size
    uint32_t bvh_records_in_lds;       // > 0: the scene's BVH records (all runs, this many) are few enough to be copied into LDS
    uint32_t frame_slots_in_lds;       // trace_kernel: newest frames of a lane kept in LDS, 1 or 2 (TopInLds)
    uint32_t lights_benign;            // 0/1: every light has a tame power and position (see DevMaterial::phong_tame)
    uint32_t bvh_lds_stack_entries;    // per-lane BVH stack entries kept in LDS (behind the blob copy); deeper ones go to scratch

    float4* photon_atlas;              // [n_slots][size][size]
    uint32_t photon_size;
    float photon_factor;
    // Photon roots (src/photonmap.cpp:38-58): a light ray deposits only after a specular bounce (:79-83), so a root
    // that cannot meet a specular object leaves nothing.  Box around every object whose material refracts or reflects
    // light rays (refr_norm > 0), conservative for the intersectors as written (triangle boxes as in DevTriChunk, spheres
    // by DevSphereBound with its rounding slack at the lights' positions); valid = 0 when such an object has no finite
    // bound (a Julia set, a degenerate triangle, a wild transform) or there is none.
    float photon_cull_lo[3];
    uint32_t photon_cull_valid;
    float photon_cull_hi[3];
    uint32_t photon_cull_pad;
};

// One render call's partition of the image (rt_bands) and where its samples go.
struct DevJob {
    uint32_t band_rows, first_band, band_stride;
    uint32_t compact_rows;       // rows rendered by this call
    uint32_t tiles_x, tiles_y;   // 8x8 tiles over the compact image
    uint32_t samples_per_pixel;  // n*n*(8 if dispersion)
    uint64_t n_samples;          // tiles_x*tiles_y*samples_per_pixel*64 (incl. padding lanes of edge tiles)
    uint64_t n_pixels;           // compact_rows*width
    float4* samples;             // [samples_per_pixel][n_pixels] or NULL when samples_per_pixel == 1
    float4* out;                 // [n_pixels], or NULL when the caller wants the quantised picture only:
    uchar4* out8;                // [n_pixels] Color::scaleOut of the pixel with gain out8_gain (src/types.h:274-282), or NULL
    float out8_gain;
    uint32_t pad_out8;
    float* stack;                // per-lane DFS frames: [frame][field][grid lane]
    unsigned long long* queue_head;
    unsigned long long* counters;// rt_stats words (only touched when STATS)
    uint32_t dof_seed;
    uint32_t id_chunk;           // sample ids a wave reserves per atomic (multiple of 64)
    uint32_t static_blocks;      // 64-id blocks every wave owns outright: wave w of n_waves owns blocks w, w + n_waves, ...
    uint32_t n_waves;
    uint64_t dyn_base;           // first id handed out through queue_head
    uint32_t elide_unlit;        // RT_OPT_ELIDE_UNLIT_SHADOWS and no counters: shadow rays whose light can only add +-0 are not traced
    uint32_t debug_abort;        // RT_OPT_DEBUG_FORCE_ABORT: trace_julia_kernel's first workgroup raises its watchdog flag at once
    uint32_t n_tiles;            // tiles_x*tiles_y
    // trace_julia_kernel with samples_per_pixel > 1 (rt_kernels.hip, "long chains first"): ids are handed out in blocks of
    // group_tiles tiles' worth of one sample, numbered sample-major (block = sample*n_groups + group), one by one through
    // queue_head; the blocks of groups whose first sample turns out to hold a long Julia march jump the queue through the
    // hot lists.  Words of `sched` ("tile" below = group of tiles):
    // [2k] hot entries of level k pushed, [2k + 1] claimed (k < kSchedLevels), [kSchedHeaderWords ..) one word per tile (the
    // highest level it is on, plus one), then one claim bit per block, then the kSchedLevels hot lists (block numbers, preset
    // to ~0).  NULL: plain tile-major order, static shares as in trace_kernel.
    uint32_t* sched;
    uint32_t hot_steps;          // a march of a first sample that reaches this many steps makes its tile hot
    uint32_t group_tiles;        // tiles per block under `sched` (a wave's fetch costs three or four round trips to memory: fewer, larger blocks)
    uint32_t n_groups;           // ceil(n_tiles / group_tiles)
    // start_sample's divisors, fixed for the launch (rt_divide.h): a sample id is taken apart with five multiplications
    // instead of a 64-bit and four 32-bit divisions (some 250 VALU instructions per sample, 4 % of C2's).  fast_decode is
    // off when an id's block number does not fit 32 bits (a frame of 2^38 samples).
    uint32_t fast_decode;
    DevDiv div_samples, div_tiles_x, div_band_rows, div_supersampling;
};
constexpr uint32_t kSchedHeaderWords = 8;
constexpr uint32_t kSchedLevels = 3;     // a march that reaches hot_steps << k steps puts its tile on list k

// Photon emission job (src/photonmap.cpp:38-57)
struct DevPhotonJob {
    const float* sin_phi;   // tables produced by the host with the reference's float loop
    const float* cos_phi;
    const float* sin_theta;
    const float* cos_theta;
    uint32_t n_phi, n_theta;
    uint32_t n_point_lights;
    const uint32_t* point_lights;   // indices into lights
    uint32_t hues;                  // 8 in dispersion mode else 1
    uint64_t n_roots;               // roots this launch emits (all of n_point_lights*n_phi*n_theta*hues, or one slice of them)
    uint64_t root_base;             // first root of the slice (rt_photon_generate_part_device)
    float* stack;
    unsigned long long* queue_head;
    unsigned long long* counters;
    uint32_t id_chunk;              // root ids a wave reserves per atomic
    uint32_t static_blocks;         // as in DevJob
    uint32_t n_waves;
    uint64_t dyn_base;
    // Reference deposit order (RT_OPT_PHOTON_ORDER = 1): deposits are RECORDED, not added — key =
    // (texel of the atlas, root id, number of the deposit within its root), value = the radiance — then
    // sorted and summed per texel in key order, which is the order of the reference's sequential pass
    // (rt_photon_order.hip).  dep_keys == NULL: float atomics straight into the atlas.
    unsigned long long* dep_keys;   // [dep_capacity], preset to ~0 (sorts last)
    float4* dep_vals;               // [dep_capacity]
    unsigned long long* dep_words;  // [0] next free record, [1] deposits that found the buffer full
    uint32_t* dep_count;            // [texels of the atlas] deposits recorded per texel (photon_order_bucket), or NULL
    uint32_t* dep_slot;             // [dep_capacity] a record's number within its texel (the count it found)
    uint32_t dep_capacity;
    uint32_t dep_seq_bits;          // key = texel << (dep_root_bits + dep_seq_bits) | root << dep_seq_bits | seq
    uint32_t dep_root_bits;
    uint32_t static_sixteenths;     // share of the ids handed out as static blocks (for the split the kernel makes itself, below)
    // RT_OPT_CULL_PHOTON_ROOTS and no counters: the pass runs over the directions photon_cull_kernel kept (those whose ray
    // may meet a specular object, DevScene::photon_cull_*) instead of over all of them.  NULL: every root.
    const uint32_t* live_dirs;              // direction = (light * n_phi + phi) * n_theta + theta
    const unsigned long long* live_count;   // how many (device memory: known only after photon_cull_kernel ran)
#ifdef RT_GUARD
    // developer build (tools/guard_build.sh): every index of the pass is checked against these; the first violation is
    // reported in guard_words[0] (site) and [1] (index) and the access is skipped
    unsigned long long* guard_words;
    uint64_t guard_texels, guard_dirs, guard_stack_frames;
#endif
};
#ifdef RT_GUARD
#define RT_GUARD_OK(words, site, index, limit) \
    (static_cast<unsigned long long>(index) < static_cast<unsigned long long>(limit) ? true : \
     (atomicCAS((words), 0ull, static_cast<unsigned long long>(site)) == 0ull ? ((words)[1] = static_cast<unsigned long long>(index), (words)[2] = static_cast<unsigned long long>(limit), false) : false))
#else
#define RT_GUARD_OK(words, site, index, limit) true
#endif
#ifndef RT_DEPOSIT_CHUNK
#define RT_DEPOSIT_CHUNK 64      // [MI355X, C5] photon_kernel 245 us at 32, 234 at 64, 232 at 128, 230 at 256 (tools/photon_chunk_ab.sh); a wave leaves half a chunk unused
#endif
constexpr uint32_t kDepositChunk = RT_DEPOSIT_CHUNK;      // records a wave reserves per atomic (one counter for the whole chip)
constexpr uint32_t kDepositLongRun = 32;    // deposits into one texel from which a wave sums them (rt_photon_order.hip)

// Shape of trace_julia_kernel's workgroup (rt_kernels.hip), shared with the launch code
#ifndef RT_JULIA_WAVES
#define RT_JULIA_WAVES 12
#endif
#ifndef RT_JULIA_MARCHERS
#define RT_JULIA_MARCHERS 9       // [MI355X, round 3] 10 + 2: 26.7 ms, eighths of the bands 14.2-17.0; 9 + 3: 26.1, 13.0-16.2; 8 + 4: 26.7, 11.8-15.1 but 90.7 against 87 ms at four times the size
#endif
#ifndef RT_JULIA_DUAL
#define RT_JULIA_DUAL 0             // 1: marcher lanes hold two jobs and one estimator state (rt_kernels.hip: julia_marcher_dual)
#endif
#ifndef RT_JULIA_TAIL_RING
#define RT_JULIA_TAIL_RING 0        // 1: march steps pooled through a fifth ring (rt_kernels.hip: julia_marcher_pooled; measured slower)
#endif
#ifndef RT_JULIA_POOL
#define RT_JULIA_POOL (RT_JULIA_TAIL_RING ? 896 : 960)     // with the rings: all the LDS of a CU
#endif
constexpr int kJuliaWaves = RT_JULIA_WAVES;              // per workgroup, a multiple of 4 (waves per SIMD = kJuliaWaves / 4)
constexpr int kJuliaMarcherWaves = RT_JULIA_MARCHERS;    // the rest are shader waves
constexpr int kJuliaPoolContexts = RT_JULIA_POOL;        // ray contexts per workgroup (LDS pool)
constexpr uint32_t kJuliaStaticLdsBytes = 160 * 1024;    // its static LDS (pool + rings) may take the whole CU's

// waves per SIMD of trace_kernel (= its 256-lane workgroups per CU): sets its register budget (168 VGPRs at 3, 128 at 4)
// and the LDS a workgroup may take (rt_api.hip)
#ifndef RT_TRACE_WAVES_PER_SIMD
#define RT_TRACE_WAVES_PER_SIMD 3
#endif

enum CounterSlot : int {
    C_PRIMARY = 0, C_SECONDARY, C_SHADOW, C_PHOTON_SEGMENTS, C_PHOTON_DEPOSITS, C_SPHERE_TESTS, C_TRIANGLE_TESTS,
    C_JULIA_QUERIES, C_JULIA_STEPS, C_JULIA_DE_ITERATIONS, C_JULIA_NORMALS, C_SHADED_HITS, C_PHONG_LIGHTS,
    C_FRESNEL, C_TEXTURE_LOOKUPS, C_BVH_NODE_VISITS, C_SHADOW_ELIDED, C_PHOTON_ROOTS_CULLED,
    // scheduler diagnostics (rt_debug_counters; not part of rt_stats)
    D_QUERY_PHASES, D_QUERY_LANES, D_MARCH_PHASES, D_MARCH_LANES, D_NORMAL_PHASES, D_NORMAL_LANES,
    D_RESOLVE_PHASES, D_RESOLVE_LANES, D_DE_WAVE_ITERS,
    D_QUERY_CYCLES, D_MARCH_CYCLES, D_NORMAL_CYCLES, D_RESOLVE_CYCLES, D_TRI_CYCLES, D_SPHERE_CYCLES,
    D_R_NODE_CYCLES, D_R_LIGHT_CYCLES, D_R_SHADOWSETUP_CYCLES, D_R_FRESNEL_CYCLES, D_R_RETURN_CYCLES, D_R_FETCH_CYCLES,
    D_M_TAIL_CYCLES, D_M_IDLE_CYCLES, D_M_REFILL_CYCLES, D_T_WAIT_CYCLES, D_M_TAIL_BLOCKS, D_M_TAIL_LANES,
    D_STEP_BLOCKS_1, D_STEP_BLOCKS_2, D_STEP_BLOCKS_4, D_STEP_BLOCKS_8, D_STEP_BLOCKS_16, D_STEP_BLOCKS_32, D_STEP_BLOCKS_MORE,   // march steps by estimator blocks of four
    D_STEP_CLASS_SWITCH, D_STEP_BLOCK_SUM,   // steps whose class (<= 4 blocks or more) differs from the ray's previous step; total blocks
    D_RING_SAMPLES, D_RING_MARCH_SUM, D_RING_NORMAL_SUM, D_RING_READY_T_SUM, D_RING_READY_S_SUM, D_MARCH_HAS_SUM,   // ring lengths as the shader waves see them; lanes holding a job when a marcher polls
    D_MARCH_POLLS,
    D_CHUNK_TESTS, D_CHUNK_PASSED, D_CHUNK_LANES, D_SPHERE_CULLS, D_SPHERE_PASSED,   // wave-level: chunk / sphere cull tests made, not culled; lanes overlapping a passed chunk
    D_RESOLVE_MIXED, D_RESOLVE_MINORITY_LANES,   // trace_kernel: resolve phases holding both ray results and shadow results; lanes of the rarer kind in them
    D_PHOTON_CULL_VIOLATIONS,   // photon_kernel, counters build: roots outside DevScene::photon_cull_* that met a specular object all the same (must stay 0)
    C_COUNT
};

}  // namespace rtd
