/*
 * rt_hip.h — C ABI of the MI355X (gfx950) trace path.
 *
 * This is the drop-in boundary for the two calls the reference's render loops make
 * (reference src/main.cpp:18-24,59-63,100-104,142-146):
 *
 *     PhotonMapper::generate(scene);                 // reference src/photonmap.h:7
 *     const Picture p = raytracer.raytrace(scene);   // reference src/raytracer.h:9
 *
 * Everything crossing the boundary is plain C: POD structs, pointers and sizes.  No HIP, torch or
 * C++ types appear in any signature.  Every entry point returns RT_OK (0) or a negative rt_status;
 * none throws or aborts.  The message for the last failure on a context is rt_last_error(ctx)
 * (rt_last_error(NULL) for a failed rt_create).  Reference behaviour replaced: std::runtime_error
 * caught once in main (reference src/main.cpp:202-205).
 *
 * Threading: calls on one rt_ctx are serialised by the caller.  Different contexts (one per GPU)
 * may be driven from different host threads or processes; see rt_create for the one constraint on
 * the process's FIRST call.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 1

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARGUMENT = -1,
    RT_ERR_NO_DEVICE = -2,      /* HIP runtime/device missing: the product path never falls back to CPU */
    RT_ERR_HIP = -3,            /* a HIP call failed; text in rt_last_error */
    RT_ERR_NO_SCENE = -4,       /* render/photon call before rt_scene_upload */
    RT_ERR_LIMIT = -5           /* a size is beyond what the kernels were built for */
} rt_status;

/* ---- scene description (input of the boundary; reference src/scene.h:14-30 getters) ---------- */

/* Row-major 3x4 affine matrix, last row implied (0,0,0,1).  reference src/types.h:116-117 */
typedef struct rt_mat34 { float m[12]; } rt_mat34;

/* reference src/objects.h:165-205 (Camera) */
typedef struct rt_camera {
    rt_mat34 camera_to_world;   /* Matrix34::lookAt(position, lookAt, up), src/types.h:176-186 */
    float    fov;               /* radians; the trace path uses it as a HALF angle, src/raytracer.cpp:23-24 */
    uint32_t width, height;     /* resolution */
    uint32_t supersampling;     /* sub-pixels per axis (>=1) */
    uint32_t max_bounces;
    float    focus_distance;    /* |focusPoint - position|, src/objects.h:191 */
    float    lens_size;         /* 0 = pinhole */
} rt_camera;

enum { RT_LIGHT_PARALLEL = 0, RT_LIGHT_POINT = 1 };   /* reference src/objects.h:140-143 */
typedef struct rt_light {
    int32_t type;
    float   position[3];        /* direction for a parallel light, src/objects.h:155-156 */
    float   power[4];           /* RGBA */
} rt_light;

/* reference src/objects.h:18-31 (Material); the texture Picture is replaced by an index */
typedef struct rt_material {
    float   color[4];
    int32_t texture;            /* index into textures, -1 = none (then `color` is used) */
    float   ka, kd, ks, exponent;
    float   reflectance, transmittance;   /* on/off switches only, src/raytracer.cpp:151-158 */
    float   refraction_re, refraction_im; /* complex index of refraction */
    float   dispersion;
} rt_material;

/* RGBA8 texture; float texel = u8/255.0f (reference src/types.h:272) */
typedef struct rt_texture {
    uint32_t width, height;
    uint64_t offset;            /* byte offset of texel (0,0) in rt_scene_desc.texels; rows top-down */
} rt_texture;

/* reference src/objects.h:34-56 */
typedef struct rt_sphere {
    float    center[3];
    float    radius;
    rt_mat34 world_to_object, object_to_world, object_to_world_normals;
} rt_sphere;

/* reference src/objects.h:58-74; vertices are already in world space (src/wavefobj.cpp:82-106) */
typedef struct rt_triangle {
    float position[3][3];
    float normal[3][3];
    float uv[3][2];
} rt_triangle;

/* reference src/objects.h:76-102 */
typedef struct rt_julia {
    float    position[3];
    float    scale;
    float    c[4];              /* quaternion r,a,b,c */
    float    cut_plane;
    rt_mat34 world_to_object, object_to_world, object_to_world_normals;
} rt_julia;

enum { RT_OBJ_SPHERE = 0, RT_OBJ_TRIANGLE = 1, RT_OBJ_JULIA = 2 };
/* One entry per element of Scene::objects(), IN SCENE ORDER (ties between equally distant hits go
 * to the later object, reference src/raytracer.cpp:126-141). */
typedef struct rt_object {
    int32_t  kind;              /* RT_OBJ_* */
    uint32_t index;             /* into spheres / triangles / julias */
    uint32_t material;          /* into materials */
} rt_object;

typedef struct rt_scene_desc {
    uint32_t abi_version;       /* RT_ABI_VERSION */
    rt_camera camera;
    float    background[4];
    float    ambient[4];
    uint32_t n_lights;     const rt_light*    lights;
    uint32_t n_objects;    const rt_object*   objects;
    uint32_t n_spheres;    const rt_sphere*   spheres;
    uint32_t n_triangles;  const rt_triangle* triangles;
    uint32_t n_julias;     const rt_julia*    julias;
    uint32_t n_materials;  const rt_material* materials;
    uint32_t n_textures;   const rt_texture*  textures;
    uint64_t n_texel_bytes; const uint8_t*    texels;
    int32_t  dispersion_mode;   /* any(material.dispersion != 0), reference src/sceneparser.cpp:63-67 */
    /* RGB of HSVtoRGB(h,100,100) for h = 0,45,...,315.  DATA, not code: the reference's table
     * depends on how it is compiled (src/types.h:318, unqualified abs).  See rt_hue_table(). */
    float    hue_rgb[8][3];
    float    photon_steps;      /* <caustic steps>; 0 = no photon pass */
    uint32_t photon_texture_size;
    float    photon_factor;
} rt_scene_desc;

enum { RT_HUE_INT_ABS = 0, RT_HUE_FLOAT_ABS = 1 };
/* Fills one of the two named presets of the dispersion hue table (int-abs is what both pinned
 * reference builds produce). */
int rt_hue_table(int preset, float out_rgb[8][3]);

/* ---- work partition ------------------------------------------------------------------------- */

/* Horizontal bands of `band_rows` image rows; this call renders bands first_band, first_band +
 * band_stride, ...  The output buffer is COMPACT: the rendered bands back to back, each band
 * band_rows*width RGBA float pixels (the last band of the image may be shorter).  NULL = whole image.
 * No counterpart in the reference (its row scheduler is src/raytracer.cpp:48-53). */
typedef struct rt_bands {
    uint32_t band_rows;
    uint32_t first_band;
    uint32_t band_stride;
} rt_bands;

/* Number of RGBA float pixels rt_render writes for this partition of a width x height image. */
uint64_t rt_bands_pixel_count(uint32_t width, uint32_t height, const rt_bands* bands_or_null);

/* ---- counters ------------------------------------------------------------------------------- */

typedef struct rt_stats {
    uint64_t primary_rays;          /* depth-0 castRay calls */
    uint64_t secondary_rays;        /* reflection + refraction castRay calls */
    uint64_t shadow_rays;
    uint64_t photon_segments;       /* PhotonMapper::castRay calls */
    uint64_t photon_deposits;
    uint64_t sphere_tests;
    uint64_t triangle_tests;
    uint64_t julia_queries;         /* Julia::intersect calls */
    uint64_t julia_steps;           /* march steps == full-iteration DE calls */
    uint64_t julia_de_iterations;   /* quaternion iterations incl. the 6x8-capped normal probes */
    uint64_t julia_normals;
    uint64_t shaded_hits;           /* accepted nearest hits that were shaded */
    uint64_t phong_lights;          /* per-light Phong evaluations (each casts one shadow ray) */
    uint64_t fresnel_evals;
    uint64_t texture_lookups;
    uint64_t bvh_node_visits;
    double   kernel_ms;             /* device time of the last call's kernels (HIP events) */
    uint64_t shadow_rays_elided;    /* of shadow_rays: towards lights whose Phong terms can only be +-0 at that point (diffuse term:
                                       kd == 0 or the light is behind the surface; specular term: ks == 0 or the mirrored light
                                       direction points away from the viewer); a launch without counters does not trace them
                                       (RT_OPT_ELIDE_UNLIT_SHADOWS), a counters launch does */
    uint64_t photon_roots_culled;   /* of the photon pass's roots: light rays that pass every refracting or reflecting object by and
                                       so can leave nothing (RT_OPT_CULL_PHOTON_ROOTS); counted by a counters launch, which traces them */
} rt_stats;

/* ---- context -------------------------------------------------------------------------------- */

typedef struct rt_ctx rt_ctx;

/* rt_create makes the process's first HIP calls when nothing else has.  CONSTRAINT (of the HIP runtime, not of this
 * library; found with tools/hip_init_race.cpp, a host-only program with none of this library's code in it): while the
 * runtime starts up it unmaps a page it no longer owns, so a thread that maps memory at that moment — malloc of a
 * large buffer, a new thread's stack, a file mapping — can find a one-page hole in what it was just given (SIGSEGV on
 * first touch; 2 in 200 runs).  Make the first rt_create (or any other first HIP call) while the process's other
 * threads are not allocating, ideally before they exist; afterwards contexts may be created, used and destroyed from
 * any thread (calls made with the runtime up were never seen to disturb another thread's mappings: 300 runs).
 * rt_create calls are serialised among themselves. */
int  rt_create(rt_ctx** out_ctx, int device_id);
void rt_destroy(rt_ctx* ctx);
const char* rt_last_error(const rt_ctx* ctx_or_null);

/* Deep-copies the scene to the device (the caller keeps ownership of every pointer in `desc`),
 * builds the triangle BVH on the GPU and clears the photon atlas.  Replaces the `const Scene&`
 * argument of reference src/raytracer.h:9. */
int rt_scene_upload(rt_ctx* ctx, const rt_scene_desc* desc);

/* Options that are not part of the scene (never read from scene XML). */
enum {
    RT_OPT_COLLECT_STATS = 1,   /* 0/1: fill rt_stats counters (slower kernels); default 0 */
    RT_OPT_BVH_MIN_TRIANGLES = 2,/* BVH traversal for runs of consecutive triangle objects of at least this many (default 48);
                                    runs holding a zero-area triangle always keep the sequential scan */
    RT_OPT_DOF_SEED = 3,        /* seed of the counter-based lens RNG; default 0 */
    RT_OPT_PHOTON_ORDER = 4,    /* 1 (default): photon deposits are recorded, sorted and summed per texel in the order of the
                                   reference's sequential pass (src/photonmap.cpp:32-56, src/objects.cpp:244-256): textures
                                   bit-identical to the reference's.  0: float atomics, textures equal to rounding (fastest).
                                   rt_photon_generate_device on a scene whose deposits outgrow the record buffer adds the
                                   surplus with atomics and sizes the buffer for the next pass; rt_photon_generate reruns. */
    RT_OPT_PHOTON_RECORDS = 5,  /* capacity of that record buffer in deposits; 0 (default) = a sixteenth of the root photons,
                                   at least 262144; grows by itself as described above */
    RT_OPT_DEBUG_FORCE_ABORT = 6, /* test hook, 0/1: the Julia kernel's first workgroup raises its watchdog flag at once, as a wave
                                   stuck on the job queues would: the launch is then reported as abandoned (RT_ERR_HIP) */
    RT_OPT_ELIDE_UNLIT_SHADOWS = 7, /* 1 (default): a shadow ray towards a light that can only add +-0 to the radiance is not traced
                                   (the reference casts it and multiplies the result by zero, src/raytracer.cpp:214-235).  That is
                                   the case when BOTH Phong terms of that light vanish at the point: the diffuse one because
                                   kd == 0 or n.l <= 0 (light behind the surface), the specular one because ks == 0 or r.v <= 0
                                   with exponent > 0 (pow(0, e) = 0) — the back of every object as seen from the light, all of a
                                   mirror or of clear glass — and no factor is infinite or NaN (material, lights, point and
                                   normal are checked).  Same picture bit for bit.  rt_stats.shadow_rays still counts such
                                   rays, shadow_rays_elided says how many.  0: trace every shadow ray. */
    RT_OPT_CULL_PHOTON_ROOTS = 8,  /* 1 (default): a root of the photon pass (src/photonmap.cpp:38-58) whose ray misses the box around
                                   every object that refracts or reflects light rays is not traced: the reference deposits a light
                                   ray only after such a bounce (src/photonmap.cpp:79-83), so these roots leave nothing.  Same photon
                                   textures bit for bit; rt_stats.photon_segments of a counters launch still counts them and
                                   photon_roots_culled says how many.  0: trace every root. */
    RT_OPT_SHARED_GPU = 9          /* 0 (default): the GPU is this launch's own while a render runs.  1: kernels of other streams (a
                                   collective's, another library's) may sit on it when the render is dispatched.  The render is a
                                   grid sized to fill the GPU whose waves own a share of the tiles outright (no shared counter
                                   while those last); a workgroup that finds no room at dispatch starts when others END, and the
                                   frame then waits for its share: C2 1.67 -> 2.00 ms, C3 7.7 -> 9.2 ms beside eight small
                                   workgroups of another stream, however briefly they stay (tools/coresident_probe.py).  With 1
                                   a wave owns its first tile only and every other tile goes through the launch's counter: C2
                                   1.71 ms alone and 1.73-1.76 beside them (a frame with several samples per pixel keeps six sixteenths
                                   of its tiles owned instead of ten: C3 8.6-8.9 ms beside them).  Same picture bit for bit.  Frames of at most two tiles per wave (C5's, which is what
                                   several contexts per GPU are for) and the photon pass are handed out as before. */
};
int rt_set_option(rt_ctx* ctx, int option, int64_t value);

/* ≙ PhotonMapper::generate(Scene&)  (reference src/photonmap.cpp:16-59).  The photon textures stay
 * on the device and are read by the next rt_render.  No-op when photon_steps <= 0. */
int rt_photon_generate(rt_ctx* ctx, rt_stats* stats_or_null);

/* ≙ Picture RayTracer::raytrace(const Scene&)  (reference src/raytracer.cpp:10-15).  Synchronous.
 * `rgba_out` is caller-owned HOST memory of rt_bands_pixel_count()*4 floats, row-major RGBA, the
 * layout of Picture (reference src/types.h:385). */
int rt_render(rt_ctx* ctx, const rt_bands* bands_or_null, float* rgba_out, rt_stats* stats_or_null);

/* raytrace + Color::scaleOut for the writers: what the reference's four loops do with the Picture they get back is
 * quantise it (writePNG / writeAPNGFrame -> Color::scaleOut, reference src/pngwriter.cpp:205-216, src/types.h:274-282),
 * after accumulating sub-frames for motion blur (Picture::mulAdd, reference src/main.cpp:64,147).  These three do
 * that on the device, so that 4 bytes per pixel cross the host link instead of 16 (the float frame's copy
 * takes longer than the render):
 *   rt_render_rgba8          picture -> RGBA8 (u8(clamp(v*gain,0,1)*255.0f), truncating) into caller-owned host memory
 *   rt_render_accumulate     acc = (first ? 0 : acc) + picture * weight, element by element in the rounding order of
 *                            Picture::mulAdd; the accumulated picture stays on the device, inside the context
 *   rt_accumulated_rgba8     the accumulated picture -> RGBA8 into caller-owned host memory
 * Synchronous, default stream; `rgba8_out` holds rt_bands_pixel_count()*4 bytes. */
int rt_render_rgba8(rt_ctx* ctx, const rt_bands* bands_or_null, uint8_t* rgba8_out, float gain, rt_stats* stats_or_null);
int rt_render_accumulate(rt_ctx* ctx, const rt_bands* bands_or_null, float weight, int first);
int rt_accumulated_rgba8(rt_ctx* ctx, const rt_bands* bands_or_null, uint8_t* rgba8_out, float gain);

/* Same, but `d_rgba_out` is DEVICE memory on the context's GPU and the kernels are enqueued on
 * `hip_stream` (a hipStream_t passed as void*, NULL = the default stream) without synchronising.
 * This is the entry point the multi-GPU driver uses so that frames stay in HBM for the RCCL gather.
 * stats (if requested) are valid after the stream has been synchronised by the caller and
 * rt_fetch_stats is called. */
int rt_render_device(rt_ctx* ctx, const rt_bands* bands_or_null, void* d_rgba_out, void* hip_stream);

/* rt_render_device followed by rt_quantize_device in one: `d_rgba8_out` (DEVICE memory, rt_bands_pixel_count()*4 bytes)
 * receives Color::scaleOut of every pixel (reference src/types.h:274-282, as writePNG / writeAPNGFrame apply it with
 * gain 1: src/main.cpp:31,105) and the float picture is never written — where a pixel has one sample the trace kernel
 * quantises as it stores, otherwise the kernel that sums a pixel's samples does.  Same bytes as the two calls; a quarter
 * of the output traffic and one launch fewer (C2: 133 MB of float pixels and 24 us per frame).  Asynchronous on
 * `hip_stream` like rt_render_device; rt_launch_status applies in the same way. */
int rt_render_rgba8_device(rt_ctx* ctx, const rt_bands* bands_or_null, float gain, void* d_rgba8_out, void* hip_stream);
int rt_photon_generate_device(rt_ctx* ctx, void* hip_stream);
int rt_fetch_stats(rt_ctx* ctx, rt_stats* out);

/* rt_render_device for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  NO counterpart in the reference.  Same arguments, same picture bit for
 * bit; but the launch works in a fixed workspace — a file-scope __device__ array of RT_NOALLOC_SCRATCH_BYTES that
 * every GPU holds from the moment it loads the library until the process ends — instead of the context's own
 * buffers, and the call makes no hipMalloc, hipFree or other allocating call.  The workspace holds the work queue
 * words, the frame's sample buffer ([samples per pixel][pixels] float4, only when supersampling or dispersion give
 * more than one sample per pixel) and the per-lane ray-tree frame stacks; when the stacks of a full persistent grid
 * do not fit, fewer workgroups are launched (slower, same result).  RT_ERR_LIMIT with a message that names the
 * sizes when the sample buffer, or the stacks of a single workgroup, do not fit.  Counters are not collected
 * (RT_OPT_COLLECT_STATS is ignored).  The lists by which a frame with Julia sets starts its long samples first come off
 * the workspace's room for the frame stacks (0.2 MB at 960 x 600 x 4 samples); if that leaves no room for one workgroup's
 * stacks the samples go out in plain image order: same picture, slower.  ALL calls on one GPU share that GPU's workspace: two calls that overlap in
 * time on one GPU (two streams, two contexts, two threads) race.  rt_launch_status applies as for rt_render_device:
 * the watchdog word of the fixed workspace stays set until a context has seen it there, so an abandoned launch is
 * reported even when further launches of either entry point were enqueued before the status was asked for (and,
 * the word being shared like the rest of the workspace, by every context whose no-alloc Julia launch ran before then). */
#ifndef RT_NOALLOC_SCRATCH_BYTES
#define RT_NOALLOC_SCRATCH_BYTES (32u << 20)
#endif
int rt_render_device_noalloc(rt_ctx* ctx, const rt_bands* bands_or_null, void* d_rgba_out, void* hip_stream);
/* Waits for the last rt_render_device launch on this context and returns RT_ERR_HIP if the Julia kernel's watchdog
 * abandoned it (a wave waited too long on the job queues; the picture is incomplete), else RT_OK.  rt_render and
 * rt_fetch_stats make the same check themselves.  Callers of rt_render_device call this before they use the picture.
 * Replaces: nothing in the reference (its trace cannot stall; errors there are exceptions, src/main.cpp:202-205). */
int rt_launch_status(rt_ctx* ctx);

/* One photon pass split over several GPUs: emits only slice `part` of `n_parts` equal
 * contiguous slices of the root-photon sequence of PhotonMapper::generate (reference
 * src/photonmap.cpp:32-56: point lights in scene order, then phi, theta, hue) into this context's
 * zeroed photon textures.  The caller then SUMS the textures of all parts — rt_photon_atlas_device
 * gives the device buffer (all receivers' textures, contiguous floats) for an in-place all-reduce —
 * before any rt_render.  The pointer stays valid until the next rt_scene_upload or rt_destroy on this
 * context (an upload reallocates and clears the textures).  part 0 of 1 is rt_photon_generate_device.  The reference's pass is
 * single-threaded; the sum of the parts equals its textures up to the order of the float additions
 * (within a slice the deposits keep the reference's order, RT_OPT_PHOTON_ORDER). */
int rt_photon_generate_part_device(rt_ctx* ctx, void* hip_stream, uint32_t part, uint32_t n_parts);

/* Waits for the last photon pass enqueued on this context and reports whether ALL of its deposits were summed in
 * the reference's order (*ordered_out = 1), or not (0: RT_OPT_PHOTON_ORDER is off, the key did not fit 64 bits, or
 * the pass outgrew the record buffer and its surplus was added with atomics — the buffer has then been resized and
 * enqueueing the pass again gives the ordered result).  For callers of the asynchronous entry points that want the
 * guarantee rt_photon_generate gives by itself.  RT_ERR_HIP with a message if the ordered sum met a record or a run
 * outside its buffers (the kernels check every segment before they touch it; never seen with sound bookkeeping: an
 * internal error, the textures of that pass are incomplete) — rt_photon_generate reports the same.
 * No counterpart in the reference (its pass is sequential). */
int rt_photon_ordered(rt_ctx* ctx, int* ordered_out);
int rt_photon_atlas_device(rt_ctx* ctx, void** d_atlas_out, uint64_t* n_floats_out);

/* Test hook: copies the photon texture of scene object `object_index` (texture_size^2 RGBA floats;
 * zeros if the object is not a diffuse receiver).  ≙ Object::getPhoton, reference src/objects.cpp:257-267 */
int rt_photon_download(rt_ctx* ctx, uint32_t object_index, float* rgba_out);

/* ≙ Picture::mulAdd (reference src/types.h:372-376): d_acc[i] += d_src[i] * factor over n_floats
 * device floats, on `hip_stream`.  Motion-blur accumulation (reference src/main.cpp:64,147). */
int rt_mul_add_device(rt_ctx* ctx, void* d_acc, const void* d_src, float factor, uint64_t n_floats,
                      void* hip_stream);

/* ≙ Color::scaleOut (reference src/types.h:274-282): RGBA float -> RGBA8 by
 * u8(clamp(v*gain,0,1)*255.0f), truncating.  n_pixels pixels, device to device. */
int rt_quantize_device(rt_ctx* ctx, const void* d_rgba, void* d_rgba8, float gain, uint64_t n_pixels,
                       void* hip_stream);

/* ---- test hooks -------------------------------------------------------------------------------
 * One function of the path per call, for the known-answer parity tests (tests/test_gpu_parity.py: test_intersectors_against_reference_answers; vectors tests/golden/<name>.kat.bin).
 * rt_kat_intersect ≙ Object::intersect (reference src/objects.h:124-126): n rays of 7 floats
 * (origin, direction as given to the Ray constructor, max_distance) against scene object
 * object_index[i]; out = 12 floats per ray (hit flag, distance, point, normal, uv, photon uv).
 * rt_kat_fresnel ≙ calcFresnel (reference src/raytracer.cpp:243-268): n records of 5 floats
 * (refraction re, im, dispersion, cos, wavelength); out = kr per record. */
int rt_kat_intersect(rt_ctx* ctx, const uint32_t* object_index, const float* rays, uint32_t n, float* out);
int rt_kat_fresnel(rt_ctx* ctx, const float* records, uint32_t n, float* out);
/* Raw counter words of the last counters-enabled launch, incl. the wave-scheduler diagnostics
 * (developer tool; returns the number of words). */
int rt_debug_counters(rt_ctx* ctx, uint64_t* out, uint32_t cap);
/* The host builder of the BVH of short triangle runs (surface-area tree, no counterpart in the reference: it tests every
 * object for every ray, src/raytracer.cpp:126), without a device: n triangles of 9 floats (p0, e1, e2) in, n-1 records
 * of 16 words out (rt_scene_dev.h: DevBvhPair — boxes of both children, child words with bit 31 = leaf | triangle index).
 * Host-only (tests/test_bvh_host.py checks the tree's invariants on the CPU). */
int rt_debug_bvh_host(const float* triangles, uint32_t n, uint32_t* records_out);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
