// rt_api.hip — host side of the C ABI declared in include/rt_hip.h.
//
// Owns the device copy of the scene (layout: rt_scene_dev.h), precomputes on the host the few
// constants the reference computes once per frame (camera frame src/raytracer.cpp:17-29, the rayX
// row table :59/:111, the photon scan tables src/photonmap.cpp:38-45) and launches the kernels of
// rt_kernels.hip.  No CPU fallback exists: without a HIP device every entry point fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_scene_dev.h"

namespace rtd {
template <bool STATS, bool DENSE>
__global__ void trace_kernel(const DevScene sc, const DevJob job);
template <bool STATS>
__global__ void trace_julia_kernel(const DevScene sc, const DevJob job);
template <bool HAS_JULIA, bool STATS>
__global__ void photon_kernel(const DevScene sc, const DevPhotonJob job);
__global__ void photon_cull_kernel(const DevScene sc, const DevPhotonJob job, unsigned n_dirs, uint32_t* live_dirs, unsigned long long* live_count);
__global__ void sum_samples_kernel(const float4* samples, float4* out, unsigned long long n_pixels, unsigned samples_per_pixel, float inv_samples, uchar4* out8, float gain);
__global__ void kat_intersect_kernel(const DevScene sc, const unsigned* kind, const unsigned* prim, const float* rays, unsigned n, float* out);
__global__ void kat_fresnel_kernel(const float* in, unsigned n, float* out);
__global__ void mul_add_kernel(float* acc, const float* src, float factor, unsigned long long n);
__global__ void quantize_kernel(const float4* rgba, uchar4* out, float gain, unsigned long long n);
hipError_t build_triangle_bvh(const DevTriTest* tris, uint32_t first, uint32_t n, DevBvhPair* pairs);
void build_triangle_bvh_host(const DevTriTest* tris, uint32_t first, uint32_t n, DevBvhPair* pairs);
hipError_t photon_order_temp_bytes(unsigned capacity, unsigned key_bits, size_t* bytes);
hipError_t photon_order_bucket(const unsigned long long* keys, const float4* vals, const unsigned* slot, unsigned capacity, unsigned texel_shift, unsigned n_texels,
                               float4* atlas, unsigned* count, unsigned* seg_first, uint4* runs, unsigned long long* seg_keys, float4* seg_vals,
                               unsigned* order, unsigned long long* cursor, hipStream_t stream);
hipError_t photon_order_reduce(const unsigned long long* keys, const float4* vals, unsigned capacity, unsigned key_bits, unsigned texel_shift,
                               unsigned long long n_texels, float4* atlas, unsigned long long* sorted_keys, unsigned* sorted_index,
                               unsigned* long_runs, unsigned long long* words, void* temp, size_t temp_bytes, hipStream_t stream);
constexpr int kFrameFieldsHost = 11;
#ifndef RT_BVH_HOST_BUILD_MAX
#define RT_BVH_HOST_BUILD_MAX 1024
#endif
constexpr uint32_t kBvhHostBuildMax = RT_BVH_HOST_BUILD_MAX;   // runs up to this long get the host-built surface-area tree (rt_bvh.hip), longer ones the GPU-built Morton tree
constexpr uint32_t kBvhLdsStackEntries = 16;   // 16 KB per workgroup of 256 lanes: with the 36 KB of frame tops and parked node state, 3 blocks per CU still fit
// dynamic LDS of trace_kernel / photon_kernel: [copy of the shade blob, if small][BVH stack columns, if the scene has a BVH]
static size_t dynamic_lds_bytes(const rtd::DevScene& sc) {
    return (sc.shade_blob_in_lds ? ((size_t(sc.shade_blob_bytes) + 15) & ~size_t(15)) : 0) + size_t(sc.bvh_lds_stack_entries) * 256 * sizeof(uint32_t) +
           (sc.frame_slots_in_lds == 2 ? size_t(kFrameFieldsHost) * 256 * sizeof(float) : 0) + size_t(sc.bvh_records_in_lds) * sizeof(rtd::DevBvhPair);
}
constexpr size_t kTraceStaticLdsBytes = (size_t(kFrameFieldsHost) + 26) * 256 * sizeof(float);   // first frame slot + parked state (rt_kernels.hip)
constexpr size_t kLdsPerBlockFor3 = 160 * 1024 / RT_TRACE_WAVES_PER_SIMD - 512;                  // RT_TRACE_WAVES_PER_SIMD workgroups per CU
constexpr uint64_t kDenseMinSamples = 1ull << 20;     // trace_kernel<.., DENSE> from a million samples per launch
#ifndef RT_MAX_LDS_BLOB
#define RT_MAX_LDS_BLOB (16 * 1024)
#endif
constexpr size_t kMaxLdsBlobBytes = RT_MAX_LDS_BLOB;   // + 11 KB of frame tops + 25 KB of parked node state: 3 blocks per CU within the 160 KB LDS
constexpr int kFrameRecordWordsHost = 12;       // trace_julia_kernel keeps every frame as a 48-byte record of its own
constexpr int kJuliaBlockThreads = kJuliaWaves * 64, kJuliaShaderWavesHost = kJuliaWaves - kJuliaMarcherWaves;
}  // namespace rtd

using namespace rtd;

namespace {

thread_local std::string g_create_error;

struct Buffer {
    void* ptr = nullptr;
    size_t bytes = 0;
};

// Device memory a render launch works in besides the scene and its output.
struct Workspace {
    unsigned long long* queue = nullptr;      // [0] queue head, [1] watchdog flag of trace_julia_kernel
    unsigned long long* counters = nullptr;   // C_COUNT words
    float* stack = nullptr;                   // per-lane DFS frames
    size_t stack_bytes = 0;                   // capacity
    float4* samples = nullptr;                // [samples_per_pixel][pixels] when samples_per_pixel > 1
    size_t samples_bytes = 0;                 // capacity
    uint32_t* sched = nullptr;                // DevJob::sched (Julia scenes with several samples per pixel), or NULL
};

// ---- rt_render_device_noalloc: the workspace of the entry point that allocates nothing --------------------------
// One array per GPU that loads this library, held until the process ends.  Host code never passes `scratch` itself
// to a kernel or a memset (its name is a host-side stand-in, not a device address): rt_create asks the runtime for
// the array's address on ITS device (hipGetSymbolAddress) and every use goes through that pointer.
#ifndef RT_JULIA_HOT_STEPS
#define RT_JULIA_HOT_STEPS 192         // DevJob::hot_steps: measured on C4; 0 = plain order
#endif
#ifndef RT_JULIA_GROUP_TILES
#define RT_JULIA_GROUP_TILES 2
#endif
constexpr size_t kNoAllocScratchBytes = size_t(RT_NOALLOC_SCRATCH_BYTES);
constexpr size_t kNoAllocHeaderBytes = 4096;   // queue words at 0, counters at 256
static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];
// the counter words, padded so that clearing them is one fill kernel (hipMemsetAsync splits a size that is no multiple of
// its wide store into a body and a tail: two launches, 5 us each, per render and per photon pass)
constexpr size_t kCounterBytes = (C_COUNT * sizeof(unsigned long long) + 255) & ~size_t(255);
static_assert(kCounterBytes + 256 <= kNoAllocHeaderBytes, "scratch header");

}  // namespace

struct rt_ctx {
    int device = 0;
    int n_cus = 0;
    std::string error;
    bool has_scene = false;
    DevScene dev{};
    // the uploaded scene's tables live in ONE device allocation that is kept and reused from upload to upload (a
    // frame of an animation re-uploads a scene of the same size 20 x K times): tables are packed into a pinned host
    // block and go over with a single asynchronous copy; nothing is freed or allocated unless the scene outgrows it
    Buffer scene_arena, bvh_nodes, atlas;
    unsigned char* h_arena = nullptr;      // pinned staging
    size_t h_arena_bytes = 0;
    std::vector<unsigned char> stage;      // tables of the upload in progress
    struct PendingPtr { size_t offset; const void** slot; };
    std::vector<PendingPtr> pending;
    hipEvent_t ev_upload = nullptr;        // the last upload's copy has left the pinned block
    bool upload_in_flight = false;
    // host-side facts about the uploaded scene
    uint32_t n_objects = 0, n_slots = 0;
    std::vector<int32_t> photon_slot;
    std::vector<uint32_t> object_kind, object_prim;
    std::vector<uint32_t> point_lights;
    float photon_steps = 0.0f;
    uint32_t hues = 1;
    // photon tables
    const float* d_photon_tables = nullptr;   // sin/cos of phi, sin/cos of theta (in the scene arena)
    uint32_t n_phi = 0, n_theta = 0;
    const uint32_t* d_point_lights = nullptr;
    // options
    bool collect_stats = false;
    int64_t bvh_min_triangles = 48;   // measured: C2 (one run of 64 triangles) 1.95 ms with the BVH against 2.04 ms with the chunk-culled scan;
                                      // C5 (20 triangles) 0.90 against 0.79 ms, so short runs keep the scan
    uint32_t dof_seed = 0;
    bool photon_order = true;          // RT_OPT_PHOTON_ORDER: deposits summed in the reference's order
    // ordered photon deposits (rt_photon_order.hip)
    Buffer dep_keys, dep_vals, dep_sorted_keys, dep_sorted_index, dep_long_runs, dep_temp;
    Buffer dep_count, dep_seg_first, dep_runs, dep_seg_vals;   // photon_order_bucket: deposits per texel (zero between passes), first record of a texel's segment, run list
    bool dep_bucket = false;                     // the pass in preparation sums by buckets (atlases below 2^31 texels)
    bool dense_with_bvh = false;                 // RT_DENSE_BVH=1 in the environment (developer A/B): the dense variant for BVH scenes too
    bool dep_force_sort = false;                 // RT_PHOTON_ORDER_SORT=1 in the environment: the device-wide sort instead (developer A/B, tests)
    Buffer live_dirs;                            // RT_OPT_CULL_PHOTON_ROOTS: the directions of the photon pass that may meet a specular object
    uint32_t dep_capacity = 0;         // records; 0 = choose from the number of roots
    size_t dep_temp_bytes = 0;
    unsigned long long* d_dep_words = nullptr;   // [0] records reserved, [1] deposits that found the buffer full, [2] long runs, [3] deposits beyond a root's key field, [4] live directions (photon_cull_kernel), [5] photon_order_bucket's cursor, [6] records or runs it found outside the record buffer (internal error), [8..10] developer build RT_GUARD: first index violation (site, index, limit), [16] the pass's queue head
    unsigned long long* h_dep_words = nullptr;   // pinned copy, written by the stream after every ordered pass
    hipEvent_t ev_dep = nullptr;
    bool dep_check_pending = false;
    bool dep_last_pass_ordered = false;
    bool dep_last_pass_overflowed = false;   // sticky until the next pass is enqueued
    bool dep_key_too_narrow = false;         // a root made more deposits than the order key has bits for
    bool dep_misplaced = false;              // the bucket kernels met a segment outside the record buffer (never seen with sound counts: an internal error)
    bool poison_allocations = false;         // RT_POISON_ALLOC=1 when the context was created
    // per-call scratch, grown on demand and kept
    Buffer samples, stack, out_staging, out8, acc, kat_a, kat_b, kat_c, kat_d, sched;
    uint32_t julia_group_tiles = RT_JULIA_GROUP_TILES;   // DevJob::group_tiles; RT_JULIA_GROUP_TILES in the environment overrides it (developer sweeps)
    uint32_t julia_hot_steps = RT_JULIA_HOT_STEPS;   // DevJob::hot_steps (0: plain order); RT_JULIA_HOT_STEPS in the environment overrides it (developer sweeps)
    unsigned char* h_out = nullptr;        // pinned block the synchronous entry points copy their pictures through
    size_t h_out_bytes = 0;
    unsigned long long* d_queue = nullptr;      // [0] = queue head, [1] = watchdog flag of trace_julia_kernel (sticky until read)
    unsigned long long* h_status = nullptr;     // pinned copies of the watchdog words, written by the stream after every Julia launch:
                                                // [0] d_queue[1] (the context's own workspace), [1] the fixed workspace's (rt_render_device_noalloc)
    unsigned status_pending = 0;                // bit s: a Julia launch whose watchdog word h_status[s] has not been looked at yet
    bool debug_force_abort = false;             // RT_OPT_DEBUG_FORCE_ABORT
    bool elide_unlit_shadows = true;            // RT_OPT_ELIDE_UNLIT_SHADOWS
    bool cull_photon_roots = true;              // RT_OPT_CULL_PHOTON_ROOTS
    int static_share = -1;                      // sixteenths of a render's tiles that its waves own outright; < 0: the measured defaults
                                                // (RT_STATIC_SHARE_*); RT_STATIC_SHARE in the environment: developer sweeps
    bool shared_gpu = false;                    // RT_OPT_SHARED_GPU: a wave owns one tile at most (its first: 3 072 waves asking the counter
                                                // at once is what the share saves a GPU of one's own; none at all: 1.73 ms alone instead of 1.71)
    unsigned long long* status_flag[2] = {nullptr, nullptr};   // device words the pending statuses were copied from (each cleared once seen)
    unsigned char* noalloc_scratch = nullptr;   // address of `scratch` on this context's device (hipGetSymbolAddress)
    // persistent grid of the render kernels for the uploaded scene ([0] plain, [1] counters build): asked once per upload
    int trace_blocks_per_cu[2] = {0, 0};
    int dense_blocks_per_cu[2] = {0, 0};       // trace_kernel<.., DENSE>: 4 when the scene allows it, else 0 (never used)
    int julia_blocks_per_cu[2] = {0, 0};
    unsigned long long* d_counters = nullptr;   // rt_stats words (kCounterBytes)
    bool counters_clean = false;                // all zero, and nothing enqueued that writes them: only counters launches (RT_OPT_COLLECT_STATS) do
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    bool events_pending = false;
};

namespace {

int fail(rt_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->error = buf; else g_create_error = buf;
    return code;
}

#define RT_HIP_TRY(ctx, expr)                                                                          \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(ctx, RT_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));  \
    } while (0)

// RT_HIP_SYNC_EACH=1 (developer aid): wait after every kernel of a photon pass and name the one that failed, instead of
// hearing of a fault at the next synchronisation with no kernel attached to it.
int step_check(rt_ctx* ctx, hipStream_t stream, const char* what) {
    static const bool sync_each = [] { const char* e = std::getenv("RT_HIP_SYNC_EACH"); return e != nullptr && e[0] != '0' && e[0] != 0; }();
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && sync_each) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return fail(ctx, RT_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return RT_OK;
}

int grow(rt_ctx* ctx, Buffer& b, size_t bytes) {
    if (b.bytes >= bytes && b.ptr) return RT_OK;
    if (b.ptr) RT_HIP_TRY(ctx, hipFree(b.ptr));
    b.ptr = nullptr;
    b.bytes = 0;
    RT_HIP_TRY(ctx, hipMalloc(&b.ptr, bytes));
    b.bytes = bytes;
    // RT_POISON_ALLOC=1 (tests): a new buffer starts full of 0xA5 bytes, so that code which counts on fresh memory being
    // zero fails every time instead of when the allocator happens to return a used block
    // (hipMemset may return before the fill has run, and the caller's stream need not wait for the default stream: wait here)
    if (ctx->poison_allocations) {
        RT_HIP_TRY(ctx, hipMemset(b.ptr, 0xA5, bytes));
        RT_HIP_TRY(ctx, hipDeviceSynchronize());
    }
    return RT_OK;
}

// Places a table in the upload's staging block; *out receives its device address in flush_uploads.
template <typename T>
int upload(rt_ctx* ctx, const std::vector<T>& host, const T** out) {
    *out = nullptr;
    const size_t bytes = (host.empty() ? 1 : host.size()) * sizeof(T);
    const size_t offset = (ctx->stage.size() + 255) & ~size_t(255);
    ctx->stage.resize(offset + bytes, 0);
    if (!host.empty()) std::memcpy(ctx->stage.data() + offset, host.data(), host.size() * sizeof(T));
    ctx->pending.push_back({offset, reinterpret_cast<const void**>(out)});
    return RT_OK;
}

// Sends the staged tables with one copy on the default stream and resolves the pending pointers.
int flush_uploads(rt_ctx* ctx) {
    const size_t bytes = ctx->stage.size();
    if (bytes == 0) return RT_OK;
    if (ctx->scene_arena.bytes < bytes) {
        // the scene outgrew the arena: only now is anything freed, so only now must the device be idle
        RT_HIP_TRY(ctx, hipDeviceSynchronize());
        const int rc = grow(ctx, ctx->scene_arena, bytes + bytes / 4);
        if (rc != RT_OK) return rc;
    }
    if (ctx->upload_in_flight) {
        RT_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_upload));   // the pinned block is free again
        ctx->upload_in_flight = false;
    }
    if (ctx->h_arena_bytes < bytes) {
        if (ctx->h_arena) RT_HIP_TRY(ctx, hipHostFree(ctx->h_arena));
        ctx->h_arena = nullptr;
        ctx->h_arena_bytes = 0;
        RT_HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->h_arena), bytes + bytes / 4));
        ctx->h_arena_bytes = bytes + bytes / 4;
    }
    std::memcpy(ctx->h_arena, ctx->stage.data(), bytes);
    RT_HIP_TRY(ctx, hipMemcpyAsync(ctx->scene_arena.ptr, ctx->h_arena, bytes, hipMemcpyHostToDevice, nullptr));
    RT_HIP_TRY(ctx, hipEventRecord(ctx->ev_upload, nullptr));
    ctx->upload_in_flight = true;
    for (const rt_ctx::PendingPtr& p : ctx->pending) *p.slot = static_cast<const unsigned char*>(ctx->scene_arena.ptr) + p.offset;
    ctx->pending.clear();
    ctx->stage.clear();
    return RT_OK;
}

void free_scene(rt_ctx* ctx) {
    ctx->pending.clear();
    ctx->stage.clear();
    ctx->has_scene = false;
}

// The dense variant of trace_kernel (four workgroups per CU) for a scene with a BVH: what is left of the CU's LDS beside
// the parked state and the frame tops is two stack entries per lane; shading tables and BVH records are read through the
// vector cache.
void dense_bvh_layout(DevScene& sc) {
    sc.shade_blob_in_lds = 0;
    sc.bvh_records_in_lds = 0;
    sc.bvh_lds_stack_entries = 2;
}

// tiles / samples of one render call
void make_job(const rt_ctx* ctx, const rt_bands* bands, DevJob* job) {
    const DevScene& sc = ctx->dev;
    rt_bands b = {sc.height ? sc.height : 1u, 0u, 1u};
    if (bands) b = *bands;
    std::memset(job, 0, sizeof *job);
    job->band_rows = b.band_rows;
    job->first_band = b.first_band;
    job->band_stride = b.band_stride;
    job->n_pixels = rt_bands_pixel_count(sc.width, sc.height, &b);
    job->compact_rows = sc.width ? static_cast<uint32_t>(job->n_pixels / sc.width) : 0;
    job->tiles_x = (sc.width + 7) / 8;
    job->tiles_y = (job->compact_rows + 7) / 8;
    job->samples_per_pixel = sc.supersampling * sc.supersampling * (sc.dispersion ? 8u : 1u);
    job->n_samples = static_cast<uint64_t>(job->tiles_x) * job->tiles_y * job->samples_per_pixel * 64ull;
    job->dof_seed = ctx->dof_seed;
    job->fast_decode = (job->n_samples >> 6) <= 0xffffffffull ? 1u : 0u;
    job->div_samples = make_div(std::max<uint32_t>(1u, job->samples_per_pixel));
    job->div_tiles_x = make_div(std::max<uint32_t>(1u, job->tiles_x));
    job->div_band_rows = make_div(std::max<uint32_t>(1u, job->band_rows));
    job->div_supersampling = make_div(std::max<uint32_t>(1u, sc.supersampling));
}

// Ids a wave reserves per atomic: at least 16 reservations per wave for balance, whole 8x8 tiles.
uint32_t id_chunk_for(uint64_t items, uint64_t lanes) {
    const uint64_t per_wave = items / std::max<uint64_t>(1, lanes / 64);
    uint64_t c = (per_wave / 16) & ~uint64_t(63);
    if (c < 64) c = 64;
    if (c > 1024) c = 1024;
    return static_cast<uint32_t>(c);
}
// 64-id blocks every wave owns outright (see fetch_ids): `sixteenths`/16 of an even split.
uint32_t static_blocks_per_wave(uint64_t items, uint64_t lanes, unsigned sixteenths) {
    const uint64_t waves = std::max<uint64_t>(1, lanes / 64);
    return static_cast<uint32_t>(std::min<uint64_t>(items / 64 * sixteenths / 16 / waves, 0x7fffffffull));
}
#ifndef RT_STATIC_SHARE_TRACE
#define RT_STATIC_SHARE_TRACE 2     // C2 (bench, 40 steps): 1.915 ms at 1, 1.878 at 2, 1.884 at 3, 1.905 at 4, 2.16 at 5
#endif
// with more than one sample per pixel (measured with the round-2 kernels: C3 9.21 ms at 4/16, 8.91 at 6, 8.90 at 8, 8.83 at 10,
// 9.6 at 12; 6_supersampling 0.43 / 0.38 / 0.35 / 0.31 / 0.28; 7_dof 2.17 / 2.08 / 2.00 / 1.97 / 2.06; C2, one sample per pixel,
// 2.04 at 4 and 2.40 at 8)
#ifndef RT_STATIC_SHARE_TRACE_SUPERSAMPLED
#define RT_STATIC_SHARE_TRACE_SUPERSAMPLED 10
#endif
#ifndef RT_STATIC_SHARE_JULIA
#define RT_STATIC_SHARE_JULIA 4
#endif
#ifndef RT_STATIC_SHARE_PHOTON
#define RT_STATIC_SHARE_PHOTON 14
#endif

template <typename K>
int persistent_grid(rt_ctx* ctx, K kernel, uint64_t work_items, int* blocks, size_t lds_bytes = 0) {
    int per_cu = 0;
    RT_HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds_bytes));
    if (per_cu < 1) per_cu = 1;
    uint64_t g = static_cast<uint64_t>(ctx->n_cus) * per_cu;
    const uint64_t needed = (work_items + 255) / 256;
    if (g > needed) g = needed;
    if (g < 1) g = 1;
    *blocks = static_cast<int>(g);
    return RT_OK;
}

// Clears the counter words for a launch.  The context's own are cleared only when a counters launch has written them
// since, or is about to: the product kernels never touch them (three fill kernels fewer per frame of C5).
int clear_counters(rt_ctx* ctx, unsigned long long* counters, bool stats, hipStream_t stream) {
    const bool own = counters == ctx->d_counters;
#ifndef RT_PHASE_TIMING      // (that developer build adds its clocks to the words from every launch)
    if (own && ctx->counters_clean && !stats) return RT_OK;
#endif
    RT_HIP_TRY(ctx, hipMemsetAsync(counters, 0, kCounterBytes, stream));
    if (own) ctx->counters_clean = !stats;
    return RT_OK;
}

// Enqueues one render.  `fixed` == NULL: the context's own buffers, grown on demand (hipMalloc).  Otherwise the launch
// is fitted into the given workspace and nothing is allocated: fewer persistent workgroups when the frame stacks of
// a full grid do not fit, RT_ERR_LIMIT when not even one workgroup or the sample buffer fits.
// d_out8 != NULL: the quantised picture (Color::scaleOut with `gain`) is written instead of the float one — by the trace
// kernel itself where a pixel has one sample, by sum_samples_kernel otherwise; d_out is not touched then.
int render_enqueue(rt_ctx* ctx, const rt_bands* bands, float4* d_out, hipStream_t stream, const Workspace* fixed = nullptr, uchar4* d_out8 = nullptr, float gain = 1.0f) {
    if (!ctx->has_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_render before rt_scene_upload");
    if (bands && (bands->band_rows == 0 || bands->band_stride == 0)) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_bands: band_rows and band_stride must be > 0");
    DevJob job;
    make_job(ctx, bands, &job);
    if (job.n_pixels == 0) return RT_OK;
    const bool julia = ctx->dev.n_julias > 0;
    const bool stats = ctx->collect_stats && fixed == nullptr;
    int blocks = 0;
    int rc;
    DevScene dev = ctx->dev;
    if (julia) dev.shade_blob_in_lds = 0;     // trace_julia_kernel's LDS belongs to its ray pool
    // the dense variant of trace_kernel (4 waves per SIMD) where the scene fits four workgroups per CU and the frame has
    // the samples to occupy them several times over (measured: C3, 33 M samples, gains 8 %; C5, 90 k samples, loses 7 %)
    const bool dense = !julia && ctx->dense_blocks_per_cu[stats] >= 4 && job.n_samples >= kDenseMinSamples;
    if (dense) dev.frame_slots_in_lds = 1;
    if (dense && dev.bvh_nodes != nullptr) dense_bvh_layout(dev);
    const size_t lds = julia ? 0 : dynamic_lds_bytes(dev);
    const uint64_t block_lanes = julia ? kJuliaPoolContexts : 256;   // ray trees in flight per workgroup
    {
        const int per_cu = julia ? std::min(1, ctx->julia_blocks_per_cu[stats]) : dense ? ctx->dense_blocks_per_cu[stats] : ctx->trace_blocks_per_cu[stats];
        if (per_cu < 1) return fail(ctx, RT_ERR_HIP, "%s does not fit a compute unit", julia ? "trace_julia_kernel" : "trace_kernel");
        const uint64_t needed = (job.n_samples + block_lanes - 1) / block_lanes;
        blocks = static_cast<int>(std::max<uint64_t>(1, std::min<uint64_t>(static_cast<uint64_t>(ctx->n_cus) * per_cu, needed)));
    }
    const size_t frame_bytes_per_lane = static_cast<size_t>(ctx->dev.max_bounces + 1) * (julia ? kFrameRecordWordsHost : kFrameFieldsHost) * sizeof(float);
    const size_t samples_bytes = job.samples_per_pixel > 1 ? job.n_pixels * job.samples_per_pixel * sizeof(float4) : 0;
    // Julia scenes with several samples per pixel: sample-major blocks and the hot lists (DevJob::sched)
    job.n_tiles = job.tiles_x * job.tiles_y;
    size_t sched_zero_words = 0, sched_list_words = 0;
    job.group_tiles = std::max<uint32_t>(1u, ctx->julia_group_tiles);
    job.n_groups = (job.n_tiles + job.group_tiles - 1) / job.group_tiles;
    if (julia && job.samples_per_pixel > 1 && ctx->julia_hot_steps != 0u) {
        const uint64_t n_blocks = static_cast<uint64_t>(job.n_groups) * job.samples_per_pixel;
        if (n_blocks < 0x7fffffffull) {
            sched_zero_words = kSchedHeaderWords + job.n_groups + static_cast<size_t>((n_blocks + 31) / 32);
            sched_list_words = static_cast<size_t>(kSchedLevels) * job.n_groups * (job.samples_per_pixel - 1);
        }
    }
    const size_t sched_bytes = ((sched_zero_words + sched_list_words) * sizeof(uint32_t) + 255) & ~size_t(255);
    Workspace ws;
    if (fixed != nullptr) {
        ws = *fixed;
        // the lists come off the front of what the fixed workspace has for the frame stacks (C4: 0.2 MB of its 32), unless
        // that leaves no room for one workgroup's stacks: then the samples go out in plain order, same picture
        if (sched_bytes > 0 && ws.stack_bytes >= sched_bytes + block_lanes * frame_bytes_per_lane) {
            ws.sched = reinterpret_cast<uint32_t*>(ws.stack);
            ws.stack = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(ws.stack) + sched_bytes);
            ws.stack_bytes -= sched_bytes;
        }
        if (samples_bytes > ws.samples_bytes)
            return fail(ctx, RT_ERR_LIMIT, "rt_render_device_noalloc: the sample buffer of this frame (%zu bytes: %llu pixels x %u samples) exceeds the %zu bytes the fixed workspace has for it",
                        samples_bytes, static_cast<unsigned long long>(job.n_pixels), job.samples_per_pixel, ws.samples_bytes);
        const uint64_t fit = ws.stack_bytes / (block_lanes * frame_bytes_per_lane);
        if (fit < 1)
            return fail(ctx, RT_ERR_LIMIT, "rt_render_device_noalloc: the frame stacks of one workgroup (%zu bytes at max_bounces %u) exceed the fixed workspace",
                        static_cast<size_t>(block_lanes * frame_bytes_per_lane), ctx->dev.max_bounces);
        if (static_cast<uint64_t>(blocks) > fit) blocks = static_cast<int>(fit);
    } else {
        ws.queue = ctx->d_queue;
        ws.counters = ctx->d_counters;
    }
    const uint64_t lanes = static_cast<uint64_t>(blocks) * block_lanes;
    if (fixed == nullptr) {
        rc = grow(ctx, ctx->stack, lanes * frame_bytes_per_lane);
        if (rc != RT_OK) return rc;
        ws.stack = static_cast<float*>(ctx->stack.ptr);
        if (samples_bytes > 0) {
            rc = grow(ctx, ctx->samples, samples_bytes);
            if (rc != RT_OK) return rc;
            ws.samples = static_cast<float4*>(ctx->samples.ptr);
        }
    }
    if (job.samples_per_pixel > 1) job.samples = ws.samples;
    if (fixed == nullptr && sched_bytes > 0) {
        rc = grow(ctx, ctx->sched, sched_bytes);
        if (rc != RT_OK) return rc;
        ws.sched = static_cast<uint32_t*>(ctx->sched.ptr);
    }
    job.sched = ws.sched;
    job.hot_steps = ctx->julia_hot_steps;
    job.out = d_out8 != nullptr ? nullptr : d_out;
    job.out8 = d_out8;
    job.out8_gain = gain;
    job.id_chunk = 1;   // see fetch_ids
    // waves that draw sample ids: every wave of trace_kernel, the shader waves of trace_julia_kernel
    const uint64_t id_lanes = julia ? static_cast<uint64_t>(blocks) * kJuliaShaderWavesHost * 64 : lanes;
    job.n_waves = static_cast<uint32_t>(id_lanes / 64);
    // a quarter of the tiles is owned outright (measured on C2, whose 16 MB texture likes the waves to
    // sweep the image together: 0/16 2.24 ms, 4/16 2.16, 8/16 2.60, 12/16 2.86; C3: 12.3, 11.5, 11.2, 11.5);
    // a launch of at most two tiles per wave needs no counter at all (C1: 0.10 -> 0.06 ms)
    const bool tiny = job.n_samples / 64 <= 2 * (id_lanes / 64);
    job.static_blocks = tiny ? static_cast<uint32_t>((job.n_samples / 64 + id_lanes / 64 - 1) / (id_lanes / 64))
                             : static_blocks_per_wave(job.n_samples, id_lanes, ctx->static_share >= 0 ? static_cast<unsigned>(ctx->static_share)
                                                      : julia ? RT_STATIC_SHARE_JULIA : job.samples_per_pixel > 1 ? RT_STATIC_SHARE_TRACE_SUPERSAMPLED : RT_STATIC_SHARE_TRACE);
    if (job.sched != nullptr) job.static_blocks = 0;   // every block goes through the counter (and the claim bits)
    // RT_OPT_SHARED_GPU (include/rt_hip.h; tools/coresident_probe.py).  One sample per pixel: a late workgroup owes one tile.
    // Several: the samples of a pixel like to be rendered together and the share is worth more than the tail costs — beside
    // the probe's stand-ins C3 takes 9.0-9.2 ms at the default 10/16, 8.64-8.87 at 6/16 (4/16: 8.76-8.98, 2/16: 9.1-9.3,
    // one tile: 9.5-9.7)
    if (ctx->shared_gpu && !tiny && !julia && ctx->static_share < 0) {
        if (job.samples_per_pixel == 1) job.static_blocks = std::min<uint32_t>(job.static_blocks, 1u);
        else job.static_blocks = std::min<uint32_t>(job.static_blocks, static_blocks_per_wave(job.n_samples, id_lanes, 6));
    }
    job.dyn_base = static_cast<uint64_t>(job.n_waves) * job.static_blocks * 64ull;
    job.stack = ws.stack;
    job.queue_head = ws.queue;
    job.counters = ws.counters;
    job.debug_abort = ctx->debug_force_abort ? 1u : 0u;
    job.elide_unlit = (ctx->elide_unlit_shadows && !stats) ? 1u : 0u;
    if (ctx->upload_in_flight && stream != nullptr) RT_HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->ev_upload, 0));   // the scene's copy ran on the default stream
    // Both workspaces (the context's own and the fixed one of the no-alloc entry point, which every call on this GPU
    // shares) keep the last call's contents: the queue head [0] and the counters start at zero, the frame stacks and the
    // sample buffer are written before they are read.  Word [1], the watchdog flag, is sticky: cleared only once a host
    // has seen it (launch_status), so that two launches without rt_launch_status in between do not lose the first one's
    // abandonment
    RT_HIP_TRY(ctx, hipMemsetAsync(ws.queue, 0, sizeof(unsigned long long), stream));
    if (const int rc = clear_counters(ctx, ws.counters, stats, stream); rc != RT_OK) return rc;
    if (job.sched != nullptr) {
        RT_HIP_TRY(ctx, hipMemsetAsync(job.sched, 0, sched_zero_words * sizeof(uint32_t), stream));
        RT_HIP_TRY(ctx, hipMemsetAsync(job.sched + sched_zero_words, 0xff, sched_list_words * sizeof(uint32_t), stream));
    }
    RT_HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, stream));
    if (julia) {
        if (stats) hipLaunchKernelGGL((trace_julia_kernel<true>), dim3(blocks), dim3(kJuliaBlockThreads), lds, stream, dev, job);
        else hipLaunchKernelGGL((trace_julia_kernel<false>), dim3(blocks), dim3(kJuliaBlockThreads), lds, stream, dev, job);
    } else {
        if (dense) {
            if (stats) hipLaunchKernelGGL((trace_kernel<true, true>), dim3(blocks), dim3(256), lds, stream, dev, job);
            else hipLaunchKernelGGL((trace_kernel<false, true>), dim3(blocks), dim3(256), lds, stream, dev, job);
        } else {
            if (stats) hipLaunchKernelGGL((trace_kernel<true, false>), dim3(blocks), dim3(256), lds, stream, dev, job);
            else hipLaunchKernelGGL((trace_kernel<false, false>), dim3(blocks), dim3(256), lds, stream, dev, job);
        }
    }
    RT_HIP_TRY(ctx, hipGetLastError());
    if (job.samples_per_pixel > 1) {
        const unsigned g = static_cast<unsigned>((job.n_pixels + 255) / 256);
        hipLaunchKernelGGL(sum_samples_kernel, dim3(g), dim3(256), 0, stream, job.samples, job.out, job.n_pixels, job.samples_per_pixel, ctx->dev.inv_samples, job.out8, job.out8_gain);
        RT_HIP_TRY(ctx, hipGetLastError());
    }
    if (julia) {
        // the word stays set on the device until launch_status has seen it, so a later launch's copy still carries an
        // earlier launch's abandonment; the two workspaces have a word (and a pinned copy) each
        const int slot = fixed != nullptr ? 1 : 0;
        RT_HIP_TRY(ctx, hipMemcpyAsync(ctx->h_status + slot, ws.queue + 1, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        ctx->status_pending |= 1u << slot;
        ctx->status_flag[slot] = ws.queue + 1;
    }
    RT_HIP_TRY(ctx, hipEventRecord(ctx->ev_end, stream));
    ctx->events_pending = true;
    return RT_OK;
}

// Waits for the last render launch and reports a launch that trace_julia_kernel's watchdog abandoned (a wave waited
// too long on the job queues: the picture is incomplete).  The device flag stays set until it has been seen here.
int launch_status(rt_ctx* ctx) {
    if (ctx->status_pending == 0u) return RT_OK;
    RT_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_end));   // the last launch's: every copy enqueued before it has landed
    const unsigned pending = ctx->status_pending;
    ctx->status_pending = 0u;
    bool abandoned = false;
    for (int slot = 0; slot < 2; slot++) {
        if ((pending & (1u << slot)) == 0u || ctx->h_status[slot] == 0) continue;
        abandoned = true;
        ctx->h_status[slot] = 0;
        RT_HIP_TRY(ctx, hipMemset(ctx->status_flag[slot], 0, sizeof(unsigned long long)));
    }
    if (!abandoned) return RT_OK;
    return fail(ctx, RT_ERR_HIP, "trace_julia_kernel: a wave waited too long on the job queues and the launch was abandoned");
}

unsigned bit_width(unsigned long long v) {
    unsigned n = 0;
    while (v != 0) {
        n++;
        v >>= 1;
    }
    return n;
}

// Looks at what the last ordered photon pass reported (pinned words, written by its stream) once that pass
// has completed, and sizes the record buffer for the next one.  Returns true if that pass ran out of
// records (its surplus deposits were added with float atomics: correct to rounding, not bit for bit).
bool deposit_overflow_seen(rt_ctx* ctx) {
    if (!ctx->dep_check_pending || hipEventQuery(ctx->ev_dep) != hipSuccess) return false;
    ctx->dep_check_pending = false;
    ctx->dep_last_pass_overflowed = ctx->h_dep_words[1] != 0 || ctx->h_dep_words[3] != 0;
    ctx->dep_key_too_narrow = ctx->h_dep_words[3] != 0;
    ctx->dep_misplaced = ctx->h_dep_words[6] != 0;
    if (ctx->h_dep_words[1] == 0) return false;
    const unsigned long long wanted = ctx->h_dep_words[0] + ctx->h_dep_words[0] / 4;
    ctx->dep_capacity = static_cast<uint32_t>(std::min<unsigned long long>((wanted + 65535ull) & ~65535ull, 1ull << 31));
    return true;
}

int deposit_buffers(rt_ctx* ctx, unsigned long long all_roots, unsigned key_bits, unsigned long long n_texels, hipStream_t stream) {
    (void)deposit_overflow_seen(ctx);
    if (ctx->dep_capacity == 0) {
        // C5: 8.0 M roots leave 0.37 M deposits (+ 0.08 M slots that stay unused: a wave reserves records kDepositChunk at a time)
        const unsigned long long guess = std::max<unsigned long long>(1ull << 18, all_roots / 12);
        ctx->dep_capacity = static_cast<uint32_t>(std::min<unsigned long long>((guess + 65535ull) & ~65535ull, 1ull << 31));
    }
    const size_t n = ctx->dep_capacity;
    int rc;
    if ((rc = grow(ctx, ctx->dep_keys, n * sizeof(unsigned long long))) != RT_OK) return rc;
    if ((rc = grow(ctx, ctx->dep_vals, n * sizeof(float4))) != RT_OK) return rc;
    if ((rc = grow(ctx, ctx->dep_sorted_keys, n * sizeof(unsigned long long))) != RT_OK) return rc;
    if ((rc = grow(ctx, ctx->dep_sorted_index, n * sizeof(unsigned))) != RT_OK) return rc;
    ctx->dep_bucket = n_texels < (1ull << 31) && !ctx->dep_force_sort;
    if (ctx->dep_bucket) {
        if ((rc = grow(ctx, ctx->dep_long_runs, n * sizeof(unsigned))) != RT_OK) return rc;     // `order` of photon_order_bucket
        if ((rc = grow(ctx, ctx->dep_seg_first, n_texels * sizeof(unsigned))) != RT_OK) return rc;
        if ((rc = grow(ctx, ctx->dep_runs, n * sizeof(uint4))) != RT_OK) return rc;
        if ((rc = grow(ctx, ctx->dep_seg_vals, n * sizeof(float4))) != RT_OK) return rc;
        const size_t before = ctx->dep_count.bytes;
        if ((rc = grow(ctx, ctx->dep_count, n_texels * sizeof(unsigned))) != RT_OK) return rc;
        // zero once per allocation: every pass leaves the counts at zero again (place_runs_kernel).  The SIZE tells a new
        // allocation, not the address: the allocator may hand the freed block's address out again for the larger one, whose
        // tail then holds whatever was there (found by tools/fuzz_parity.py on a fresh context: counts of 10^9, records
        // scattered far outside their buffer, a memory fault in scatter_records_kernel)
        if (ctx->dep_count.bytes != before) RT_HIP_TRY(ctx, hipMemsetAsync(ctx->dep_count.ptr, 0, ctx->dep_count.bytes, stream));
        return RT_OK;
    }
    if ((rc = grow(ctx, ctx->dep_long_runs, (n / kDepositLongRun + 1) * sizeof(unsigned))) != RT_OK) return rc;
    size_t temp = 0;
    RT_HIP_TRY(ctx, photon_order_temp_bytes(ctx->dep_capacity, key_bits, &temp));
    ctx->dep_temp_bytes = temp;
    return grow(ctx, ctx->dep_temp, temp ? temp : 1);
}

int photon_enqueue(rt_ctx* ctx, hipStream_t stream, uint32_t part = 0, uint32_t n_parts = 1) {
    if (!ctx->has_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_photon_generate before rt_scene_upload");
    if (ctx->upload_in_flight && stream != nullptr) RT_HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->ev_upload, 0));
    if (const int rc = clear_counters(ctx, ctx->d_counters, ctx->collect_stats, stream); rc != RT_OK) return rc;
    RT_HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, stream));
    if (ctx->photon_steps > 0.0f && ctx->n_slots > 0 && !ctx->point_lights.empty() && ctx->n_phi > 0 && ctx->n_theta > 0) {
        const size_t atlas_bytes = static_cast<size_t>(ctx->n_slots) * ctx->dev.photon_size * ctx->dev.photon_size * sizeof(float4);
        RT_HIP_TRY(ctx, hipMemsetAsync(ctx->dev.photon_atlas, 0, atlas_bytes, stream));
        DevPhotonJob job{};
        const float* t = ctx->d_photon_tables;
        job.sin_phi = t;
        job.cos_phi = t + ctx->n_phi;
        job.sin_theta = t + 2 * ctx->n_phi;
        job.cos_theta = t + 2 * ctx->n_phi + ctx->n_theta;
        job.n_phi = ctx->n_phi;
        job.n_theta = ctx->n_theta;
        job.n_point_lights = static_cast<uint32_t>(ctx->point_lights.size());
        job.point_lights = ctx->d_point_lights;
        job.hues = ctx->hues;
        const uint64_t all_roots = static_cast<uint64_t>(job.n_point_lights) * job.n_phi * job.n_theta * job.hues;
        job.root_base = all_roots * part / n_parts;
        job.n_roots = all_roots * (part + 1) / n_parts - job.root_base;
        if (job.n_roots == 0) {   // more parts than roots
            RT_HIP_TRY(ctx, hipEventRecord(ctx->ev_end, stream));
            ctx->events_pending = true;
            return RT_OK;
        }
        const bool julia = ctx->dev.n_julias > 0;
        const bool stats = ctx->collect_stats;
        int blocks = 0;
        int rc;
        const size_t lds = dynamic_lds_bytes(ctx->dev);
        if (julia) rc = stats ? persistent_grid(ctx, photon_kernel<true, true>, job.n_roots, &blocks, lds) : persistent_grid(ctx, photon_kernel<true, false>, job.n_roots, &blocks, lds);
        else rc = stats ? persistent_grid(ctx, photon_kernel<false, true>, job.n_roots, &blocks, lds) : persistent_grid(ctx, photon_kernel<false, false>, job.n_roots, &blocks, lds);
        if (rc != RT_OK) return rc;
        const uint64_t lanes = static_cast<uint64_t>(blocks) * 256;
        rc = grow(ctx, ctx->stack, lanes * (ctx->dev.max_bounces + 1) * kFrameFieldsHost * sizeof(float));
        if (rc != RT_OK) return rc;
        job.stack = static_cast<float*>(ctx->stack.ptr);
        job.n_waves = static_cast<uint32_t>(lanes / 64);
        job.static_blocks = static_blocks_per_wave(job.n_roots, lanes, RT_STATIC_SHARE_PHOTON);
        job.dyn_base = static_cast<uint64_t>(job.n_waves) * job.static_blocks * 64ull;
        job.id_chunk = id_chunk_for(job.n_roots - job.dyn_base, lanes);
        job.queue_head = ctx->d_dep_words + 16;   // a cache line of its own, behind the control words' (not d_queue: its word [1] is the Julia watchdog's sticky flag)
        job.counters = ctx->d_counters;
        RT_HIP_TRY(ctx, hipMemsetAsync(ctx->d_dep_words, 0, 32 * sizeof(unsigned long long), stream));   // every control word of the pass and its queue head at once
        // reference deposit order: record, sort, sum (rt_photon_order.hip) when the key fits 64 bits
        const unsigned long long n_texels = static_cast<unsigned long long>(ctx->n_slots) * ctx->dev.photon_size * ctx->dev.photon_size;
        // a root deposits at most once per leaf of its tree, 2^(max_bounces+1) in theory; when the key has no room for
        // that many the pass still runs ordered and reports the (never yet seen) root that outgrew its field
        const unsigned texel_bits = bit_width(n_texels), root_bits = bit_width(all_roots);
        const unsigned seq_room = texel_bits + root_bits < 64 ? 64 - texel_bits - root_bits : 0;
        const unsigned seq_bits = std::min<unsigned>(ctx->dev.max_bounces + 2, std::min<unsigned>(seq_room, 31));
        const unsigned key_bits = texel_bits + root_bits + seq_bits;
        const bool ordered = ctx->photon_order && seq_bits >= std::min<unsigned>(ctx->dev.max_bounces + 2, 10);
        ctx->dep_last_pass_ordered = ordered;
        ctx->dep_last_pass_overflowed = false;
        ctx->dep_misplaced = false;
        if (ordered) {
            if ((rc = deposit_buffers(ctx, all_roots, key_bits, n_texels, stream)) != RT_OK) return rc;
            job.dep_keys = static_cast<unsigned long long*>(ctx->dep_keys.ptr);
            job.dep_vals = static_cast<float4*>(ctx->dep_vals.ptr);
            job.dep_words = ctx->d_dep_words;
            job.dep_capacity = ctx->dep_capacity;
            job.dep_count = ctx->dep_bucket ? static_cast<uint32_t*>(ctx->dep_count.ptr) : nullptr;
            job.dep_slot = static_cast<uint32_t*>(ctx->dep_sorted_index.ptr);     // (the sort's index array, unused on this path)
            job.dep_seq_bits = seq_bits;
            job.dep_root_bits = root_bits;
            RT_HIP_TRY(ctx, hipMemsetAsync(ctx->dep_keys.ptr, 0xFF, static_cast<size_t>(ctx->dep_capacity) * sizeof(unsigned long long), stream));
        }
#ifdef RT_GUARD
        job.guard_words = ctx->d_dep_words + 8;       // the first violation: site, index, limit
        job.guard_texels = n_texels;
        job.guard_dirs = static_cast<uint64_t>(job.n_point_lights) * job.n_phi * job.n_theta;
        job.guard_stack_frames = ctx->dev.max_bounces + 1;
#endif
        // roots that can leave nothing are dropped before the pass (DevScene::photon_cull_*; not for a slice of the pass,
        // and not under counters: that launch traces every root and checks the claim)
        job.static_sixteenths = RT_STATIC_SHARE_PHOTON;
        const uint64_t n_dirs = static_cast<uint64_t>(job.n_point_lights) * job.n_phi * job.n_theta;
        if (ctx->cull_photon_roots && !stats && n_parts == 1 && ctx->dev.photon_cull_valid != 0u && all_roots < (1ull << 32)) {
            if (ctx->live_dirs.bytes < n_dirs * sizeof(uint32_t)) RT_HIP_TRY(ctx, hipDeviceSynchronize());
            if ((rc = grow(ctx, ctx->live_dirs, n_dirs * sizeof(uint32_t))) != RT_OK) return rc;
            hipLaunchKernelGGL(photon_cull_kernel, dim3(static_cast<unsigned>((n_dirs + 255) / 256)), dim3(256), 0, stream, ctx->dev, job,
                               static_cast<unsigned>(n_dirs), static_cast<uint32_t*>(ctx->live_dirs.ptr), ctx->d_dep_words + 4);
            if ((rc = step_check(ctx, stream, "photon_cull_kernel")) != RT_OK) return rc;
            job.live_dirs = static_cast<const uint32_t*>(ctx->live_dirs.ptr);
            job.live_count = ctx->d_dep_words + 4;
        }
        if (julia) {
            if (stats) hipLaunchKernelGGL((photon_kernel<true, true>), dim3(blocks), dim3(256), lds, stream, ctx->dev, job);
            else hipLaunchKernelGGL((photon_kernel<true, false>), dim3(blocks), dim3(256), lds, stream, ctx->dev, job);
        } else {
            if (stats) hipLaunchKernelGGL((photon_kernel<false, true>), dim3(blocks), dim3(256), lds, stream, ctx->dev, job);
            else hipLaunchKernelGGL((photon_kernel<false, false>), dim3(blocks), dim3(256), lds, stream, ctx->dev, job);
        }
        if ((rc = step_check(ctx, stream, "photon_kernel")) != RT_OK) return rc;
        if (ordered) {
            if (ctx->dep_bucket)
                RT_HIP_TRY(ctx, photon_order_bucket(job.dep_keys, job.dep_vals, job.dep_slot, ctx->dep_capacity, root_bits + seq_bits, static_cast<unsigned>(n_texels), ctx->dev.photon_atlas,
                                                    static_cast<unsigned*>(ctx->dep_count.ptr), static_cast<unsigned*>(ctx->dep_seg_first.ptr), static_cast<uint4*>(ctx->dep_runs.ptr),
                                                    static_cast<unsigned long long*>(ctx->dep_sorted_keys.ptr), static_cast<float4*>(ctx->dep_seg_vals.ptr),
                                                    static_cast<unsigned*>(ctx->dep_long_runs.ptr), ctx->d_dep_words + 5, stream));
            else
                RT_HIP_TRY(ctx, photon_order_reduce(job.dep_keys, job.dep_vals, ctx->dep_capacity, key_bits, root_bits + seq_bits, n_texels, ctx->dev.photon_atlas,
                                                    static_cast<unsigned long long*>(ctx->dep_sorted_keys.ptr), static_cast<unsigned*>(ctx->dep_sorted_index.ptr),
                                                    static_cast<unsigned*>(ctx->dep_long_runs.ptr), ctx->d_dep_words, ctx->dep_temp.ptr, ctx->dep_temp_bytes, stream));
            if ((rc = step_check(ctx, stream, "the ordered sum of the photon deposits")) != RT_OK) return rc;
            RT_HIP_TRY(ctx, hipMemcpyAsync(ctx->h_dep_words, ctx->d_dep_words, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
            RT_HIP_TRY(ctx, hipEventRecord(ctx->ev_dep, stream));
            ctx->dep_check_pending = true;
        }
    }
    RT_HIP_TRY(ctx, hipEventRecord(ctx->ev_end, stream));
    ctx->events_pending = true;
    return RT_OK;
}

int read_stats(rt_ctx* ctx, rt_stats* out) {
    const int status = launch_status(ctx);   // before the early return: a caller without stats must hear of an abandoned launch too
    if (status != RT_OK) return status;
    if (out == nullptr) return RT_OK;
    std::memset(out, 0, sizeof *out);
    unsigned long long c[C_COUNT];
    RT_HIP_TRY(ctx, hipMemcpy(c, ctx->d_counters, sizeof c, hipMemcpyDeviceToHost));
    out->primary_rays = c[C_PRIMARY];
    out->secondary_rays = c[C_SECONDARY];
    out->shadow_rays = c[C_SHADOW];
    out->photon_segments = c[C_PHOTON_SEGMENTS];
    out->photon_deposits = c[C_PHOTON_DEPOSITS];
    out->sphere_tests = c[C_SPHERE_TESTS];
    out->triangle_tests = c[C_TRIANGLE_TESTS];
    out->julia_queries = c[C_JULIA_QUERIES];
    out->julia_steps = c[C_JULIA_STEPS];
    out->julia_de_iterations = c[C_JULIA_DE_ITERATIONS];
    out->julia_normals = c[C_JULIA_NORMALS];
    out->shaded_hits = c[C_SHADED_HITS];
    out->phong_lights = c[C_PHONG_LIGHTS];
    out->fresnel_evals = c[C_FRESNEL];
    out->texture_lookups = c[C_TEXTURE_LOOKUPS];
    out->bvh_node_visits = c[C_BVH_NODE_VISITS];
    out->shadow_rays_elided = c[C_SHADOW_ELIDED];
    out->photon_roots_culled = c[C_PHOTON_ROOTS_CULLED];
    if (ctx->events_pending) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end) == hipSuccess) out->kernel_ms = ms;
    }
    return RT_OK;
}

// Device -> caller's host memory through the context's pinned block (a pageable hipMemcpy stages through the
// runtime's own small bounce buffers at a fraction of the link rate); very large pictures go directly.
int download(rt_ctx* ctx, void* host_out, const void* d_src, size_t bytes) {
    constexpr size_t kPinnedCap = size_t(1) << 30;
    if (bytes > kPinnedCap) {
        RT_HIP_TRY(ctx, hipMemcpy(host_out, d_src, bytes, hipMemcpyDeviceToHost));
        return RT_OK;
    }
    if (ctx->h_out_bytes < bytes) {
        if (ctx->h_out) RT_HIP_TRY(ctx, hipHostFree(ctx->h_out));
        ctx->h_out = nullptr;
        ctx->h_out_bytes = 0;
        RT_HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->h_out), bytes));
        ctx->h_out_bytes = bytes;
    }
    RT_HIP_TRY(ctx, hipMemcpyAsync(ctx->h_out, d_src, bytes, hipMemcpyDeviceToHost, nullptr));
    RT_HIP_TRY(ctx, hipStreamSynchronize(nullptr));
    std::memcpy(host_out, ctx->h_out, bytes);
    return RT_OK;
}

// Color::scaleOut on the device, then 4 bytes per pixel over the link instead of 16.
int download_rgba8(rt_ctx* ctx, const float4* d_rgba, uint64_t n_pixels, float gain, uint8_t* out) {
    int rc = grow(ctx, ctx->out8, n_pixels * 4);
    if (rc != RT_OK) return rc;
    const unsigned g = static_cast<unsigned>((n_pixels + 255) / 256);
    hipLaunchKernelGGL(quantize_kernel, dim3(g), dim3(256), 0, nullptr, d_rgba, static_cast<uchar4*>(ctx->out8.ptr), gain, n_pixels);
    RT_HIP_TRY(ctx, hipGetLastError());
    return download(ctx, out, ctx->out8.ptr, n_pixels * 4);
}

// The photon scan angles (src/photonmap.cpp:38-45) are accumulated in float by the reference's loops; the same loops
// produce them here, and the tables ride in the upload's staging block together with the point-light list.
int stage_photon_tables(rt_ctx* ctx, const rt_scene_desc* d, bool dispersion) {
    ctx->photon_steps = d->photon_steps;
    ctx->hues = dispersion ? 8u : 1u;
    ctx->n_phi = ctx->n_theta = 0;
    ctx->d_photon_tables = nullptr;
    ctx->d_point_lights = nullptr;
    if (!(d->photon_steps > 0.0f)) return RT_OK;
    const float step = 2 * 3.14159265358979323846f / d->photon_steps;
    std::vector<float> sp, cp, st, ct;
    if (!(step > 0.0f)) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "photon_steps gives a non-positive scan step");
    for (float phi = 0.0f; phi < 2 * 3.14159265358979323846f; phi += step) {
        sp.push_back(sinf(phi));
        cp.push_back(cosf(phi));
        if (sp.size() > (1u << 24)) return fail(ctx, RT_ERR_LIMIT, "photon_steps too large");
    }
    for (float theta = 0.0f; theta < 3.14159265358979323846f; theta += step) {
        st.push_back(sinf(theta));
        ct.push_back(cosf(theta));
    }
    ctx->n_phi = static_cast<uint32_t>(sp.size());
    ctx->n_theta = static_cast<uint32_t>(st.size());
    std::vector<float> all;
    all.insert(all.end(), sp.begin(), sp.end());
    all.insert(all.end(), cp.begin(), cp.end());
    all.insert(all.end(), st.begin(), st.end());
    all.insert(all.end(), ct.begin(), ct.end());
    int rc;
    if ((rc = upload(ctx, all, &ctx->d_photon_tables)) != RT_OK) return rc;
    return upload(ctx, ctx->point_lights, &ctx->d_point_lights);
}

}  // namespace

extern "C" {

int rt_hue_table(int preset, float out_rgb[8][3]) {
    static const float int_abs[8][3] = {{1, 0, 0}, {1, 1, 0}, {1, 1, 0}, {0, 1, 1}, {0, 1, 1}, {0, 1, 1}, {1, 0, 1}, {1, 0, 1}};
    static const float float_abs[8][3] = {{1, 0, 0}, {1, 0.75f, 0}, {0.5f, 1, 0}, {0, 1, 0.25f}, {0, 1, 1}, {0, 0.25f, 1}, {0.5f, 0, 1}, {1, 0, 0.75f}};
    if (out_rgb == nullptr || (preset != RT_HUE_INT_ABS && preset != RT_HUE_FLOAT_ABS)) return RT_ERR_INVALID_ARGUMENT;
    std::memcpy(out_rgb, preset == RT_HUE_INT_ABS ? int_abs : float_abs, sizeof int_abs);
    return RT_OK;
}

uint64_t rt_bands_pixel_count(uint32_t width, uint32_t height, const rt_bands* bands) {
    if (bands == nullptr) return static_cast<uint64_t>(width) * height;
    if (bands->band_rows == 0 || bands->band_stride == 0) return 0;
    uint64_t rows = 0;
    for (uint64_t band = bands->first_band;; band += bands->band_stride) {
        const uint64_t y0 = band * bands->band_rows;
        if (y0 >= height) break;
        const uint64_t y1 = y0 + bands->band_rows;
        rows += (y1 > height ? height : y1) - y0;
    }
    return rows * width;
}

const char* rt_last_error(const rt_ctx* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int rt_create(rt_ctx** out_ctx, int device_id) {
    if (out_ctx == nullptr) return fail(nullptr, RT_ERR_INVALID_ARGUMENT, "rt_create: out_ctx is NULL");
    *out_ctx = nullptr;
    // One rt_create at a time: the first one makes the process's first HIP calls, and the runtime's start-up is not safe
    // beside threads that map or unmap memory (include/rt_hip.h) — at least this library's own threads stay out of it.
    static std::mutex create_mutex;
    const std::lock_guard<std::mutex> one_at_a_time(create_mutex);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(nullptr, RT_ERR_NO_DEVICE, "no HIP device is visible (this library has no CPU fallback)");
    if (device_id < 0 || device_id >= count) return fail(nullptr, RT_ERR_INVALID_ARGUMENT, "device %d out of range (0..%d)", device_id, count - 1);
    rt_ctx* ctx = new rt_ctx;
    ctx->device = device_id;
    {
        const char* e = std::getenv("RT_POISON_ALLOC");
        ctx->poison_allocations = e != nullptr && e[0] != '0' && e[0] != 0;
    }
    hipDeviceProp_t prop;
    if (hipSetDevice(device_id) != hipSuccess || hipGetDeviceProperties(&prop, device_id) != hipSuccess) {
        delete ctx;
        return fail(nullptr, RT_ERR_HIP, "cannot select HIP device %d", device_id);
    }
    ctx->n_cus = prop.multiProcessorCount;
    if (hipMalloc(reinterpret_cast<void**>(&ctx->d_queue), 256) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&ctx->d_counters), kCounterBytes) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&ctx->d_dep_words), 32 * sizeof(unsigned long long)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&ctx->h_dep_words), 8 * sizeof(unsigned long long)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&ctx->h_status), 2 * sizeof(unsigned long long)) != hipSuccess ||
        hipEventCreate(&ctx->ev_dep) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_upload, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&ctx->ev_begin) != hipSuccess || hipEventCreate(&ctx->ev_end) != hipSuccess) {
        delete ctx;
        return fail(nullptr, RT_ERR_HIP, "cannot allocate context resources on device %d", device_id);
    }
    {
        // this device's copy of `scratch` (every GPU that loads the code object holds its own, at its own address)
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || p == nullptr) {
            const int rc = fail(nullptr, RT_ERR_HIP, "hipGetSymbolAddress(scratch) on device %d: %s", device_id, hipGetErrorString(e));
            delete ctx;
            return rc;
        }
        ctx->noalloc_scratch = static_cast<unsigned char*>(p);
        if (const char* e = std::getenv("RT_STATIC_SHARE")) ctx->static_share = std::min(16, std::atoi(e));
        if (const char* e = std::getenv("RT_JULIA_HOT_STEPS")) ctx->julia_hot_steps = static_cast<uint32_t>(std::strtoul(e, nullptr, 10));
        if (const char* e = std::getenv("RT_PHOTON_ORDER_SORT")) ctx->dep_force_sort = std::atoi(e) != 0;
        if (const char* e = std::getenv("RT_DENSE_BVH")) ctx->dense_with_bvh = std::atoi(e) != 0;
        if (const char* e = std::getenv("RT_JULIA_GROUP_TILES")) ctx->julia_group_tiles = static_cast<uint32_t>(std::strtoul(e, nullptr, 10));
    }
    ctx->counters_clean = hipMemset(ctx->d_counters, 0, kCounterBytes) == hipSuccess;
    (void)hipMemset(ctx->d_queue, 0, 256);   // [0] queue head, [1] watchdog flag of trace_julia_kernel
    ctx->h_status[0] = ctx->h_status[1] = 0;
    *out_ctx = ctx;
    return RT_OK;
}

void rt_destroy(rt_ctx* ctx) {
    if (ctx == nullptr) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    free_scene(ctx);
    if (ctx->h_arena) (void)hipHostFree(ctx->h_arena);
    if (ctx->h_out) (void)hipHostFree(ctx->h_out);
    if (ctx->ev_upload) (void)hipEventDestroy(ctx->ev_upload);
    for (Buffer* b : {&ctx->scene_arena, &ctx->bvh_nodes, &ctx->atlas, &ctx->out8, &ctx->acc, &ctx->samples, &ctx->stack, &ctx->out_staging, &ctx->kat_a, &ctx->kat_b, &ctx->kat_c, &ctx->kat_d, &ctx->sched, &ctx->dep_count, &ctx->dep_seg_first, &ctx->dep_runs, &ctx->dep_seg_vals,
                      &ctx->dep_keys, &ctx->dep_vals, &ctx->dep_sorted_keys, &ctx->dep_sorted_index, &ctx->dep_long_runs, &ctx->dep_temp, &ctx->live_dirs}) {
        if (b->ptr) (void)hipFree(b->ptr);
    }
    if (ctx->d_dep_words) (void)hipFree(ctx->d_dep_words);
    if (ctx->h_dep_words) (void)hipHostFree(ctx->h_dep_words);
    if (ctx->h_status) (void)hipHostFree(ctx->h_status);
    if (ctx->ev_dep) (void)hipEventDestroy(ctx->ev_dep);
    if (ctx->d_queue) (void)hipFree(ctx->d_queue);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
    if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
    delete ctx;
}

int rt_set_option(rt_ctx* ctx, int option, int64_t value) {
    if (ctx == nullptr) return RT_ERR_INVALID_ARGUMENT;
    switch (option) {
        case RT_OPT_COLLECT_STATS: ctx->collect_stats = value != 0; return RT_OK;
        case RT_OPT_BVH_MIN_TRIANGLES: ctx->bvh_min_triangles = value; return RT_OK;
        case RT_OPT_DOF_SEED: ctx->dof_seed = static_cast<uint32_t>(value); return RT_OK;
        case RT_OPT_PHOTON_ORDER: ctx->photon_order = value != 0; return RT_OK;
        case RT_OPT_PHOTON_RECORDS:
            if (value < 0 || value > (1ll << 31)) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "RT_OPT_PHOTON_RECORDS out of range");
            ctx->dep_capacity = static_cast<uint32_t>((value + 63) & ~63ll);
            return RT_OK;
        case RT_OPT_DEBUG_FORCE_ABORT: ctx->debug_force_abort = value != 0; return RT_OK;
        case RT_OPT_ELIDE_UNLIT_SHADOWS: ctx->elide_unlit_shadows = value != 0; return RT_OK;
        case RT_OPT_CULL_PHOTON_ROOTS: ctx->cull_photon_roots = value != 0; return RT_OK;
        case RT_OPT_SHARED_GPU: ctx->shared_gpu = value != 0; return RT_OK;
        default: return fail(ctx, RT_ERR_INVALID_ARGUMENT, "unknown option %d", option);
    }
}

int rt_scene_upload(rt_ctx* ctx, const rt_scene_desc* d) {
    if (ctx == nullptr) return RT_ERR_INVALID_ARGUMENT;
    if (d == nullptr) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_scene_upload: desc is NULL");
    if (d->abi_version != RT_ABI_VERSION) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_scene_desc.abi_version %u != %u", d->abi_version, RT_ABI_VERSION);
    const rt_camera& cam = d->camera;
    if (cam.width == 0 || cam.height == 0 || cam.supersampling == 0) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "camera resolution and supersampling must be > 0");
    if (cam.max_bounces > 64) return fail(ctx, RT_ERR_LIMIT, "max_bounces %u > 64", cam.max_bounces);
    if (cam.supersampling > 64) return fail(ctx, RT_ERR_LIMIT, "supersampling %u > 64", cam.supersampling);
    // validate every index before anything reaches a kernel
    for (uint32_t i = 0; i < d->n_objects; i++) {
        const rt_object& o = d->objects[i];
        const uint32_t limit = o.kind == RT_OBJ_SPHERE ? d->n_spheres : o.kind == RT_OBJ_TRIANGLE ? d->n_triangles : o.kind == RT_OBJ_JULIA ? d->n_julias : 0;
        if (o.index >= limit) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "object %u: kind %d index %u out of range", i, o.kind, o.index);
        if (o.material >= d->n_materials) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "object %u: material %u out of range", i, o.material);
    }
    for (uint32_t i = 0; i < d->n_materials; i++) {
        if (d->materials[i].texture >= static_cast<int32_t>(d->n_textures)) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "material %u: texture %d out of range", i, d->materials[i].texture);
    }
    for (uint32_t i = 0; i < d->n_textures; i++) {
        const rt_texture& t = d->textures[i];
        if (t.width == 0 || t.height == 0 || t.offset + static_cast<uint64_t>(t.width) * t.height * 4 > d->n_texel_bytes) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "texture %u lies outside the texel pool", i);
    }
    if (d->photon_steps > 0.0f && d->photon_texture_size == 0) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "photon_texture_size must be > 0 when photon_steps > 0");

    RT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    // the tables are overwritten in place by a copy on the default stream: it must run after the work enqueued on
    // this context so far (an event wait on the device; the host does not block)
    if (ctx->events_pending) RT_HIP_TRY(ctx, hipStreamWaitEvent(nullptr, ctx->ev_end, 0));
    free_scene(ctx);
    DevScene& sc = ctx->dev;
    std::memset(&sc, 0, sizeof sc);

    // ---- camera frame (src/raytracer.cpp:17-29), evaluated in binary32 on the host ------------------
    std::memcpy(sc.cam, cam.camera_to_world.m, sizeof sc.cam);
    const float wf = static_cast<float>(cam.width), hf = static_cast<float>(cam.height);
    const float half_x = -tanf(cam.fov), half_y = tanf(cam.fov) * (hf / wf);
    sc.pixel_size[0] = (-2.0f / wf) * half_x;
    sc.pixel_size[1] = (-2.0f / hf) * half_y;
    std::vector<float> ray_x(cam.width), ray_y(cam.height), sub_disp(cam.supersampling);
    {
        float rx = half_x + 0.5f * sc.pixel_size[0];
        for (uint32_t x = 0; x < cam.width; x++) {
            ray_x[x] = rx;
            rx += sc.pixel_size[0];
        }
        for (uint32_t y = 0; y < cam.height; y++) ray_y[y] = half_y + y * sc.pixel_size[1] + 0.5f * sc.pixel_size[1];
        for (uint32_t i = 0; i < cam.supersampling; i++) sub_disp[i] = 2.0f * (i + 1) / (cam.supersampling + 1) - 1.0f;
    }
    int rc;
    if ((rc = upload(ctx, ray_x, &sc.ray_x)) != RT_OK) return rc;
    if ((rc = upload(ctx, ray_y, &sc.ray_y)) != RT_OK) return rc;
    if ((rc = upload(ctx, sub_disp, &sc.sub_disp)) != RT_OK) return rc;
    sc.focus_distance = cam.focus_distance;
    sc.lens_size = cam.lens_size;
    sc.width = cam.width;
    sc.height = cam.height;
    sc.supersampling = cam.supersampling;
    sc.max_bounces = cam.max_bounces;
    sc.inv_samples = 1.0f / (cam.supersampling * cam.supersampling);
    sc.dispersion = d->dispersion_mode ? 1 : 0;
    for (int h = 0; h < 8; h++) {
        for (int c = 0; c < 3; c++) sc.hue[h][c] = d->hue_rgb[h][c];
        sc.hue[h][3] = 1.0f;
    }
    std::memcpy(sc.background, d->background, sizeof sc.background);
    std::memcpy(sc.ambient, d->ambient, sizeof sc.ambient);

    // ---- lights ------------------------------------------------------------------------------------
    std::vector<DevLight> lights(d->n_lights);
    ctx->point_lights.clear();
    for (uint32_t i = 0; i < d->n_lights; i++) {
        std::memcpy(lights[i].position, d->lights[i].position, sizeof lights[i].position);
        lights[i].type = d->lights[i].type == RT_LIGHT_PARALLEL ? 0 : 1;
        std::memcpy(lights[i].power, d->lights[i].power, sizeof lights[i].power);
        if (lights[i].type == 1) ctx->point_lights.push_back(i);
    }
    sc.n_lights = d->n_lights;
    // DevMaterial::phong_tame / resolve_lane: a light is benign when nothing about it can turn a product with zero into a NaN
    auto tame = [](float v) { return std::fabs(v) <= 1e18f; };     // (false for NaN)
    sc.lights_benign = 1u;
    for (const DevLight& l : lights) {
        bool ok = tame(l.power[0]) && tame(l.power[1]) && tame(l.power[2]) && tame(l.position[0]) && tame(l.position[1]) && tame(l.position[2]);
        if (l.type == 0) ok = ok && (l.position[0] != 0.0f || l.position[1] != 0.0f || l.position[2] != 0.0f) &&
                              std::fabs(l.position[0]) + std::fabs(l.position[1]) + std::fabs(l.position[2]) >= 1e-18f;
        if (!ok) sc.lights_benign = 0u;
    }

    // ---- materials, textures -----------------------------------------------------------------------
    std::vector<DevMaterial> mats(d->n_materials);
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const rt_material& m = d->materials[i];
        DevMaterial& o = mats[i];
        std::memcpy(o.color, m.color, sizeof o.color);
        o.texture = m.texture;
        o.ka = m.ka; o.kd = m.kd; o.ks = m.ks; o.exponent = m.exponent;
        o.reflectance = m.reflectance; o.transmittance = m.transmittance;
        o.refr_re = m.refraction_re; o.refr_im = m.refraction_im; o.dispersion = m.dispersion;
        o.refr_norm = m.refraction_re * m.refraction_re + m.refraction_im * m.refraction_im;   // std::norm
        o.phong_tame = (tame(m.kd) && tame(m.ks) && m.exponent >= 0.0f && m.exponent <= 1e6f && tame(m.color[0]) && tame(m.color[1]) && tame(m.color[2])) ? 1u : 0u;
    }
    std::vector<DevTexture> texs(d->n_textures);
    std::vector<uint32_t> texels;     // RGBA8 as given; the kernels form u8 / 255.0f themselves (rt_intersect.h: unorm8)
    for (uint32_t i = 0; i < d->n_textures; i++) {
        const rt_texture& t = d->textures[i];
        texs[i] = {t.width, t.height, static_cast<uint32_t>(texels.size()), 0u};
        const uint8_t* src = d->texels + t.offset;
        const size_t n = static_cast<size_t>(t.width) * t.height;
        const size_t at = texels.size();
        texels.resize(at + n);
        for (size_t k = 0; k < n; k++) {
            texels[at + k] = static_cast<uint32_t>(src[k * 4 + 0]) | (static_cast<uint32_t>(src[k * 4 + 1]) << 8) |
                             (static_cast<uint32_t>(src[k * 4 + 2]) << 16) | (static_cast<uint32_t>(src[k * 4 + 3]) << 24);
        }
    }
    if (texels.size() > 0xffffffffull) return fail(ctx, RT_ERR_LIMIT, "texture pool too large");
    if ((rc = upload(ctx, texels, &sc.texels)) != RT_OK) return rc;

    // ---- geometry ----------------------------------------------------------------------------------
    std::vector<DevSphere> spheres(d->n_spheres);
    for (uint32_t i = 0; i < d->n_spheres; i++) {
        const rt_sphere& s = d->spheres[i];
        std::memcpy(spheres[i].center, s.center, sizeof s.center);
        spheres[i].radius = s.radius;
        std::memcpy(spheres[i].w2o, s.world_to_object.m, sizeof spheres[i].w2o);
        std::memcpy(spheres[i].o2w, s.object_to_world.m, sizeof spheres[i].o2w);
        std::memcpy(spheres[i].o2wn, s.object_to_world_normals.m, sizeof spheres[i].o2wn);
    }
    std::vector<DevTriTest> tt(d->n_triangles);
    std::vector<DevTriShade> ts(d->n_triangles);
    for (uint32_t i = 0; i < d->n_triangles; i++) {
        const rt_triangle& t = d->triangles[i];
        for (int c = 0; c < 3; c++) {
            tt[i].p0[c] = t.position[0][c];
            tt[i].e1[c] = t.position[1][c] - t.position[0][c];   // edge1, src/objects.cpp:56
            tt[i].e2[c] = t.position[2][c] - t.position[0][c];   // edge2, src/objects.cpp:57
            tt[i].pad[c] = 0.0f;
            ts[i].n0[c] = t.normal[0][c];
            ts[i].n1[c] = t.normal[1][c];
            ts[i].n2[c] = t.normal[2][c];
        }
        for (int c = 0; c < 2; c++) {
            ts[i].uv0[c] = t.uv[0][c];
            ts[i].uv1[c] = t.uv[1][c];
            ts[i].uv2[c] = t.uv[2][c];
        }
        ts[i].pad = 0.0f;
    }
    // per-object tables, runs (Julia objects are kept out of the uniform loop) and photon slots
    std::vector<uint32_t> obj_material(d->n_objects);
    ctx->photon_slot.assign(d->n_objects, -1);
    ctx->object_kind.assign(d->n_objects, 0);
    ctx->object_prim.assign(d->n_objects, 0);
    std::vector<DevRun> runs;
    std::vector<DevJulia> julias;
    ctx->n_slots = 0;
    // geometry tables are re-ordered into scene order so that a run is contiguous
    std::vector<DevSphere> spheres_ord;
    std::vector<DevTriTest> tt_ord;
    std::vector<DevTriShade> ts_ord;
    for (uint32_t i = 0; i < d->n_objects; i++) {
        const rt_object& o = d->objects[i];
        obj_material[i] = o.material;
        ctx->object_kind[i] = static_cast<uint32_t>(o.kind);
        const DevMaterial& m = mats[o.material];
        if (d->photon_steps > 0.0f && m.refr_norm <= 0.0f) ctx->photon_slot[i] = static_cast<int32_t>(ctx->n_slots++);
        if (o.kind == RT_OBJ_JULIA) {
            const rt_julia& j = d->julias[o.index];
            DevJulia dj{};
            std::memcpy(dj.position, j.position, sizeof dj.position);
            dj.scale = j.scale;
            std::memcpy(dj.c, j.c, sizeof dj.c);
            dj.cut_plane = j.cut_plane;
            dj.inv_scale = 1.0f / j.scale;
            dj.object = i;
            dj.material = o.material;
            std::memcpy(dj.w2o, j.world_to_object.m, sizeof dj.w2o);
            std::memcpy(dj.o2w, j.object_to_world.m, sizeof dj.o2w);
            std::memcpy(dj.o2wn, j.object_to_world_normals.m, sizeof dj.o2wn);
            ctx->object_prim[i] = static_cast<uint32_t>(julias.size());
            julias.push_back(dj);
            continue;
        }
        const int kind = o.kind == RT_OBJ_SPHERE ? RUN_SPHERES : RUN_TRIANGLES;
        uint32_t prim;
        if (kind == RUN_SPHERES) {
            prim = static_cast<uint32_t>(spheres_ord.size());
            spheres_ord.push_back(spheres[o.index]);
        } else {
            prim = static_cast<uint32_t>(tt_ord.size());
            tt_ord.push_back(tt[o.index]);
            ts_ord.push_back(ts[o.index]);
        }
        ctx->object_prim[i] = prim;
        if (!runs.empty() && runs.back().kind == kind && runs.back().first_object + runs.back().count == i) {
            runs.back().count++;
        } else {
            DevRun r{};
            r.kind = kind;
            r.first_object = i;
            r.first_prim = prim;
            r.count = 1;
            runs.push_back(r);
        }
    }
    sc.n_objects = d->n_objects;
    sc.n_runs = static_cast<uint32_t>(runs.size());
    sc.n_julias = static_cast<uint32_t>(julias.size());
    // chunk bounds of the triangle runs (same conservative growth as the BVH leaf boxes in rt_bvh.hip;
    // a chunk with a degenerate triangle is unbounded: src/objects.cpp:66-78 divides by zero there and
    // accepts NaN hits for rays that come nowhere near it)
    std::vector<DevTriChunk> chunks;
    for (DevRun& r : runs) {
        if (r.kind != RUN_TRIANGLES) continue;
        r.first_chunk = static_cast<uint32_t>(chunks.size());
        for (uint32_t i = 0; i < r.count; i += kTriChunk) {
            DevTriChunk c{};
            bool bounded = true;
            for (int a = 0; a < 3; a++) { c.lo[a] = INFINITY; c.hi[a] = -INFINITY; }
            for (uint32_t k = i; k < r.count && k < i + kTriChunk; k++) {
                const DevTriTest& t = tt_ord[r.first_prim + k];
                const double e1[3] = {t.e1[0], t.e1[1], t.e1[2]}, e2[3] = {t.e2[0], t.e2[1], t.e2[2]};
                const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
                const double area2 = cx * cx + cy * cy + cz * cz;
                const double l1 = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2], l2 = e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2];
                const double p0 = std::fabs(double(t.p0[0])) + std::fabs(double(t.p0[1])) + std::fabs(double(t.p0[2]));
                if (!(area2 > 1e-12 * l1 * l2) || !std::isfinite(area2) || !std::isfinite(p0)) bounded = false;
                for (int a = 0; a < 3; a++) {
                    const float q0 = t.p0[a], q1 = t.p0[a] + t.e1[a], q2 = t.p0[a] + t.e2[a];
                    const float lo = std::fmin(q0, std::fmin(q1, q2)), hi = std::fmax(q0, std::fmax(q1, q2));
                    const float grow = 2.0f * 0.0001f * (std::fabs(t.e1[a]) + std::fabs(t.e2[a])) + 4e-6f * std::fmax(std::fabs(lo), std::fabs(hi)) + 1e-30f;
                    c.lo[a] = std::fmin(c.lo[a], lo - grow);
                    c.hi[a] = std::fmax(c.hi[a], hi + grow);
                }
            }
            // distance slack of the cull (rt_kernels.hip: query_bvh explains): 1e-4 of the box's size here, 1e-4 of
            // the distance in the kernel
            c.slack = 1e-4f * ((c.hi[0] - c.lo[0]) + (c.hi[1] - c.lo[1]) + (c.hi[2] - c.lo[2]));
            if (!bounded) {
                for (int a = 0; a < 3; a++) { c.lo[a] = -INFINITY; c.hi[a] = INFINITY; }
                c.slack = 0.0f;
            }
            chunks.push_back(c);
        }
    }
    if ((rc = upload(ctx, chunks, &sc.tri_chunks)) != RT_OK) return rc;
    if ((rc = upload(ctx, runs, &sc.runs)) != RT_OK) return rc;
    if ((rc = upload(ctx, tt_ord, &sc.tri_test)) != RT_OK) return rc;
    if ((rc = upload(ctx, spheres_ord, &sc.spheres)) != RT_OK) return rc;
    {
        // Bounding spheres for the wave-level sphere culling.  A ray is skipped only when, in exact
        // arithmetic, it misses the bounding sphere by more than every rounding error Sphere::intersect
        // (src/objects.cpp:12-44) can make: that code works in object space with binary32 operations on
        // magnitudes up to |W||o| + |t| + |centre| + r, so its discriminant, roots and object-space
        // maximum are off by at most a few ulps of those magnitudes squared; mapped back to world space
        // the error grows by the condition number of the transform.  slack = 256 * 2^-24 * cond^2 *
        // (|o|^2 + magnitude2) covers it with two orders of magnitude to spare; transforms that are not
        // finite or are singular get an unbounded sphere (never skipped).
        std::vector<DevSphereBound> sb(spheres_ord.size());
        for (size_t i = 0; i < spheres_ord.size(); i++) {
            const DevSphere& s = spheres_ord[i];
            DevSphereBound& b = sb[i];
            double a[3][3], cw[3];
            bool ok = true;
            for (int r = 0; r < 3; r++) {
                cw[r] = s.o2w[r * 4 + 3];
                for (int c = 0; c < 3; c++) {
                    a[r][c] = s.o2w[r * 4 + c];
                    cw[r] += a[r][c] * s.center[c];
                    ok = ok && std::isfinite(a[r][c]) && std::isfinite(double(s.w2o[r * 4 + c]));
                }
                ok = ok && std::isfinite(cw[r]) && std::isfinite(double(s.w2o[r * 4 + 3]));
            }
            // spectral norms of o2w and w2o (3x3 parts), bounded by the Frobenius norm
            double fo = 0.0, fw = 0.0, tw = 0.0;
            for (int r = 0; r < 3; r++) {
                for (int c = 0; c < 3; c++) {
                    fo += a[r][c] * a[r][c];
                    fw += double(s.w2o[r * 4 + c]) * double(s.w2o[r * 4 + c]);
                }
                tw += double(s.w2o[r * 4 + 3]) * double(s.w2o[r * 4 + 3]);
            }
            fo = std::sqrt(fo);
            fw = std::sqrt(fw);
            const double cond = fo * fw;   // >= true condition number (>= 1 for any invertible map, Frobenius >= spectral)
            const double rad = std::fabs(double(s.radius)) * fo * 1.001 + 1e-30;
            const double c2 = double(s.center[0]) * s.center[0] + double(s.center[1]) * s.center[1] + double(s.center[2]) * s.center[2];
            // magnitudes of the object-space operands, mapped to world units with |o2w|
            const double mag2 = cw[0] * cw[0] + cw[1] * cw[1] + cw[2] * cw[2] + rad * rad + fo * fo * (tw + c2 + double(s.radius) * s.radius);
            ok = ok && std::isfinite(cond) && cond >= 1.0 && cond < 1e6 && std::isfinite(mag2) && std::isfinite(rad) && s.radius == s.radius;
            b.center[0] = static_cast<float>(cw[0]);
            b.center[1] = static_cast<float>(cw[1]);
            b.center[2] = static_cast<float>(cw[2]);
            b.radius = ok ? static_cast<float>(rad) : INFINITY;
            b.radius2 = ok ? static_cast<float>(rad * rad * 1.000001) : INFINITY;
            b.slack_scale = ok ? static_cast<float>(256.0 * 5.9604644775390625e-08 * cond * cond) : INFINITY;
            b.magnitude2 = ok ? static_cast<float>(mag2 * 1.000001) : INFINITY;
            b.pad = 0.0f;
        }
        if ((rc = upload(ctx, sb, &sc.sphere_bounds)) != RT_OK) return rc;
        // DevScene::photon_cull_*: the box no photon root can do anything outside of
        bool valid = true, any = false;
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        double oo_max = 0.0;      // |origin|^2 of a root: the lights' positions (the sphere slack grows with it)
        for (const DevLight& l : lights) oo_max = std::fmax(oo_max, double(l.position[0]) * l.position[0] + double(l.position[1]) * l.position[1] + double(l.position[2]) * l.position[2]);
        auto specular = [&](uint32_t object) { return !(mats[obj_material[object]].refr_norm <= 0.0f); };   // as photon_kernel decides it
        for (const DevJulia& j : julias) {
            if (!(mats[j.material].refr_norm <= 0.0f)) valid = false;
        }
        for (const DevRun& r : runs) {
            for (uint32_t k = 0; k < r.count && valid; k++) {
                if (!specular(r.first_object + k)) continue;
                any = true;
                if (r.kind == RUN_TRIANGLES) {
                    const DevTriChunk& c = chunks[r.first_chunk + k / kTriChunk];     // (its chunk's box: a superset)
                    for (int a = 0; a < 3; a++) {
                        lo[a] = std::fmin(lo[a], double(c.lo[a]));
                        hi[a] = std::fmax(hi[a], double(c.hi[a]));
                    }
                } else if (r.kind == RUN_SPHERES) {
                    const DevSphereBound& b = sb[r.first_prim + k];
                    const double reach = std::sqrt(double(b.radius2) + double(b.slack_scale) * (oo_max + double(b.magnitude2))) * 1.00001;
                    for (int a = 0; a < 3; a++) {
                        lo[a] = std::fmin(lo[a], double(b.center[a]) - reach);
                        hi[a] = std::fmax(hi[a], double(b.center[a]) + reach);
                    }
                } else {
                    valid = false;
                }
            }
        }
        for (int a = 0; a < 3; a++) {
            // outward-rounded floats with a relative margin (the root test uses a 1-ulp reciprocal)
            const double pad = 1e-5 * (std::fabs(lo[a]) + std::fabs(hi[a])) + 1e-30;
            sc.photon_cull_lo[a] = std::nextafter(static_cast<float>(lo[a] - pad), -INFINITY);
            sc.photon_cull_hi[a] = std::nextafter(static_cast<float>(hi[a] + pad), INFINITY);
            valid = valid && std::isfinite(sc.photon_cull_lo[a]) && std::isfinite(sc.photon_cull_hi[a]);
        }
        valid = valid && any && std::isfinite(oo_max);
        sc.photon_cull_valid = valid ? 1u : 0u;
        sc.photon_cull_pad = 0u;
    }
    if ((rc = upload(ctx, julias, &sc.julias)) != RT_OK) return rc;
    {
        // the shade blob: every table the shading code reads per lane, 16-byte aligned pieces
        std::vector<unsigned char> blob;
        auto place = [&blob](const void* data, size_t bytes) -> size_t {
            const size_t off = (blob.size() + 15) & ~size_t(15);
            blob.resize(off + (bytes ? bytes : 16), 0);
            if (bytes) std::memcpy(blob.data() + off, data, bytes);
            return off;
        };
        const size_t o_mat = place(mats.data(), mats.size() * sizeof(DevMaterial));
        const size_t o_light = place(lights.data(), lights.size() * sizeof(DevLight));
        const size_t o_tex = place(texs.data(), texs.size() * sizeof(DevTexture));
        const size_t o_om = place(obj_material.data(), obj_material.size() * sizeof(uint32_t));
        const size_t o_ps = place(ctx->photon_slot.data(), ctx->photon_slot.size() * sizeof(int32_t));
        const size_t o_ts = place(ts_ord.data(), ts_ord.size() * sizeof(DevTriShade));
        blob.resize((blob.size() + 15) & ~size_t(15), 0);
        const unsigned char* d_blob = nullptr;
        if ((rc = upload(ctx, blob, &d_blob)) != RT_OK) return rc;
        // photon scan tables and the point-light list ride in the same block
        if ((rc = stage_photon_tables(ctx, d, sc.dispersion != 0)) != RT_OK) return rc;
        if ((rc = flush_uploads(ctx)) != RT_OK) return rc;     // every pointer staged so far is valid from here on
        sc.shade_blob = d_blob;
        sc.shade_blob_bytes = static_cast<uint32_t>(blob.size());
        sc.shade_blob_in_lds = blob.size() <= kMaxLdsBlobBytes ? 1u : 0u;
        sc.materials = reinterpret_cast<const DevMaterial*>(d_blob + o_mat);
        sc.lights = reinterpret_cast<const DevLight*>(d_blob + o_light);
        sc.textures = reinterpret_cast<const DevTexture*>(d_blob + o_tex);
        sc.object_material = reinterpret_cast<const uint32_t*>(d_blob + o_om);
        sc.object_photon_slot = reinterpret_cast<const int32_t*>(d_blob + o_ps);
        sc.tri_shade = reinterpret_cast<const DevTriShade*>(d_blob + o_ts);
    }
    sc.bvh_nodes = nullptr;
    sc.bvh_prims = nullptr;
    sc.bvh_lds_stack_entries = 0;
    sc.bvh_records_in_lds = 0;
    size_t total_bvh_records = 0;
    {
        // GPU-built BVH for every triangle run that is long enough (RT_OPT_BVH_MIN_TRIANGLES)
        // A run qualifies only if every triangle has a finite, non-zero area: a zero-area triangle makes
        // src/objects.cpp:66-78 divide by zero and accept hits with NaN distances for rays that come
        // nowhere near it, which only the sequential scan reproduces.
        auto qualifies = [&](const DevRun& r) {
            if (r.kind != RUN_TRIANGLES || r.count < 2 || static_cast<int64_t>(r.count) < ctx->bvh_min_triangles) return false;
            for (uint32_t k = 0; k < r.count; k++) {
                const DevTriTest& t = tt_ord[r.first_prim + k];
                const double e1[3] = {t.e1[0], t.e1[1], t.e1[2]}, e2[3] = {t.e2[0], t.e2[1], t.e2[2]};
                const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
                const double area2 = cx * cx + cy * cy + cz * cz;
                const double l1 = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2], l2 = e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2];
                const double p0 = std::fabs(double(t.p0[0])) + std::fabs(double(t.p0[1])) + std::fabs(double(t.p0[2]));
                if (!(area2 > 1e-12 * l1 * l2) || !std::isfinite(area2) || !std::isfinite(p0)) return false;   // sin(angle) <= 1e-6
            }
            return true;
        };
        size_t total_nodes = 0;
        for (const DevRun& r : runs) {
            if (qualifies(r)) total_nodes += size_t(r.count) - 1;
        }
        if (total_nodes > 0) {
            if (ctx->bvh_nodes.bytes < total_nodes * sizeof(DevBvhPair)) RT_HIP_TRY(ctx, hipDeviceSynchronize());
            if ((rc = grow(ctx, ctx->bvh_nodes, total_nodes * sizeof(DevBvhPair))) != RT_OK) return rc;
            void* d_nodes = ctx->bvh_nodes.ptr;
            size_t at = 0;
            for (DevRun& r : runs) {
                if (qualifies(r)) {
                    if (r.count <= kBvhHostBuildMax) {
                        std::vector<DevBvhPair> pairs(size_t(r.count) - 1);
                        build_triangle_bvh_host(tt_ord.data(), r.first_prim, r.count, pairs.data());
                        RT_HIP_TRY(ctx, hipMemcpy(static_cast<DevBvhPair*>(d_nodes) + at, pairs.data(), pairs.size() * sizeof(DevBvhPair), hipMemcpyHostToDevice));
                    } else {
                        RT_HIP_TRY(ctx, build_triangle_bvh(sc.tri_test, r.first_prim, r.count, static_cast<DevBvhPair*>(d_nodes) + at));
                    }
                    r.kind = RUN_TRIANGLE_BVH;
                    r.bvh_root = static_cast<uint32_t>(at);
                    at += size_t(r.count) - 1;
                }
            }
            sc.bvh_nodes = static_cast<const DevBvhPair*>(d_nodes);
            sc.bvh_lds_stack_entries = sc.shade_blob_in_lds ? 8 : kBvhLdsStackEntries;
            total_bvh_records = total_nodes;
            // the runs table was uploaded before the kinds changed: refresh it
            RT_HIP_TRY(ctx, hipMemcpy(const_cast<DevRun*>(sc.runs), runs.data(), runs.size() * sizeof(DevRun), hipMemcpyHostToDevice));
        }
    }

    // ---- photon atlas (the scan tables were staged with the scene tables: stage_photon_tables) -------------------
    sc.photon_size = d->photon_texture_size;
    sc.photon_factor = d->photon_factor;
    sc.photon_atlas = nullptr;
    if (d->photon_steps > 0.0f && ctx->n_slots > 0) {
        const size_t atlas_bytes = static_cast<size_t>(ctx->n_slots) * sc.photon_size * sc.photon_size * sizeof(float4);
        if (ctx->atlas.bytes < atlas_bytes) RT_HIP_TRY(ctx, hipDeviceSynchronize());
        if ((rc = grow(ctx, ctx->atlas, atlas_bytes)) != RT_OK) return rc;
        RT_HIP_TRY(ctx, hipMemsetAsync(ctx->atlas.ptr, 0, atlas_bytes, nullptr));
        sc.photon_atlas = static_cast<float4*>(ctx->atlas.ptr);
    }
    // the BVH records of a small scene (C2: 63 records, 4 KB) go into LDS too, if need be at the price of stack entries
    sc.frame_slots_in_lds = 1;
    if (sc.n_julias == 0 && total_bvh_records > 0 && total_bvh_records <= 128) {
        const uint32_t entries = sc.bvh_lds_stack_entries;
        sc.bvh_records_in_lds = static_cast<uint32_t>(total_bvh_records);
        while (kTraceStaticLdsBytes + dynamic_lds_bytes(sc) > kLdsPerBlockFor3 && sc.bvh_lds_stack_entries > 4) sc.bvh_lds_stack_entries -= 2;
        if (kTraceStaticLdsBytes + dynamic_lds_bytes(sc) > kLdsPerBlockFor3) {
            sc.bvh_records_in_lds = 0;
            sc.bvh_lds_stack_entries = entries;
        }
    }
    // a second frame slot per lane in LDS when three workgroups per CU still fit with it (TopInLds)
    if (sc.n_julias == 0) {
        sc.frame_slots_in_lds = 2;
        if (kTraceStaticLdsBytes + dynamic_lds_bytes(sc) > kLdsPerBlockFor3) sc.frame_slots_in_lds = 1;
    }
    {
        // persistent-grid sizes of the render kernels for this scene (the LDS copy of the shade blob decides), asked
        // once per upload so that a render call makes no runtime query (rt_render_device_noalloc makes none at all)
        const size_t lds_t = dynamic_lds_bytes(sc);
        RT_HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&ctx->trace_blocks_per_cu[0], trace_kernel<false, false>, 256, lds_t));
        RT_HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&ctx->trace_blocks_per_cu[1], trace_kernel<true, false>, 256, lds_t));
        ctx->dense_blocks_per_cu[0] = ctx->dense_blocks_per_cu[1] = 0;
        if (sc.n_julias == 0 && (sc.bvh_nodes == nullptr || ctx->dense_with_bvh)) {
            DevScene one_slot = sc;
            one_slot.frame_slots_in_lds = 1;
            if (sc.bvh_nodes != nullptr) dense_bvh_layout(one_slot);
            const size_t lds_d = dynamic_lds_bytes(one_slot);
            RT_HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&ctx->dense_blocks_per_cu[0], trace_kernel<false, true>, 256, lds_d));
            RT_HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&ctx->dense_blocks_per_cu[1], trace_kernel<true, true>, 256, lds_d));
        }
        if (std::getenv("RT_HIP_DEBUG_LDS") != nullptr)
            std::fprintf(stderr, "rt_scene_upload: shade blob %u bytes (%s), BVH stack entries in LDS %u, BVH records in LDS %u, frame slots %u, dynamic LDS %zu bytes, trace_kernel blocks per CU %d (dense variant: %d)\n",
                         sc.shade_blob_bytes, sc.shade_blob_in_lds ? "in LDS" : "in HBM", sc.bvh_lds_stack_entries, sc.bvh_records_in_lds, sc.frame_slots_in_lds, lds_t, ctx->trace_blocks_per_cu[0], ctx->dense_blocks_per_cu[0]);
        if (sc.n_julias > 0) {
            RT_HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&ctx->julia_blocks_per_cu[0], trace_julia_kernel<false>, kJuliaBlockThreads, 0));
            RT_HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&ctx->julia_blocks_per_cu[1], trace_julia_kernel<true>, kJuliaBlockThreads, 0));
        }
    }
    ctx->n_objects = d->n_objects;
    ctx->has_scene = true;
    return RT_OK;
}

int rt_photon_generate_device(rt_ctx* ctx, void* hip_stream) {
    if (ctx == nullptr) return RT_ERR_INVALID_ARGUMENT;
    RT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return photon_enqueue(ctx, static_cast<hipStream_t>(hip_stream));
}

int rt_photon_generate_part_device(rt_ctx* ctx, void* hip_stream, uint32_t part, uint32_t n_parts) {
    if (ctx == nullptr) return RT_ERR_INVALID_ARGUMENT;
    if (n_parts == 0 || part >= n_parts) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_photon_generate_part_device: part must be < n_parts");
    RT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return photon_enqueue(ctx, static_cast<hipStream_t>(hip_stream), part, n_parts);
}

int rt_photon_ordered(rt_ctx* ctx, int* ordered_out) {
    if (ctx == nullptr || ordered_out == nullptr) return RT_ERR_INVALID_ARGUMENT;
    *ordered_out = 0;
    if (!ctx->dep_last_pass_ordered) return RT_OK;
    RT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->dep_check_pending) RT_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_dep));
    const bool overflowed = deposit_overflow_seen(ctx) || ctx->dep_last_pass_overflowed;
    ctx->dep_last_pass_overflowed = overflowed;
    if (ctx->dep_misplaced) return fail(ctx, RT_ERR_HIP, "photon pass: the ordered sum met records outside its buffer (internal error: the textures of this pass are incomplete)");
    *ordered_out = overflowed ? 0 : 1;
    return RT_OK;
}

int rt_photon_atlas_device(rt_ctx* ctx, void** d_atlas_out, uint64_t* n_floats_out) {
    if (ctx == nullptr || d_atlas_out == nullptr || n_floats_out == nullptr) return RT_ERR_INVALID_ARGUMENT;
    if (!ctx->has_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_photon_atlas_device before rt_scene_upload");
    *d_atlas_out = ctx->dev.photon_atlas;
    *n_floats_out = ctx->dev.photon_atlas == nullptr ? 0 : static_cast<uint64_t>(ctx->n_slots) * ctx->dev.photon_size * ctx->dev.photon_size * 4;
    return RT_OK;
}

int rt_photon_generate(rt_ctx* ctx, rt_stats* stats) {
    int rc = rt_photon_generate_device(ctx, nullptr);
    if (rc != RT_OK) return rc;
    RT_HIP_TRY(ctx, hipStreamSynchronize(nullptr));
#ifdef RT_GUARD
    {
        unsigned long long w[16] = {};
        RT_HIP_TRY(ctx, hipMemcpy(w, ctx->d_dep_words, sizeof w, hipMemcpyDeviceToHost));
        std::fprintf(stderr, "RT_GUARD: records %llu, live dirs %llu, cursor %llx; first violation: site %llu index %llu limit %llu\n", w[0], w[4], w[5], w[8], w[9], w[10]);
    }
#endif
    if (deposit_overflow_seen(ctx)) {
        // the record buffer was too small for this scene: it has been resized from what the pass reported;
        // run the pass again so that the caller gets the reference's order
        if ((rc = rt_photon_generate_device(ctx, nullptr)) != RT_OK) return rc;
        RT_HIP_TRY(ctx, hipStreamSynchronize(nullptr));
        if (deposit_overflow_seen(ctx)) return fail(ctx, RT_ERR_LIMIT, "photon deposits do not fit the record buffer (%llu wanted)", ctx->h_dep_words[0]);
    }
    if (ctx->dep_misplaced) return fail(ctx, RT_ERR_HIP, "photon pass: the ordered sum met records outside its buffer (internal error: the textures of this pass are incomplete)");
    return read_stats(ctx, stats);
}

int rt_render_device(rt_ctx* ctx, const rt_bands* bands, void* d_rgba_out, void* hip_stream) {
    if (ctx == nullptr) return RT_ERR_INVALID_ARGUMENT;
    if (d_rgba_out == nullptr) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_render_device: output pointer is NULL");
    RT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return render_enqueue(ctx, bands, static_cast<float4*>(d_rgba_out), static_cast<hipStream_t>(hip_stream));
}

int rt_render_device_noalloc(rt_ctx* ctx, const rt_bands* bands, void* d_rgba_out, void* hip_stream) {
    if (ctx == nullptr) return RT_ERR_INVALID_ARGUMENT;
    if (d_rgba_out == nullptr) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_render_device_noalloc: output pointer is NULL");
    RT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->has_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_render_device_noalloc before rt_scene_upload");
    // carve the fixed workspace: header | sample buffer of this frame | frame stacks (whatever is left)
    const uint64_t n_pixels = rt_bands_pixel_count(ctx->dev.width, ctx->dev.height, bands);
    const uint64_t spp = static_cast<uint64_t>(ctx->dev.supersampling) * ctx->dev.supersampling * (ctx->dev.dispersion ? 8u : 1u);
    const uint64_t room = kNoAllocScratchBytes - kNoAllocHeaderBytes;
    uint64_t samples_bytes = spp > 1 ? n_pixels * spp * sizeof(float4) : 0;
    samples_bytes = (samples_bytes + 255) & ~uint64_t(255);
    Workspace ws;
    unsigned char* base = ctx->noalloc_scratch;   // never the array's name: see `scratch`
    ws.queue = reinterpret_cast<unsigned long long*>(base);
    ws.counters = reinterpret_cast<unsigned long long*>(base + 256);
    ws.samples = reinterpret_cast<float4*>(base + kNoAllocHeaderBytes);
    ws.samples_bytes = static_cast<size_t>(std::min<uint64_t>(samples_bytes, room));
    ws.stack = reinterpret_cast<float*>(base + kNoAllocHeaderBytes + ws.samples_bytes);
    ws.stack_bytes = static_cast<size_t>(room - ws.samples_bytes);
    if (samples_bytes > room) ws.samples_bytes = static_cast<size_t>(room), ws.stack_bytes = 0;   // render_enqueue reports the limit
    return render_enqueue(ctx, bands, static_cast<float4*>(d_rgba_out), static_cast<hipStream_t>(hip_stream), &ws);
}

int rt_render(rt_ctx* ctx, const rt_bands* bands, float* rgba_out, rt_stats* stats) {
    if (ctx == nullptr) return RT_ERR_INVALID_ARGUMENT;
    if (rgba_out == nullptr) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_render: output pointer is NULL");
    if (!ctx->has_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_render before rt_scene_upload");
    RT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t n_pixels = rt_bands_pixel_count(ctx->dev.width, ctx->dev.height, bands);
    if (n_pixels == 0) return read_stats(ctx, stats);
    int rc = grow(ctx, ctx->out_staging, n_pixels * sizeof(float4));
    if (rc != RT_OK) return rc;
    rc = render_enqueue(ctx, bands, static_cast<float4*>(ctx->out_staging.ptr), nullptr);
    if (rc != RT_OK) return rc;
    if ((rc = download(ctx, rgba_out, ctx->out_staging.ptr, n_pixels * sizeof(float4))) != RT_OK) return rc;
    return read_stats(ctx, stats);
}

int rt_render_rgba8(rt_ctx* ctx, const rt_bands* bands, uint8_t* rgba8_out, float gain, rt_stats* stats) {
    if (ctx == nullptr) return RT_ERR_INVALID_ARGUMENT;
    if (rgba8_out == nullptr) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_render_rgba8: output pointer is NULL");
    if (!ctx->has_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_render_rgba8 before rt_scene_upload");
    RT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t n_pixels = rt_bands_pixel_count(ctx->dev.width, ctx->dev.height, bands);
    if (n_pixels == 0) return read_stats(ctx, stats);
    int rc = grow(ctx, ctx->out8, n_pixels * 4);
    if (rc != RT_OK) return rc;
    if ((rc = render_enqueue(ctx, bands, nullptr, nullptr, nullptr, static_cast<uchar4*>(ctx->out8.ptr), gain)) != RT_OK) return rc;
    if ((rc = download(ctx, rgba8_out, ctx->out8.ptr, n_pixels * 4)) != RT_OK) return rc;
    return read_stats(ctx, stats);
}

int rt_render_rgba8_device(rt_ctx* ctx, const rt_bands* bands, float gain, void* d_rgba8_out, void* hip_stream) {
    if (ctx == nullptr) return RT_ERR_INVALID_ARGUMENT;
    if (d_rgba8_out == nullptr) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_render_rgba8_device: output pointer is NULL");
    RT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return render_enqueue(ctx, bands, nullptr, static_cast<hipStream_t>(hip_stream), nullptr, static_cast<uchar4*>(d_rgba8_out), gain);
}

int rt_render_accumulate(rt_ctx* ctx, const rt_bands* bands, float weight, int first) {
    if (ctx == nullptr) return RT_ERR_INVALID_ARGUMENT;
    if (!ctx->has_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_render_accumulate before rt_scene_upload");
    RT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t n_pixels = rt_bands_pixel_count(ctx->dev.width, ctx->dev.height, bands);
    if (n_pixels == 0) return RT_OK;
    int rc = grow(ctx, ctx->out_staging, n_pixels * sizeof(float4));
    if (rc != RT_OK) return rc;
    if (ctx->acc.bytes < n_pixels * sizeof(float4) && !first) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_render_accumulate: no accumulated picture of this size (first call must pass first = 1)");
    if ((rc = grow(ctx, ctx->acc, n_pixels * sizeof(float4))) != RT_OK) return rc;
    if (first) RT_HIP_TRY(ctx, hipMemsetAsync(ctx->acc.ptr, 0, n_pixels * sizeof(float4), nullptr));   // Picture acc(width, height): zeros
    if ((rc = render_enqueue(ctx, bands, static_cast<float4*>(ctx->out_staging.ptr), nullptr)) != RT_OK) return rc;
    const uint64_t n_floats = n_pixels * 4;
    hipLaunchKernelGGL(mul_add_kernel, dim3(static_cast<unsigned>((n_floats + 255) / 256)), dim3(256), 0, nullptr, static_cast<float*>(ctx->acc.ptr),
                       static_cast<const float*>(ctx->out_staging.ptr), weight, n_floats);
    RT_HIP_TRY(ctx, hipGetLastError());
    return launch_status(ctx);
}

int rt_accumulated_rgba8(rt_ctx* ctx, const rt_bands* bands, uint8_t* rgba8_out, float gain) {
    if (ctx == nullptr) return RT_ERR_INVALID_ARGUMENT;
    if (rgba8_out == nullptr) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_accumulated_rgba8: output pointer is NULL");
    RT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t n_pixels = rt_bands_pixel_count(ctx->dev.width, ctx->dev.height, bands);
    if (n_pixels == 0) return RT_OK;
    if (ctx->acc.ptr == nullptr || ctx->acc.bytes < n_pixels * sizeof(float4)) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_accumulated_rgba8 before rt_render_accumulate");
    return download_rgba8(ctx, static_cast<const float4*>(ctx->acc.ptr), n_pixels, gain, rgba8_out);
}

int rt_fetch_stats(rt_ctx* ctx, rt_stats* out) {
    if (ctx == nullptr || out == nullptr) return RT_ERR_INVALID_ARGUMENT;
    RT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return read_stats(ctx, out);
}

int rt_launch_status(rt_ctx* ctx) {
    if (ctx == nullptr) return RT_ERR_INVALID_ARGUMENT;
    RT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return launch_status(ctx);
}

int rt_photon_download(rt_ctx* ctx, uint32_t object_index, float* rgba_out) {
    if (ctx == nullptr || rgba_out == nullptr) return RT_ERR_INVALID_ARGUMENT;
    if (!ctx->has_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_photon_download before rt_scene_upload");
    if (object_index >= ctx->n_objects) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "object index %u out of range", object_index);
    const size_t n = static_cast<size_t>(ctx->dev.photon_size) * ctx->dev.photon_size;
    const int32_t slot = ctx->photon_slot[object_index];
    if (slot < 0 || ctx->dev.photon_atlas == nullptr) {
        std::memset(rgba_out, 0, n * sizeof(float4));
        return RT_OK;
    }
    RT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    RT_HIP_TRY(ctx, hipMemcpy(rgba_out, ctx->dev.photon_atlas + n * slot, n * sizeof(float4), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_mul_add_device(rt_ctx* ctx, void* d_acc, const void* d_src, float factor, uint64_t n_floats, void* hip_stream) {
    if (ctx == nullptr) return RT_ERR_INVALID_ARGUMENT;
    if (n_floats == 0) return RT_OK;
    if (d_acc == nullptr || d_src == nullptr) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_mul_add_device: NULL buffer");
    RT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const unsigned g = static_cast<unsigned>((n_floats + 255) / 256);
    hipLaunchKernelGGL(mul_add_kernel, dim3(g), dim3(256), 0, static_cast<hipStream_t>(hip_stream), static_cast<float*>(d_acc), static_cast<const float*>(d_src), factor, n_floats);
    RT_HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

int rt_quantize_device(rt_ctx* ctx, const void* d_rgba, void* d_rgba8, float gain, uint64_t n_pixels, void* hip_stream) {
    if (ctx == nullptr) return RT_ERR_INVALID_ARGUMENT;
    if (n_pixels == 0) return RT_OK;
    if (d_rgba == nullptr || d_rgba8 == nullptr) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_quantize_device: NULL buffer");
    RT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const unsigned g = static_cast<unsigned>((n_pixels + 255) / 256);
    hipLaunchKernelGGL(quantize_kernel, dim3(g), dim3(256), 0, static_cast<hipStream_t>(hip_stream), static_cast<const float4*>(d_rgba), static_cast<uchar4*>(d_rgba8), gain, n_pixels);
    RT_HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

// Scheduler diagnostics of the last counters-enabled launch: out[i] = raw counter word i (layout:
// rtd::CounterSlot in rt_scene_dev.h).  Developer tool (tools/gpu_check.py --debug).
int rt_debug_counters(rt_ctx* ctx, uint64_t* out, uint32_t cap) {
    if (ctx == nullptr || out == nullptr) return RT_ERR_INVALID_ARGUMENT;
    unsigned long long c[C_COUNT];
    RT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    RT_HIP_TRY(ctx, hipMemcpy(c, ctx->d_counters, sizeof c, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < cap && i < C_COUNT; i++) out[i] = c[i];
    return C_COUNT;
}

// The host BVH builder on caller-supplied triangles (no device involved).
int rt_debug_bvh_host(const float* triangles, uint32_t n, uint32_t* records_out) {
    if (triangles == nullptr || records_out == nullptr || n < 2 || n > (1u << 24)) return RT_ERR_INVALID_ARGUMENT;
    std::vector<DevTriTest> tris(n);
    for (uint32_t i = 0; i < n; i++) {
        for (int c = 0; c < 3; c++) {
            tris[i].p0[c] = triangles[i * 9 + c];
            tris[i].e1[c] = triangles[i * 9 + 3 + c];
            tris[i].e2[c] = triangles[i * 9 + 6 + c];
            tris[i].pad[c] = 0.0f;
        }
    }
    static_assert(sizeof(DevBvhPair) == 16 * sizeof(uint32_t), "DevBvhPair");
    build_triangle_bvh_host(tris.data(), 0, n, reinterpret_cast<DevBvhPair*>(records_out));
    return RT_OK;
}

// ---- known-answer hooks (tests only; declared in include/rt_hip.h under "test hooks") ----------------

// Object::intersect (reference src/objects.h:124-126) of scene object object_index[i] for n rays given
// as 7 floats each (origin, direction as passed to the Ray constructor, max_distance);
// out = 12 floats per ray: hit flag, distance, point, normal, uv, photon uv.
int rt_kat_intersect(rt_ctx* ctx, const uint32_t* object_index, const float* rays, uint32_t n, float* out) {
    if (ctx == nullptr) return RT_ERR_INVALID_ARGUMENT;
    if (!ctx->has_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_kat_intersect before rt_scene_upload");
    if (n == 0) return RT_OK;
    if (object_index == nullptr || rays == nullptr || out == nullptr) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_kat_intersect: NULL buffer");
    std::vector<uint32_t> kind(n), prim(n);
    for (uint32_t i = 0; i < n; i++) {
        if (object_index[i] >= ctx->n_objects) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "ray %u: object %u out of range", i, object_index[i]);
        kind[i] = ctx->object_kind[object_index[i]];
        prim[i] = ctx->object_prim[object_index[i]];
    }
    RT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = grow(ctx, ctx->kat_a, n * sizeof(uint32_t))) != RT_OK) return rc;
    if ((rc = grow(ctx, ctx->kat_b, n * sizeof(uint32_t))) != RT_OK) return rc;
    if ((rc = grow(ctx, ctx->kat_c, static_cast<size_t>(n) * 7 * sizeof(float))) != RT_OK) return rc;
    if ((rc = grow(ctx, ctx->kat_d, static_cast<size_t>(n) * 12 * sizeof(float))) != RT_OK) return rc;
    RT_HIP_TRY(ctx, hipMemcpy(ctx->kat_a.ptr, kind.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    RT_HIP_TRY(ctx, hipMemcpy(ctx->kat_b.ptr, prim.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    RT_HIP_TRY(ctx, hipMemcpy(ctx->kat_c.ptr, rays, static_cast<size_t>(n) * 7 * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kat_intersect_kernel, dim3((n + 63) / 64), dim3(64), 0, nullptr, ctx->dev, static_cast<const unsigned*>(ctx->kat_a.ptr),
                       static_cast<const unsigned*>(ctx->kat_b.ptr), static_cast<const float*>(ctx->kat_c.ptr), n, static_cast<float*>(ctx->kat_d.ptr));
    RT_HIP_TRY(ctx, hipGetLastError());
    RT_HIP_TRY(ctx, hipMemcpy(out, ctx->kat_d.ptr, static_cast<size_t>(n) * 12 * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
}

// calcFresnel (reference src/raytracer.cpp:243-268) for n records of 5 floats
// (refraction re, im, dispersion, cos, wavelength); out = kr per record.
int rt_kat_fresnel(rt_ctx* ctx, const float* records, uint32_t n, float* out) {
    if (ctx == nullptr) return RT_ERR_INVALID_ARGUMENT;
    if (n == 0) return RT_OK;
    if (records == nullptr || out == nullptr) return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_kat_fresnel: NULL buffer");
    RT_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = grow(ctx, ctx->kat_c, static_cast<size_t>(n) * 5 * sizeof(float))) != RT_OK) return rc;
    if ((rc = grow(ctx, ctx->kat_d, static_cast<size_t>(n) * sizeof(float))) != RT_OK) return rc;
    RT_HIP_TRY(ctx, hipMemcpy(ctx->kat_c.ptr, records, static_cast<size_t>(n) * 5 * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kat_fresnel_kernel, dim3((n + 63) / 64), dim3(64), 0, nullptr, static_cast<const float*>(ctx->kat_c.ptr), n, static_cast<float*>(ctx->kat_d.ptr));
    RT_HIP_TRY(ctx, hipGetLastError());
    RT_HIP_TRY(ctx, hipMemcpy(out, ctx->kat_d.ptr, static_cast<size_t>(n) * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
}

}  // extern "C"
