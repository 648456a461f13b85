// Photon deposits in the reference's order (RT_OPT_PHOTON_ORDER = 1).
//
// PhotonMapper::generate is sequential: every deposit is `texture[texel] += radiance`
// (src/objects.cpp:244-256) in the order lights x phi x theta x hue, and within one root photon in the
// order of the recursion (refraction subtree, then reflection; src/photonmap.cpp:118-144).  Float
// addition does not commute in the last bit, so float atomics give textures that agree to rounding
// only.  Here the deposits recorded by photon_kernel — key = (texel, root id, number within the root),
// exactly that order — are brought into key order per texel and every texel's run is added up front to
// back, starting from the zero the texture is created with: the same additions in the same order as
// the reference, hence the same floats.
//
// photon_order_bucket (the default): no global sort.  What matters is the order WITHIN a texel, and runs
// are short (C5: 0.37 M deposits in 16 k texels, the longest run 201; 9_caustic: 2.1 M in 66 k, longest
// 288): photon_kernel counts the deposits per texel while it records them; every non-empty texel gets
// a segment, the records are scattered into their segments, and one wave per texel ranks its run in
// LDS (rank = number of smaller keys) and adds it up in rank order, four lanes adding one channel each.
// Three launches instead of the thirty-odd
// of a 52-bit device-wide sort [MI355X, C5: 0.25 -> see DESIGN.md 4.6].  Runs beyond the LDS tile are
// ranked against global memory (slow, correct for any length).
//
// photon_order_reduce (the fallback, for atlases of 2^31 texels or more): rocPRIM radix/merge sort over
// the significant key bits with the record index as payload, one thread per short run, a wave per long
// one (>= kDepositLongRun: 64 records staged in LDS per round).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#include <cstring>   // rocprim's texture_cache_iterator.hpp uses memset without including it

#include <rocprim/rocprim.hpp>

#include "rt_scene_dev.h"

namespace rtd {

namespace {

__device__ __forceinline__ unsigned long long texel_of(unsigned long long key, unsigned shift) { return key >> shift; }

// one thread per sorted record; the thread of a run's first record adds the run up
__global__ void sum_short_runs_kernel(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ index,
                                      const float4* __restrict__ vals, unsigned n, unsigned shift, unsigned long long n_texels,
                                      float4* __restrict__ atlas, unsigned* __restrict__ long_runs, unsigned long long* __restrict__ n_long_runs) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long texel = texel_of(keys[i], shift);
    if (texel >= n_texels) return;                                   // unused slot (key ~0)
    if (i > 0 && texel_of(keys[i - 1], shift) == texel) return;      // not the first of its run
    // length of the run (the keys are contiguous: a cache line holds eight)
    unsigned len = 1;
    while (len <= kDepositLongRun && i + len < n && texel_of(keys[i + len], shift) == texel) len++;
    if (len > kDepositLongRun) {
        long_runs[atomicAdd(n_long_runs, 1ull)] = i;
        return;
    }
    // the texel holds zero — or, when the record buffer ran full, what the surplus deposits added with atomics
    float4 acc = atlas[texel];
    for (unsigned j = 0; j < len; j += 8) {
        // eight gathers in flight, then the additions in order
        float4 v[8];
#pragma unroll
        for (unsigned k = 0; k < 8; k++) {
            if (j + k < len) v[k] = vals[index[i + j + k]];
        }
#pragma unroll
        for (unsigned k = 0; k < 8; k++) {
            if (j + k < len) {
                acc.x += v[k].x;
                acc.y += v[k].y;
                acc.z += v[k].z;
                acc.w += v[k].w;
            }
        }
    }
    atlas[texel] = acc;
}

// one 64-lane workgroup per long run (grid-stride over the list)
__global__ __launch_bounds__(64) void sum_long_runs_kernel(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ index,
                                                           const float4* __restrict__ vals, unsigned n, unsigned shift,
                                                           float4* __restrict__ atlas, const unsigned* __restrict__ long_runs,
                                                           const unsigned long long* __restrict__ n_long_runs) {
    __shared__ float staged[64][4];
    const unsigned lane = threadIdx.x;
    const unsigned long long n_runs = *n_long_runs;
    for (unsigned long long r = blockIdx.x; r < n_runs; r += gridDim.x) {
        const unsigned first = long_runs[r];
        const unsigned long long texel = texel_of(keys[first], shift);
        float acc = 0.0f;                                            // lanes 0..3: one channel each
        for (unsigned base = first;; base += 64) {
            const unsigned j = base + lane;
            const bool in_run = j < n && texel_of(keys[j], shift) == texel;
            if (in_run) {
                const float4 v = vals[index[j]];
                staged[lane][0] = v.x;
                staged[lane][1] = v.y;
                staged[lane][2] = v.z;
                staged[lane][3] = v.w;
            }
            const unsigned count = static_cast<unsigned>(__popcll(__ballot(in_run)));   // the run is contiguous: lanes 0..count-1
            __syncthreads();
            if (lane < 4) {
                for (unsigned k = 0; k < count; k++) acc += staged[k][lane];
            }
            __syncthreads();
            if (count < 64) break;
        }
        if (lane < 4) reinterpret_cast<float*>(atlas + texel)[lane] += acc;
    }
}

// ---- photon_order_bucket ------------------------------------------------------------------------------------------------
#ifdef RT_GUARD
// developer build (tools/guard_build.sh): index checks; the first violation goes to cursor[3..5] = control words 8..10 (site, index, limit) and the access is skipped
#define RT_PO_OK(cursor, site, index, limit) \
    (static_cast<unsigned long long>(index) < static_cast<unsigned long long>(limit) ? true : \
     (atomicCAS(const_cast<unsigned long long*>(cursor) + 3, 0ull, static_cast<unsigned long long>(site)) == 0ull ? (const_cast<unsigned long long*>(cursor)[4] = static_cast<unsigned long long>(index), const_cast<unsigned long long*>(cursor)[5] = static_cast<unsigned long long>(limit), false) : false))
#else
#define RT_PO_OK(cursor, site, index, limit) true
#endif
constexpr unsigned kRunTile = 320;       // records of a run that a wave ranks and sums in LDS at a time (7.5 KB: twenty-one waves per CU;
                                         // the kernel is a chain of dependent loads per run, so waves in flight are what it needs)

// One thread per texel: every non-empty texel gets a segment of the scattered arrays and an entry {texel, first, length} in
// the run list.  Segments and entries are handed out block by block with one atomic (low word: records, high word: runs) —
// where a texel's segment lies does not matter.
__global__ __launch_bounds__(256) void place_runs_kernel(unsigned* __restrict__ count, unsigned n_texels, unsigned* __restrict__ seg_first,
                                                         uint4* __restrict__ runs, unsigned long long* __restrict__ cursor, unsigned capacity) {
    __shared__ unsigned long long wave_total[4];
    __shared__ unsigned long long block_base;
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    const unsigned c = t < n_texels ? count[t] : 0u;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned long long v = static_cast<unsigned long long>(c) | (c != 0u ? 1ull << 32 : 0ull);   // inclusive scan of (records, runs) over the block
    for (unsigned off = 1; off < 64; off <<= 1) {
        const unsigned long long up = __shfl_up(v, off);
        if (lane >= off) v += up;
    }
    if (lane == 63) wave_total[wave] = v;
    __syncthreads();
    unsigned long long before = 0;
    for (unsigned w = 0; w < wave; w++) before += wave_total[w];
    if (threadIdx.x == 255) {
        const unsigned long long total = before + v;
        block_base = total != 0ull ? atomicAdd(cursor, total) : 0ull;
    }
    __syncthreads();
    if (c != 0u) {
        const unsigned long long mine = block_base + before + v - (static_cast<unsigned long long>(c) | (1ull << 32));   // exclusive
        const unsigned first = static_cast<unsigned>(mine), slot = static_cast<unsigned>(mine >> 32);
        seg_first[t] = first;
        (void)RT_PO_OK(cursor, 102, static_cast<unsigned long long>(first) + c - 1ull, capacity);
        if (RT_PO_OK(cursor, 101, slot, capacity) && slot < capacity) runs[slot] = make_uint4(t, first, c, 0u);   // (slot < capacity with sound counts: a run has a record)
        else atomicAdd(cursor + 1, 1ull);
        count[t] = 0u;                                  // for the next pass
    }
}

// Records into their texels' segments: record i goes to the place photon_kernel's count gave it (slot[i] = the texel's
// count when it arrived), so no atomic is needed here.
__global__ __launch_bounds__(256) void scatter_records_kernel(const unsigned long long* __restrict__ keys, unsigned capacity, unsigned shift,
                                                              unsigned long long n_texels, const unsigned* __restrict__ slot, const unsigned* __restrict__ seg_first,
                                                              const float4* __restrict__ vals, unsigned long long* __restrict__ seg_keys, float4* __restrict__ seg_vals,
                                                              const unsigned long long* __restrict__ cursor) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= capacity) return;
    const unsigned long long key = keys[i];
    const unsigned long long texel = texel_of(key, shift);
    if (texel >= n_texels) return;
    const unsigned at = seg_first[texel] + slot[i];
    if (!RT_PO_OK(cursor, 111, at, capacity)) return;
    if (at >= capacity) {       // never with sound counts (the segments partition the records): reported, not written
        atomicAdd(const_cast<unsigned long long*>(cursor) + 1, 1ull);
        return;
    }
    seg_keys[at] = key;
    seg_vals[at] = vals[i];
}

__device__ __forceinline__ bool key_before(unsigned long long a, unsigned ia, unsigned long long b, unsigned ib) { return a < b || (a == b && ia < ib); }

// acc + v[0] + v[1] + ... in that order, channel `c`: eight LDS reads in flight, then their additions (a read per addition
// would wait out the LDS latency n times over)
__device__ __forceinline__ float add_in_order(float acc, const float4* v, unsigned n, unsigned c) {
    unsigned j = 0;
    for (; j + 8 <= n; j += 8) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; k++) t[k] = reinterpret_cast<const float*>(&v[j + k])[c];
#pragma unroll
        for (int k = 0; k < 8; k++) acc += t[k];
    }
    for (; j < n; j++) acc += reinterpret_cast<const float*>(&v[j])[c];
    return acc;
}

// One wave per run (grid-stride over the run list): rank, gather in rank order, add.  A run is a chain of dependent
// accesses — run count, descriptor, keys, ranks in LDS, values, the sum — and C5's 16 000 runs (372 000 records, 23 on
// average, 201 at most) are two rounds of such chains for the chip's wave slots, so the kernel's time is the chains'
// length: the first descriptor is read beside the run count, and a run's values beside its keys (into registers; they
// go to LDS at their rank), two memory round trips per run instead of four.
__global__ __launch_bounds__(64) void sum_runs_kernel(const uint4* __restrict__ runs, unsigned capacity, const unsigned long long* __restrict__ cursor,
                                                      const unsigned long long* __restrict__ seg_keys, const float4* __restrict__ seg_vals,
                                                      float4* __restrict__ atlas, unsigned* __restrict__ order, unsigned n_texels) {
    const unsigned guard_texels = n_texels;
    __shared__ unsigned long long lkeys[kRunTile];
    __shared__ float4 lvals[kRunTile];
    constexpr unsigned kPerLane = kRunTile / 64u;
    static_assert(kRunTile % 64u == 0u, "a tile is a whole number of wave-wide rows");
    const unsigned lane = threadIdx.x;
    uint4 run = blockIdx.x < capacity ? runs[blockIdx.x] : uint4{0u, 0u, 0u, 0u};    // (entries from the run count up hold whatever the last pass left)
    const unsigned counted = static_cast<unsigned>(*cursor >> 32);
    const unsigned n_runs = counted < capacity ? counted : capacity;
    for (unsigned r = blockIdx.x; r < n_runs; r += gridDim.x) {
        if (r != blockIdx.x) run = runs[r];
        const unsigned texel = run.x, first = run.y, len = run.z;
#ifdef RT_GUARD
        if (!RT_PO_OK(cursor, 121, texel, guard_texels) || !RT_PO_OK(cursor, 122, static_cast<unsigned long long>(first) + len - 1ull, capacity) || len == 0u) continue;
#endif
        if (texel >= guard_texels || static_cast<unsigned long long>(first) + len > capacity) {     // (as in scatter_records_kernel)
            if (lane == 0) atomicAdd(const_cast<unsigned long long*>(cursor) + 1, 1ull);
            continue;
        }
        float acc = lane < 4 ? reinterpret_cast<const float*>(atlas + texel)[lane] : 0.0f;   // zero, or the surplus of a pass that outgrew the record buffer
        if (len <= kRunTile) {
            unsigned long long k[kPerLane];
            float4 v[kPerLane];
#pragma unroll
            for (unsigned i = 0; i < kPerLane; i++) {
                const unsigned e = lane + 64u * i;
                if (e < len) {
                    k[i] = seg_keys[first + e];
                    v[i] = seg_vals[first + e];
                }
            }
#pragma unroll
            for (unsigned i = 0; i < kPerLane; i++) {
                const unsigned e = lane + 64u * i;
                if (e < len) lkeys[e] = k[i];
            }
            __syncthreads();
#pragma unroll
            for (unsigned i = 0; i < kPerLane; i++) {
                const unsigned e = lane + 64u * i;
                if (e < len) {
                    unsigned rank = 0;
                    for (unsigned j = 0; j < len; j++) rank += key_before(lkeys[j], j, k[i], e) ? 1u : 0u;
                    if (RT_PO_OK(cursor, 123, rank, len)) lvals[rank] = v[i];
                }
            }
            __syncthreads();
            if (lane < 4) acc = add_in_order(acc, lvals, len, lane);
            __syncthreads();
        } else {
            // a run beyond the tile: ranks against global memory (the run's own segment of `order` takes the record indices
            // in key order), then the sum tile by tile
            for (unsigned e = lane; e < len; e += 64) {
                const unsigned long long k = seg_keys[first + e];
                unsigned rank = 0;
                for (unsigned j = 0; j < len; j++) rank += key_before(seg_keys[first + j], j, k, e) ? 1u : 0u;
                if (RT_PO_OK(cursor, 124, rank, len)) order[first + rank] = first + e;
            }
            __threadfence_block();
            __syncthreads();
            for (unsigned base = 0; base < len; base += kRunTile) {
                const unsigned n = len - base < kRunTile ? len - base : kRunTile;
                for (unsigned e = lane; e < n; e += 64) {
                    const unsigned from = order[first + base + e];
                    lvals[e] = RT_PO_OK(cursor, 125, from, capacity) ? seg_vals[from] : float4{0.0f, 0.0f, 0.0f, 0.0f};
                }
                __syncthreads();
                if (lane < 4) acc = add_in_order(acc, lvals, n, lane);
                __syncthreads();
            }
        }
        if (lane < 4) reinterpret_cast<float*>(atlas + texel)[lane] = acc;
    }
}

}  // namespace

// The ordered sums without a global sort (see the head of this file).  count: [n_texels], the deposits per texel as
// photon_kernel counted them while it recorded (DevPhotonJob::dep_count; left at zero again); slot: [capacity], a
// record's number within its texel (DevPhotonJob::dep_slot); seg_first: [n_texels]; runs, seg_keys, seg_vals, order:
// [capacity]; cursor: one word, zero.
hipError_t photon_order_bucket(const unsigned long long* keys, const float4* vals, const unsigned* slot, unsigned capacity, unsigned texel_shift, unsigned n_texels,
                               float4* atlas, unsigned* count, unsigned* seg_first, uint4* runs, unsigned long long* seg_keys, float4* seg_vals,
                               unsigned* order, unsigned long long* cursor, hipStream_t stream) {
    // RT_HIP_SYNC_EACH=1 (developer aid, as in rt_api.hip): wait after every kernel and say on stderr which one failed
    static const bool sync_each = [] { const char* v = std::getenv("RT_HIP_SYNC_EACH"); return v != nullptr && v[0] != '0' && v[0] != 0; }();
    auto after = [&](const char* what) {
        hipError_t e = hipGetLastError();
        if (e == hipSuccess && sync_each) e = hipStreamSynchronize(stream);
        if (e != hipSuccess && sync_each) std::fprintf(stderr, "photon_order_bucket: %s: %s\n", what, hipGetErrorString(e));
        return e;
    };
    hipError_t e;
    hipLaunchKernelGGL(place_runs_kernel, dim3((n_texels + 255) / 256), dim3(256), 0, stream, count, n_texels, seg_first, runs, cursor, capacity);
    if ((e = after("place_runs_kernel")) != hipSuccess) return e;
    hipLaunchKernelGGL(scatter_records_kernel, dim3((capacity + 255) / 256), dim3(256), 0, stream, keys, capacity, texel_shift,
                       static_cast<unsigned long long>(n_texels), slot, seg_first, vals, seg_keys, seg_vals, cursor);
    if ((e = after("scatter_records_kernel")) != hipSuccess) return e;
    const unsigned blocks = capacity < 16384u ? (capacity > 0u ? capacity : 1u) : 16384u;
    hipLaunchKernelGGL(sum_runs_kernel, dim3(blocks), dim3(64), 0, stream, runs, capacity, cursor, seg_keys, seg_vals, atlas, order, n_texels);
    return after("sum_runs_kernel");
}

// Bytes of temporary storage the sort needs for `capacity` records
hipError_t photon_order_temp_bytes(unsigned capacity, unsigned key_bits, size_t* bytes) {
    *bytes = 0;
    return rocprim::radix_sort_pairs(nullptr, *bytes, static_cast<const unsigned long long*>(nullptr), static_cast<unsigned long long*>(nullptr),
                                     rocprim::counting_iterator<unsigned>(0), static_cast<unsigned*>(nullptr), capacity, 0u, key_bits, hipStream_t(nullptr));
}

// keys/vals: what photon_kernel recorded (unused keys ~0).  Adds every texel's run to the atlas, which holds zeros
// (0 + x = x exactly) or the out-of-order surplus of a pass that outgrew the record buffer.  sorted_keys/sorted_index/long_runs: [capacity] scratch; words[2] = long-run counter.
hipError_t photon_order_reduce(const unsigned long long* keys, const float4* vals, unsigned capacity, unsigned key_bits, unsigned texel_shift,
                               unsigned long long n_texels, float4* atlas, unsigned long long* sorted_keys, unsigned* sorted_index,
                               unsigned* long_runs, unsigned long long* words, void* temp, size_t temp_bytes, hipStream_t stream) {
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys, sorted_keys, rocprim::counting_iterator<unsigned>(0), sorted_index, capacity, 0u,
                                             key_bits, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sum_short_runs_kernel, dim3((capacity + 255) / 256), dim3(256), 0, stream, sorted_keys, sorted_index, vals, capacity, texel_shift,
                       n_texels, atlas, long_runs, words + 2);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(sum_long_runs_kernel, dim3(512), dim3(64), 0, stream, sorted_keys, sorted_index, vals, capacity, texel_shift, atlas, long_runs, words + 2);
    return hipGetLastError();
}

}  // namespace rtd
