// Photon deposits in the reference's order (RT_OPT_PHOTON_ORDER = 1).
//
// PhotonMapper::generate is sequential: every deposit is `texture[texel] += radiance`
// (src/objects.cpp:244-256) in the order lights x phi x theta x hue, and within one root photon in the
// order of the recursion (refraction subtree, then reflection; src/photonmap.cpp:118-144).  Float
// addition does not commute in the last bit, so float atomics give textures that agree to rounding
// only.  Here the deposits recorded by photon_kernel — key = (texel, root id, number within the root),
// exactly that order — are sorted (rocPRIM radix sort over the significant key bits, payload = record
// index) and every texel's run is added up front to back, starting from the zero the texture is
// created with: the same additions in the same order as the reference, hence the same floats.
//
// Runs are short almost everywhere (one thread per run); a texel in a caustic's focus can receive
// thousands of deposits, and such a run (>= kDepositLongRun) is summed by a wave: 64 records staged in
// LDS per round, four lanes adding one channel each, so the chain costs an LDS read and an add per
// record instead of a dependent global load.
#include <hip/hip_runtime.h>

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

}  // namespace

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
