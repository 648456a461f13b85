// rt_bvh.hip — GPU construction of the triangle BVH (LBVH: Morton codes, radix sort, Karras 2012
// hierarchy, bottom-up bounds).  No counterpart in the reference, which tests every object for every
// ray (reference src/raytracer.cpp:126); the traversal in rt_kernels.hip returns the same winner as
// that sequential scan (nearest hit, equal distances to the later scene object).
//
// Construction layout: a run of n >= 2 triangles gets 2n-1 binary nodes: internal nodes 0..n-2 (root 0), leaves
// n-1..2n-2 in Morton order.  Node = {lo[3], a, hi[3], b}: internal a/b = child node indices (relative
// to the run's first node); leaf a = index into the triangle tables, b = kBvhLeaf.
// Traversal layout (what build_triangle_bvh returns): the n-1 internal nodes as DevBvhPair records, each with the
// boxes of BOTH children (bvh_pack_kernel); leaves are folded into their parent's child word.
//
// Leaf boxes are conservative for Triangle::intersect as written (src/objects.cpp:66-76): the test
// accepts barycentrics down to -EPSILON, i.e. points up to EPSILON*(|e1|+|e2|) outside the triangle;
// boxes are grown by twice that plus a relative slack for rounding.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <vector>

#include "rt_scene_dev.h"

namespace rtd {

constexpr uint32_t kBvhLeaf = 0xffffffffu;

namespace {

__device__ __forceinline__ unsigned ordered(float f) {   // monotone float -> uint
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unordered(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

struct Box {
    float lo[3], hi[3];
};

__global__ void bvh_bounds_kernel(const DevTriTest* __restrict__ tris, uint32_t first, uint32_t n, Box* __restrict__ boxes,
                                  unsigned* __restrict__ scene_bounds /* lo[3], hi[3] ordered */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DevTriTest t = tris[first + i];
    Box b;
    for (int c = 0; c < 3; c++) {
        const float p0 = t.p0[c], p1 = t.p0[c] + t.e1[c], p2 = t.p0[c] + t.e2[c];
        const float lo = fminf(p0, fminf(p1, p2)), hi = fmaxf(p0, fmaxf(p1, p2));
        const float grow = 2.0f * 0.0001f * (fabsf(t.e1[c]) + fabsf(t.e2[c])) + 4e-6f * fmaxf(fabsf(lo), fabsf(hi)) + 1e-30f;
        b.lo[c] = lo - grow;
        b.hi[c] = hi + grow;
        atomicMin(&scene_bounds[c], ordered(0.5f * (lo + hi)));
        atomicMax(&scene_bounds[3 + c], ordered(0.5f * (lo + hi)));
    }
    boxes[i] = b;
}

__device__ __forceinline__ unsigned expand_bits(unsigned v) {   // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ void bvh_morton_kernel(const Box* __restrict__ boxes, uint32_t n, const unsigned* __restrict__ scene_bounds,
                                  unsigned long long* __restrict__ keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned code = 0;
    for (int c = 0; c < 3; c++) {
        const float lo = unordered(scene_bounds[c]), hi = unordered(scene_bounds[3 + c]);
        const float mid = 0.5f * (boxes[i].lo[c] + boxes[i].hi[c]);
        const float ext = hi - lo;
        float x = ext > 0.0f ? (mid - lo) / ext : 0.0f;
        x = fminf(fmaxf(x * 1024.0f, 0.0f), 1023.0f);
        code |= expand_bits(static_cast<unsigned>(x)) << (2 - c);
    }
    keys[i] = (static_cast<unsigned long long>(code) << 32) | i;   // unique keys: ties broken by index
}

__device__ __forceinline__ int common_prefix(const unsigned long long* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll(static_cast<long long>(keys[i] ^ keys[j]));
}

// Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees" (2012)
__global__ void bvh_hierarchy_kernel(const unsigned long long* __restrict__ keys, int n, DevBvhNode* __restrict__ nodes,
                                     unsigned* __restrict__ parent) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = common_prefix(keys, n, i, i + 1) > common_prefix(keys, n, i, i - 1) ? 1 : -1;
    const int dmin = common_prefix(keys, n, i, i - d);
    int lmax = 2;
    while (common_prefix(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2) {
        if (common_prefix(keys, n, i, i + (l + t) * d) > dmin) l += t;
    }
    const int j = i + l * d;
    const int dnode = common_prefix(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (common_prefix(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const unsigned left = (lo == gamma) ? static_cast<unsigned>(n - 1 + gamma) : static_cast<unsigned>(gamma);
    const unsigned right = (hi == gamma + 1) ? static_cast<unsigned>(n - 1 + gamma + 1) : static_cast<unsigned>(gamma + 1);
    nodes[i].left_or_first = left;
    nodes[i].count = right;
    parent[left] = static_cast<unsigned>(i);
    parent[right] = static_cast<unsigned>(i);
    if (i == 0) parent[0] = 0xffffffffu;
}

__global__ void bvh_fit_kernel(const unsigned long long* __restrict__ keys, const Box* __restrict__ boxes, uint32_t first, int n,
                               DevBvhNode* __restrict__ nodes, const unsigned* __restrict__ parent, unsigned* __restrict__ arrivals) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const unsigned prim = static_cast<unsigned>(keys[j] & 0xffffffffull);
    DevBvhNode leaf;
    for (int c = 0; c < 3; c++) {
        leaf.lo[c] = boxes[prim].lo[c];
        leaf.hi[c] = boxes[prim].hi[c];
    }
    leaf.left_or_first = first + prim;
    leaf.count = kBvhLeaf;
    nodes[n - 1 + j] = leaf;
    __threadfence();
    unsigned node = parent[n - 1 + j];
    while (node != 0xffffffffu) {
        if (atomicAdd(&arrivals[node], 1u) == 0u) return;   // the second child to arrive fits the parent
        __threadfence();
        const volatile DevBvhNode* a = &nodes[nodes[node].left_or_first];
        const volatile DevBvhNode* b = &nodes[nodes[node].count];
        for (int c = 0; c < 3; c++) {
            nodes[node].lo[c] = fminf(a->lo[c], b->lo[c]);
            nodes[node].hi[c] = fmaxf(a->hi[c], b->hi[c]);
        }
        __threadfence();
        node = parent[node];
    }
}

__global__ void bvh_pack_kernel(const DevBvhNode* __restrict__ nodes, int n, DevBvhPair* __restrict__ pairs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const DevBvhNode a = nodes[nodes[i].left_or_first], b = nodes[nodes[i].count];
    DevBvhPair p;
    for (int c = 0; c < 3; c++) {
        p.lo0[c] = a.lo[c];
        p.hi0[c] = a.hi[c];
        p.lo1[c] = b.lo[c];
        p.hi1[c] = b.hi[c];
    }
    p.child0 = a.count == kBvhLeaf ? (kBvhLeafBit | a.left_or_first) : nodes[i].left_or_first;
    p.child1 = b.count == kBvhLeaf ? (kBvhLeafBit | b.left_or_first) : nodes[i].count;
    p.slack0 = 1e-4f * ((a.hi[0] - a.lo[0]) + (a.hi[1] - a.lo[1]) + (a.hi[2] - a.lo[2]));
    p.slack1 = 1e-4f * ((b.hi[0] - b.lo[0]) + (b.hi[1] - b.lo[1]) + (b.hi[2] - b.lo[2]));
    pairs[i] = p;
}

}  // namespace

// Builds the BVH of triangles [first, first+n) of `tris` into pairs[0 .. n-2].  Synchronous.
hipError_t build_triangle_bvh(const DevTriTest* tris, uint32_t first, uint32_t n, DevBvhPair* pairs) {
    if (n < 2 || first + n >= kBvhLeafBit) return hipErrorInvalidValue;
    DevBvhNode* nodes = nullptr;
    Box* boxes = nullptr;
    unsigned* bounds = nullptr;
    unsigned long long *keys = nullptr, *keys_sorted = nullptr;
    unsigned *parent = nullptr, *arrivals = nullptr;
    void* temp = nullptr;
    size_t temp_bytes = 0;
    hipError_t e = hipSuccess;
    auto done = [&](hipError_t err) {
        (void)hipFree(nodes); (void)hipFree(boxes); (void)hipFree(bounds); (void)hipFree(keys); (void)hipFree(keys_sorted);
        (void)hipFree(parent); (void)hipFree(arrivals); (void)hipFree(temp);
        return err;
    };
    if ((e = hipMalloc(reinterpret_cast<void**>(&nodes), (2 * size_t(n) - 1) * sizeof(DevBvhNode))) != hipSuccess) return done(e);
    if ((e = hipMalloc(reinterpret_cast<void**>(&boxes), n * sizeof(Box))) != hipSuccess) return done(e);
    if ((e = hipMalloc(reinterpret_cast<void**>(&bounds), 6 * sizeof(unsigned))) != hipSuccess) return done(e);
    if ((e = hipMalloc(reinterpret_cast<void**>(&keys), n * sizeof(unsigned long long))) != hipSuccess) return done(e);
    if ((e = hipMalloc(reinterpret_cast<void**>(&keys_sorted), n * sizeof(unsigned long long))) != hipSuccess) return done(e);
    if ((e = hipMalloc(reinterpret_cast<void**>(&parent), (2 * size_t(n) - 1) * sizeof(unsigned))) != hipSuccess) return done(e);
    if ((e = hipMalloc(reinterpret_cast<void**>(&arrivals), n * sizeof(unsigned))) != hipSuccess) return done(e);
    const unsigned init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    if ((e = hipMemcpy(bounds, init, sizeof init, hipMemcpyHostToDevice)) != hipSuccess) return done(e);
    if ((e = hipMemset(arrivals, 0, n * sizeof(unsigned))) != hipSuccess) return done(e);
    const unsigned g = (n + 255) / 256;
    hipLaunchKernelGGL(bvh_bounds_kernel, dim3(g), dim3(256), 0, nullptr, tris, first, n, boxes, bounds);
    hipLaunchKernelGGL(bvh_morton_kernel, dim3(g), dim3(256), 0, nullptr, boxes, n, bounds, keys);
    if ((e = hipcub::DeviceRadixSort::SortKeys(nullptr, temp_bytes, keys, keys_sorted, static_cast<int>(n))) != hipSuccess) return done(e);
    if ((e = hipMalloc(&temp, temp_bytes ? temp_bytes : 16)) != hipSuccess) return done(e);
    if ((e = hipcub::DeviceRadixSort::SortKeys(temp, temp_bytes, keys, keys_sorted, static_cast<int>(n))) != hipSuccess) return done(e);
    hipLaunchKernelGGL(bvh_hierarchy_kernel, dim3(g), dim3(256), 0, nullptr, keys_sorted, static_cast<int>(n), nodes, parent);
    hipLaunchKernelGGL(bvh_fit_kernel, dim3(g), dim3(256), 0, nullptr, keys_sorted, boxes, first, static_cast<int>(n), nodes, parent, arrivals);
    hipLaunchKernelGGL(bvh_pack_kernel, dim3(g), dim3(256), 0, nullptr, nodes, static_cast<int>(n), pairs);
    if ((e = hipGetLastError()) != hipSuccess) return done(e);
    e = hipDeviceSynchronize();
    return done(e);
}

// ---- host builder for short runs -------------------------------------------------------------------------------------------
// A Morton-order tree is a poor one for a handful of large and small triangles (C2: five boxes and two planes, 64
// triangles): a ray visited 16 records of its 63.  Runs of at most kBvhHostBuildMax triangles get a full-sweep
// surface-area-heuristic tree built here instead — microseconds on one core, no device allocation, no launch —; any
// tree gives the same pictures (the traversal's winner does not depend on it).  Same leaf boxes as bvh_bounds_kernel;
// records in depth-first order (a record's first child follows it); depth bounded for the traversal's stack.
namespace {

struct HostBox {
    float lo[3], hi[3];
    void grow(const HostBox& o) {
        for (int c = 0; c < 3; c++) {
            lo[c] = std::fmin(lo[c], o.lo[c]);
            hi[c] = std::fmax(hi[c], o.hi[c]);
        }
    }
    double half_area() const {
        const double x = double(hi[0]) - lo[0], y = double(hi[1]) - lo[1], z = double(hi[2]) - lo[2];
        return x * y + y * z + z * x;
    }
};
constexpr HostBox kEmptyBox = {{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};

// The three centroid orders are sorted once; a split partitions each of them stably, so every node sweeps sorted
// ranges without sorting again (O(n log n) in all).
struct HostBuilder {
    const std::vector<HostBox>& boxes;
    uint32_t first;
    DevBvhPair* pairs;
    uint32_t next = 0;
    std::vector<uint32_t> order[3], scratch;
    std::vector<unsigned char> goes_left;
    std::vector<double> right_area;

    HostBuilder(const std::vector<HostBox>& b, uint32_t first_prim, DevBvhPair* out) : boxes(b), first(first_prim), pairs(out) {
        const size_t n = boxes.size();
        scratch.resize(n);
        goes_left.resize(n);
        right_area.resize(n);
        for (int axis = 0; axis < 3; axis++) {
            std::vector<uint32_t>& o = order[axis];
            o.resize(n);
            for (size_t i = 0; i < n; i++) o[i] = static_cast<uint32_t>(i);
            std::sort(o.begin(), o.end(), [&](uint32_t x, uint32_t y) {
                const float cx = boxes[x].lo[axis] + boxes[x].hi[axis], cy = boxes[y].lo[axis] + boxes[y].hi[axis];
                return cx < cy || (!(cy < cx) && x < y);   // (NaN centroids compare as equal: ordered by index)
            });
        }
    }

    // Builds the subtree of the triangles in [lo, hi) of the three orders and returns its child word.
    uint32_t build(size_t lo, size_t hi, int depth) {
        if (hi - lo == 1) return kBvhLeafBit | (first + order[0][lo]);
        const uint32_t me = next++;
        const size_t n = hi - lo;
        size_t best_at = n / 2;
        int best_axis = 0;
        double best_cost = INFINITY;
        // (past the depth limit, or with no finite cost — NaN or infinite boxes —: the median of the x order)
        for (int axis = 0; axis < 3 && depth < 32; axis++) {
            const uint32_t* o = order[axis].data() + lo;
            HostBox acc = kEmptyBox;
            for (size_t i = n; i-- > 1;) {
                acc.grow(boxes[o[i]]);
                right_area[i] = acc.half_area();
            }
            acc = kEmptyBox;
            for (size_t i = 1; i < n; i++) {
                acc.grow(boxes[o[i - 1]]);
                const double cost = acc.half_area() * double(i) + right_area[i] * double(n - i);
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = axis;
                    best_at = i;
                }
            }
        }
        HostBox b0 = kEmptyBox, b1 = kEmptyBox;
        {
            const uint32_t* o = order[best_axis].data() + lo;
            for (size_t i = 0; i < n; i++) {
                goes_left[o[i]] = i < best_at;
                (i < best_at ? b0 : b1).grow(boxes[o[i]]);
            }
        }
        for (int axis = 0; axis < 3; axis++) {
            if (axis == best_axis) continue;
            uint32_t* o = order[axis].data() + lo;
            size_t l = 0, r = 0;
            for (size_t i = 0; i < n; i++) {
                if (goes_left[o[i]]) o[l++] = o[i];
                else scratch[r++] = o[i];
            }
            std::copy(scratch.begin(), scratch.begin() + r, o + l);
        }
        const uint32_t c0 = build(lo, lo + best_at, depth + 1);
        const uint32_t c1 = build(lo + best_at, hi, depth + 1);
        DevBvhPair& p = pairs[me];
        for (int c = 0; c < 3; c++) {
            p.lo0[c] = b0.lo[c];
            p.hi0[c] = b0.hi[c];
            p.lo1[c] = b1.lo[c];
            p.hi1[c] = b1.hi[c];
        }
        p.child0 = c0;
        p.child1 = c1;
        p.slack0 = 1e-4f * ((b0.hi[0] - b0.lo[0]) + (b0.hi[1] - b0.lo[1]) + (b0.hi[2] - b0.lo[2]));
        p.slack1 = 1e-4f * ((b1.hi[0] - b1.lo[0]) + (b1.hi[1] - b1.lo[1]) + (b1.hi[2] - b1.lo[2]));
        return me;
    }
};

}  // namespace

// Host counterpart of build_triangle_bvh for n <= kBvhHostBuildMax: `tris` and `pairs` are host memory.
void build_triangle_bvh_host(const DevTriTest* tris, uint32_t first, uint32_t n, DevBvhPair* pairs) {
    std::vector<HostBox> boxes(n);
    for (uint32_t i = 0; i < n; i++) {
        const DevTriTest& t = tris[first + i];
        for (int c = 0; c < 3; c++) {
            const float p0 = t.p0[c], p1 = t.p0[c] + t.e1[c], p2 = t.p0[c] + t.e2[c];
            const float lo = std::fmin(p0, std::fmin(p1, p2)), hi = std::fmax(p0, std::fmax(p1, p2));
            const float grow = 2.0f * 0.0001f * (std::fabs(t.e1[c]) + std::fabs(t.e2[c])) + 4e-6f * std::fmax(std::fabs(lo), std::fabs(hi)) + 1e-30f;
            boxes[i].lo[c] = lo - grow;
            boxes[i].hi[c] = hi + grow;
        }
    }
    HostBuilder b(boxes, first, pairs);
    (void)b.build(0, n, 0);
}

}  // namespace rtd
