// Host-side sanitizer check of the BVH builder for short triangle runs (make asan-bvh): rt_debug_bvh_host of
// include/rt_hip.h — librt_hip's host code compiled with -Xarch_host -fsanitize=address,undefined — on random,
// flat, collinear and identical triangle sets, with the tree's invariants checked.  No GPU is touched.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../include/rt_hip.h"

static int check(const std::vector<float>& tris, uint32_t n, const char* what) {
    std::vector<uint32_t> rec(static_cast<size_t>(n - 1) * 16);
    if (rt_debug_bvh_host(tris.data(), n, rec.data()) != RT_OK) {
        std::printf("%s: rt_debug_bvh_host failed\n", what);
        return 1;
    }
    std::vector<int> seen(n, 0);
    for (uint32_t i = 0; i + 1 < n; i++) {
        for (int c = 0; c < 2; c++) {
            const uint32_t child = rec[i * 16 + (c == 0 ? 3 : 7)];
            if (child & 0x80000000u) {
                const uint32_t t = child & 0x7fffffffu;
                if (t >= n) return std::printf("%s: leaf %u out of range\n", what, t), 1;
                seen[t]++;
            } else if (child >= n - 1) {
                return std::printf("%s: child record %u out of range\n", what, child), 1;
            }
        }
    }
    for (uint32_t t = 0; t < n; t++)
        if (seen[t] != 1) return std::printf("%s: triangle %u in %d leaves\n", what, t, seen[t]), 1;
    return 0;
}

int main() {
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> u(-5.0f, 5.0f);
    int bad = 0;
    for (uint32_t n : {2u, 3u, 17u, 64u, 333u, 1024u}) {
        std::vector<float> t(static_cast<size_t>(n) * 9);
        for (float& v : t) v = u(rng);
        bad += check(t, n, "random");
        for (size_t i = 0; i < t.size(); i += 9) t[i + 1] = 0.0f, t[i + 4] = 0.0f, t[i + 7] = 0.0f;   // flat in y
        bad += check(t, n, "flat");
        for (size_t i = 0; i < t.size(); i += 9) std::memcpy(&t[i], &t[0], 9 * sizeof(float));           // all the same triangle
        bad += check(t, n, "identical");
    }
    std::printf("sanitize_bvh: %s\n", bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
