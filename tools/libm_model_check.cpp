// Checks raytracer_amd/csrc/rt_libm.h (glibc 2.35's float asinf/atanf/atan2f restated for the device)
// against the host libm, compiled for the host:
//   g++ -O2 -ffp-contract=off -o libm_model_check tools/libm_model_check.cpp -lpthread
//   ./libm_model_check [stride] [pairs_per_thread]      stride 1 = every float (about 3 CPU-minutes)
// Prints "bad=0" when every compared result has the same bit pattern (NaNs compare as NaNs).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../raytracer_amd/csrc/rt_libm.h"

namespace {
struct Tally {
    long asin_bad = 0, atan_bad = 0, atan2_bad = 0, compared = 0;
};
bool same(float a, float b) { return rtd::libm_bits(a) == rtd::libm_bits(b) || (a != a && b != b); }

void sweep(unsigned id, unsigned n_threads, uint64_t stride, long pairs, Tally* out) {
    Tally t;
    for (uint64_t u = static_cast<uint64_t>(id) * stride; u < 0x100000000ull; u += stride * n_threads) {
        const float x = rtd::libm_float(static_cast<uint32_t>(u));
        if (!same(atanf(x), rtd::glibc_atanf(x))) {
            if (t.atan_bad++ < 3) std::printf("atanf(%a): %a vs %a\n", x, atanf(x), rtd::glibc_atanf(x));
        }
        if (!same(asinf(x), rtd::glibc_asinf(x))) {
            if (t.asin_bad++ < 3) std::printf("asinf(%a): %a vs %a\n", x, asinf(x), rtd::glibc_asinf(x));
        }
        t.compared += 2;
    }
    uint64_t s = 88172645463325252ull + id * 7919ull;
    const float special[] = {0.0f, -0.0f, 1.0f, -1.0f, INFINITY, -INFINITY, NAN, 1e-40f, -1e-40f, 3e38f, -3e38f, 1e-30f, 0.5f};
    const int n_special = sizeof(special) / sizeof(special[0]);
    for (long i = 0; i < pairs; ++i) {
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        float y, x;
        switch (i & 3) {
            case 0: y = rtd::libm_float(static_cast<uint32_t>(s)), x = rtd::libm_float(static_cast<uint32_t>(s >> 32)); break;
            case 1: y = special[(s >> 8) % n_special], x = rtd::libm_float(static_cast<uint32_t>(s >> 32)); break;
            case 2: y = rtd::libm_float(static_cast<uint32_t>(s)), x = special[(s >> 40) % n_special]; break;
            default:   // components of unit vectors, what Sphere::intersect passes
                y = static_cast<float>(static_cast<int32_t>(s)) / 2147483648.0f, x = static_cast<float>(static_cast<int32_t>(s >> 32)) / 2147483648.0f;
        }
        if (!same(atan2f(y, x), rtd::glibc_atan2f(y, x))) {
            if (t.atan2_bad++ < 3) std::printf("atan2f(%a, %a): %a vs %a\n", y, x, atan2f(y, x), rtd::glibc_atan2f(y, x));
        }
        ++t.compared;
    }
    for (int a = 0; a < n_special; ++a) {
        for (int b = 0; b < n_special; ++b) {
            if (!same(atan2f(special[a], special[b]), rtd::glibc_atan2f(special[a], special[b]))) ++t.atan2_bad;
        }
    }
    *out = t;
}
}  // namespace

int main(int argc, char** argv) {
    const uint64_t stride = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1;
    const long pairs = argc > 2 ? std::strtol(argv[2], nullptr, 10) : 200000000L;
    const unsigned n_threads = 8;
    std::vector<Tally> tallies(n_threads);
    std::vector<std::thread> threads;
    for (unsigned i = 0; i < n_threads; ++i) threads.emplace_back(sweep, i, n_threads, stride, pairs, &tallies[i]);
    Tally sum;
    for (unsigned i = 0; i < n_threads; ++i) {
        threads[i].join();
        sum.asin_bad += tallies[i].asin_bad, sum.atan_bad += tallies[i].atan_bad, sum.atan2_bad += tallies[i].atan2_bad, sum.compared += tallies[i].compared;
    }
    std::printf("asinf bad=%ld atanf bad=%ld atan2f bad=%ld compared=%ld -> bad=%ld\n", sum.asin_bad, sum.atan_bad, sum.atan2_bad, sum.compared,
                sum.asin_bad + sum.atan_bad + sum.atan2_bad);
    return sum.asin_bad + sum.atan_bad + sum.atan2_bad ? 1 : 0;
}
