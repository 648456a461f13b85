// Checks raytracer_amd/csrc/rt_divide.h against the hardware division: every divisor 1..4096 and a spread of larger ones,
// against a dense set of numerators (all below 2^16, every power of two +-2, multiples of the divisor +-1, 2^32 - 1 downwards,
// and pseudo-random ones).  Prints "bad=0" when all agree.  (tests/test_divide.py)
#include <cstdint>
#include <cstdio>
#include <vector>
#include "../raytracer_amd/csrc/rt_divide.h"

int main() {
    std::vector<uint32_t> ds;
    for (uint32_t d = 1; d <= 4096; d++) ds.push_back(d);
    for (uint32_t k = 12; k < 32; k++)
        for (int o = -3; o <= 3; o++) ds.push_back((1u << k) + o);
    ds.push_back(0xffffffffu);
    ds.push_back(0xfffffffeu);
    uint32_t x = 12345;
    for (int i = 0; i < 2000; i++) {
        x = x * 1664525u + 1013904223u;
        ds.push_back(x | 1u);
    }
    unsigned long long bad = 0, checked = 0;
    for (uint32_t d : ds) {
        const rtd::DevDiv v = rtd::make_div(d);
        auto check = [&](uint32_t n) {
            uint32_t r;
            const uint32_t q = rtd::divmod_u32(n, v, r);
            checked++;
            if (q != n / d || r != n % d) {
                if (bad < 10) std::printf("n=%u d=%u: got %u rem %u, want %u rem %u\n", n, d, q, r, n / d, n % d);
                bad++;
            }
        };
        for (uint32_t n = 0; n < 65536; n += (d <= 64 ? 1 : 7)) check(n);
        for (uint32_t k = 0; k < 32; k++)
            for (int o = -2; o <= 2; o++) check((1u << k) + o);
        for (uint64_t mlt = 0; mlt <= 0xffffffffull; mlt += (0xffffffffull / 997 / d + 1) * d) {
            check(static_cast<uint32_t>(mlt));
            check(static_cast<uint32_t>(mlt) - 1u);
            check(static_cast<uint32_t>(mlt) + 1u);
        }
        for (uint32_t n = 0xffffffffu; n > 0xffffffffu - 64; n--) check(n);
        uint32_t y = d * 2654435761u;
        for (int i = 0; i < 512; i++) {
            y = y * 1664525u + 1013904223u;
            check(y);
        }
    }
    std::printf("divisors %zu, checks %llu -> bad=%llu\n", ds.size(), checked, bad);
    return bad != 0;
}
