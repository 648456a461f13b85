// Unsigned 32-bit division by a divisor that is fixed for a launch (Granlund & Montgomery, "Division by invariant
// integers using multiplication", the round-up form): the host computes (m, sh1, sh2) once, the device divides with one
// high multiply, one subtraction, one addition and two shifts instead of the ~25 (32-bit) or ~150 (64-bit) instructions of
// a division by a run-time value.  Exact for every n < 2^32 and every d in 1 .. 2^32 - 1 (tests/test_divide.py).
// Plain C++: the same header serves hipcc (device and host) and g++ (the test).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define RT_DIVIDE_FN __host__ __device__ __forceinline__
#else
#define RT_DIVIDE_FN inline
#endif

namespace rtd {

struct DevDiv {
    uint32_t m, sh1, sh2, d;
};

// d >= 1
inline DevDiv make_div(uint32_t d) {
    uint32_t l = 0;
    while (l < 32 && (uint64_t(1) << l) < d) l++;      // ceil(log2 d)
    const uint64_t m = ((uint64_t(1) << 32) * ((uint64_t(1) << l) - d)) / d + 1;
    return DevDiv{static_cast<uint32_t>(m), l < 1 ? l : 1u, l > 0 ? l - 1 : 0u, d};
}

RT_DIVIDE_FN uint32_t div_u32(uint32_t n, const DevDiv& v) {
    const uint32_t t = static_cast<uint32_t>((static_cast<uint64_t>(v.m) * n) >> 32);
    return (t + ((n - t) >> v.sh1)) >> v.sh2;
}
// quotient and remainder
RT_DIVIDE_FN uint32_t divmod_u32(uint32_t n, const DevDiv& v, uint32_t& rem) {
    const uint32_t q = div_u32(n, v);
    rem = n - q * v.d;
    return q;
}

}  // namespace rtd
