// Developer tool (tools/coresident_probe.py): a kernel that does nothing but hold workgroup slots — `blocks` workgroups of
// `threads` lanes with `lds_bytes` of LDS each, for `ms` milliseconds of wall time — the footprint a collective's
// long-lived kernel has beside a render.  Every wave leaves after the time given (s_memrealtime, 100 MHz): no memory is
// touched besides the workgroup's own LDS.
//   hipcc -O2 --offload-arch=gfx950 -shared -fPIC tools/occupier.hip -o tools/liboccupier.so
#include <hip/hip_runtime.h>

__global__ void occupy_kernel(unsigned long long ticks) {
    extern __shared__ unsigned char held[];
    held[threadIdx.x] = 1;                      // (the allocation is what matters; the store keeps it referenced)
    const unsigned long long begin = wall_clock64();
    while (wall_clock64() - begin < ticks) __builtin_amdgcn_s_sleep(64);
}

extern "C" int occupier_launch(int blocks, int threads, int lds_bytes, double ms, void* stream) {
    if (blocks <= 0 || threads <= 0 || threads > 1024 || lds_bytes < threads || lds_bytes > 160 * 1024 || ms < 0.0 || ms > 50.0) return -1;
    if (lds_bytes > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return -2;
    const unsigned long long ticks = static_cast<unsigned long long>(ms * 1e5);    // 100 MHz
    hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(threads), lds_bytes, static_cast<hipStream_t>(stream), ticks);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
