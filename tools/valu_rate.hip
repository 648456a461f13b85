// tools/valu_rate.hip — measures what one gfx950 SIMD issues per clock for the instruction kinds the
// trace kernels are made of, at 1..8 waves per SIMD.  Developer tool; the numbers price the "issue
// ceiling" quoted in DESIGN.md and bench.py.
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate tools/valu_rate.hip && tools/valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                     \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

typedef float F2 __attribute__((ext_vector_type(2)));

constexpr int kUnroll = 16;   // independent chains per lane
constexpr int kPattern = 66;  // trips of the inner loop for the ordering experiments (a multiple of 22 and 3 ... see KIND 38-41)
constexpr int kIters = 4096;

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float* out, float seed) {
    float a[kUnroll];
    F2 p[kUnroll];
    double d[kUnroll];
    for (int i = 0; i < kUnroll; i++) {
        a[i] = seed + static_cast<float>(threadIdx.x + i);
        p[i] = {a[i], a[i] * 0.5f};
        d[i] = a[i];
    }
    const float m = 1.0000001f, c = 1e-9f;
    const F2 pm = {m, m}, pc = {c, c};
    const double dm = 1.0000001, dc = 1e-9;
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int i = 0; i < (KIND >= 38 && KIND != 40 ? kPattern : kUnroll); i++) {
            if (KIND == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
            if (KIND == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
            if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
            if (KIND == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 7) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 8) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dm));
            if (KIND == 9) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dm), "v"(dc));
            if (KIND == 10) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc));
            if (KIND == 11) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(m) : "vcc");
            if (KIND == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
            if (KIND == 13) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 14) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 15) asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(a[i]) : "v"(m), "v"(c));   // dependent pair
            if (KIND == 16) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(m) : "vcc");
            if (KIND == 17) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c) : "vcc");
            if (KIND == 18) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (KIND == 19) asm volatile("s_mul_i32 s20, s20, s21" : : : "s20");
            if (KIND == 20) asm volatile("v_mul_f32 %0, %0, s20" : "+v"(a[i]) : : "s20");
            if (KIND == 21) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 22) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[i]));
            if (KIND == 23) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
            if (KIND == 24) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(seed));             // SGPR operand, no clobber
            if (KIND == 25) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(a[i]));                           // inline constant
            if (KIND == 26) asm volatile("v_mul_f32 %0, 0x3f800001, %0" : "+v"(a[i]));                    // 32-bit literal
            if (KIND == 27) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "s"(seed));
            if (KIND == 28) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(seed), "v"(c));
            if (KIND == 29) asm volatile("v_max_f32 %0, %0, %1\n\tv_min_f32 %0, %0, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (KIND == 30) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "s"(seed));
            if (KIND == 31) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
            if (KIND == 32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 33) asm volatile("v_mul_f32 %0, %0, %1\n\tv_mul_f32 %0, %2, %0" : "+v"(a[i]) : "v"(m), "s"(seed));   // alternating VGPR / SGPR
            // ---- latency: ONE dependent chain per lane (what a lone wave pays per instruction of a chain)
            if (KIND == 42) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[0]) : "v"(pc));
            if (KIND == 43) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[0]) : "v"(pm));
            if (KIND == 44) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(c));
            if (KIND == 45) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[0]) : "v"(dm), "v"(dc));
            if (KIND == 46) { if ((i & 1) == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[0]) : "v"(pc)); else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[1]) : "v"(pc)); }   // two chains
            // ---- ordering experiments: the same mix of "slow" (SGPR operand / compare / max) and "fast" instructions,
            // grouped (4 slow then 4 fast) or alternating, every instruction on its own chain
            if (KIND == 34) { if ((i & 7) < 4) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(seed)); else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m)); }
            if (KIND == 35) { if ((i & 1) == 0) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(seed)); else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m)); }
            if (KIND == 36) { if ((i & 7) < 4) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m)); else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m)); }
            if (KIND == 37) { if ((i & 1) == 0) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m)); else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m)); }
            // the triangle stage's shape as the compiler emits it: s s f  s s f  s s f  s s s s f s s f f f f f f (12 slow, 10 fast) ...
            if (KIND == 38) { const bool slow = (0xdedb >> (i % 22)) & 1; if (slow) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i % 16]) : "s"(seed)); else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i % 16]) : "v"(m)); }
            // ... and the same 12 + 10 alternating as far as the mix allows: s f s f ... s s
            if (KIND == 39) { const bool slow = (i % 22) >= 20 || ((i % 22) & 1) == 0; if (slow) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i % 16]) : "s"(seed)); else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i % 16]) : "v"(m)); }
            if (KIND == 40) { if ((i & 1) == 0) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(seed)); else asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m)); }   // slow + slow of two kinds
            if (KIND == 41) { if ((i % 3) == 0) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i % 16]) : "s"(seed)); else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i % 16]) : "v"(m)); }   // s f f
        }
    }
    float s = 0.0f;
    for (int i = 0; i < kUnroll; i++) s += a[i] + p[i].x + p[i].y + static_cast<float>(d[i]);
    if (s == 12345.678f) out[0] = s;
}

template <int KIND>
void run(const char* name, float* d_out, int n_cus, double ops_per_instr) {
    printf("%-38s", name);
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = n_cus * wps;   // 256-thread blocks: 4 waves = one per SIMD
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 3;
        const double instr_per_simd = static_cast<double>(kIters) * (KIND >= 38 && KIND != 40 ? kPattern : kUnroll) * ((KIND == 15 || KIND == 29 || KIND == 31 || KIND == 33) ? 2 : 1) * wps;
        const double cyc = ms * 1e-3 * 2.4e9 / instr_per_simd;   // cycles at 2.4 GHz per wave-instruction per SIMD
        printf("  w/simd=%d: %6.2f cyc/instr", wps, cyc);
        (void)ops_per_instr;
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cus = prop.multiProcessorCount;
    printf("%s, %d CUs, clock %d kHz; cycles are at a nominal 2.4 GHz\n", prop.name, n_cus, prop.clockRate);
    float* d_out;
    CHECK(hipMalloc(&d_out, 256));
    run<0>("v_mul_f32", d_out, n_cus, 1);
    run<1>("v_add_f32", d_out, n_cus, 1);
    run<2>("v_fma_f32", d_out, n_cus, 2);
    run<3>("v_pk_mul_f32", d_out, n_cus, 2);
    run<4>("v_pk_add_f32", d_out, n_cus, 2);
    run<5>("v_pk_fma_f32", d_out, n_cus, 4);
    run<6>("v_rcp_f32", d_out, n_cus, 1);
    run<7>("v_sqrt_f32", d_out, n_cus, 1);
    run<8>("v_mul_f64", d_out, n_cus, 1);
    run<9>("v_fma_f64", d_out, n_cus, 2);
    run<10>("v_add_f64", d_out, n_cus, 1);
    run<11>("v_cmp_gt_f32", d_out, n_cus, 1);
    run<12>("v_cndmask_b32", d_out, n_cus, 1);
    run<13>("v_max_f32", d_out, n_cus, 1);
    run<14>("v_mov_b32", d_out, n_cus, 1);
    run<15>("dependent mul,add pair", d_out, n_cus, 1);
    run<16>("v_div_scale_f32", d_out, n_cus, 1);
    run<17>("v_div_fmas_f32", d_out, n_cus, 1);
    run<18>("v_div_fixup_f32", d_out, n_cus, 1);
    run<19>("s_mul_i32", d_out, n_cus, 1);
    run<20>("v_mul_f32 with sgpr", d_out, n_cus, 1);
    run<21>("v_exp_f32", d_out, n_cus, 1);
    run<22>("v_rsq_f64", d_out, n_cus, 1);
    run<23>("v_cvt_f64_f32", d_out, n_cus, 1);
    run<24>("v_mul_f32 sgpr (arg)", d_out, n_cus, 1);
    run<25>("v_add_f32 inline const", d_out, n_cus, 1);
    run<26>("v_mul_f32 literal", d_out, n_cus, 1);
    run<27>("v_sub_f32 sgpr", d_out, n_cus, 1);
    run<28>("v_fma_f32 sgpr", d_out, n_cus, 1);
    run<29>("v_max,v_min pair", d_out, n_cus, 1);
    run<30>("v_mov_b32 from sgpr", d_out, n_cus, 1);
    run<31>("v_cmp + v_cndmask", d_out, n_cus, 1);
    run<32>("v_xor_b32", d_out, n_cus, 1);
    run<33>("v_mul vgpr, v_mul sgpr", d_out, n_cus, 1);
    run<42>("one chain of v_pk_add_f32", d_out, n_cus, 1);
    run<43>("one chain of v_pk_mul_f32", d_out, n_cus, 1);
    run<44>("one chain of v_add_f32", d_out, n_cus, 1);
    run<45>("one chain of v_fma_f64", d_out, n_cus, 1);
    run<46>("two chains of v_pk_add_f32", d_out, n_cus, 1);
    run<34>("4 sgpr-mul, 4 mul grouped", d_out, n_cus, 1);
    run<35>("sgpr-mul, mul alternating", d_out, n_cus, 1);
    run<36>("4 max, 4 mul grouped", d_out, n_cus, 1);
    run<37>("max, mul alternating", d_out, n_cus, 1);
    run<38>("12 sgpr + 10 plain, compiler's order", d_out, n_cus, 1);
    run<39>("12 sgpr + 10 plain, alternating", d_out, n_cus, 1);
    run<40>("sgpr-mul, max alternating", d_out, n_cus, 1);
    run<41>("sgpr-mul, mul, mul", d_out, n_cus, 1);
    return 0;
}
