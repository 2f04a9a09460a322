// Cost of straight-line code for a lone wave as a function of its footprint: each kernel executes the same number
// of instructions (a 1:1 mix of multiply-adds and additions, 8-byte and 4-byte encodings) as a loop over a body of
// KB kilobytes of code; small bodies are served by the instruction cache, large ones stream through it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
// one group = 5 mads (8 B each) + 5 adds (4 B each) = 60 bytes, 10 instructions
#define GROUP "v_mad_u64_u32 %0, vcc, %10, %11, %0\n\tv_add_u32 %5, %5, %10\n\tv_mad_u64_u32 %1, vcc, %10, %11, %1\n\tv_add_u32 %6, %6, %10\n\t" \
              "v_mad_u64_u32 %2, vcc, %10, %11, %2\n\tv_add_u32 %7, %7, %10\n\tv_mad_u64_u32 %3, vcc, %10, %11, %3\n\tv_add_u32 %8, %8, %10\n\t" \
              "v_mad_u64_u32 %4, vcc, %10, %11, %4\n\tv_add_u32 %9, %9, %10\n\t"
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4) : "v"(b), "v"(c) : "vcc"
template <int GROUPS> __global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t seed, int reps) {
    uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 9;
    uint32_t b = seed | 1, c = seed ^ 0x55, f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4;
    for (int it = 0; it < reps; ++it) {
        if (GROUPS == 16) asm volatile(".rept 16\n\t" GROUP ".endr" OPS);
        else if (GROUPS == 128) asm volatile(".rept 128\n\t" GROUP ".endr" OPS);
        else if (GROUPS == 512) asm volatile(".rept 512\n\t" GROUP ".endr" OPS);
        else if (GROUPS == 768) asm volatile(".rept 768\n\t" GROUP ".endr" OPS);
        else if (GROUPS == 1024) asm volatile(".rept 1024\n\t" GROUP ".endr" OPS);
        else if (GROUPS == 1536) asm volatile(".rept 1536\n\t" GROUP ".endr" OPS);
        else asm volatile(".rept 2048\n\t" GROUP ".endr" OPS);
    }
    uint64_t sink = a0 + a1 + a2 + a3 + a4 + f0 + f1 + f2 + f3 + f4;
    if (sink == 0x123456789abcdefull) out[0] = sink;
}
template <int GROUPS> void run(uint64_t* d) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int total_groups = 4096 * 8;                     // same instruction count for every footprint
    const int reps = total_groups / GROUPS;
    printf("body %4d KB x %5d passes:", GROUPS * 60 / 1024, reps);
    for (int w : {1, 2, 4}) {
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k<GROUPS>, dim3(256 * w), dim3(256), 0, 0, d, 123u, reps);
            CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("  W=%d %.2f ns/instr/SIMD", w, ms * 1e6 / ((double)total_groups * 10 * w));
    }
    printf("\n");
}
int main() {
    uint64_t* d; CHECK(hipMalloc(&d, 4096));
    for (int warm = 0; warm < 30; warm++) hipLaunchKernelGGL(k<128>, dim3(256), dim3(256), 0, 0, d, 123u, 256);   // clocks up
    CHECK(hipDeviceSynchronize());
    run<16>(d); run<128>(d); run<512>(d); run<768>(d); run<1024>(d); run<1536>(d); run<2048>(d);
    // a single pass over a large body = entirely cold code
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k<2048>, dim3(256), dim3(256), 0, 0, d, 123u, 1);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("body  120 KB x     1 pass (cold), W=1: %.2f ns/instr/SIMD (launch overhead included: %.1f us total)\n", ms * 1e6 / (2048.0 * 10), ms * 1e3);
    return 0;
}
