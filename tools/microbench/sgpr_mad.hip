// v_mad_u64_u32 with an SGPR multiplicand against an all-VGPR one (constants of the endomorphisms end up in SGPRs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITERS = 20000;
template <int MODE> __global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t seed) {
    uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 9;
    uint32_t b = (seed | 1) + threadIdx.x, c = seed ^ 0x55;
    uint32_t sc = __builtin_amdgcn_readfirstlane(seed ^ 0x1234567);
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) asm volatile("v_mad_u64_u32 %0, vcc, %5, %6, %0\n\tv_mad_u64_u32 %1, vcc, %5, %6, %1\n\tv_mad_u64_u32 %2, vcc, %5, %6, %2\n\tv_mad_u64_u32 %3, vcc, %5, %6, %3\n\tv_mad_u64_u32 %4, vcc, %5, %6, %4\n\t"
                                    "v_mad_u64_u32 %0, vcc, %6, %5, %0\n\tv_mad_u64_u32 %1, vcc, %6, %5, %1\n\tv_mad_u64_u32 %2, vcc, %6, %5, %2\n\tv_mad_u64_u32 %3, vcc, %6, %5, %3\n\tv_mad_u64_u32 %4, vcc, %6, %5, %4"
                                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(b), "v"(c) : "vcc");
        else asm volatile("v_mad_u64_u32 %0, vcc, %5, %6, %0\n\tv_mad_u64_u32 %1, vcc, %5, %6, %1\n\tv_mad_u64_u32 %2, vcc, %5, %6, %2\n\tv_mad_u64_u32 %3, vcc, %5, %6, %3\n\tv_mad_u64_u32 %4, vcc, %5, %6, %4\n\t"
                          "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\tv_mad_u64_u32 %1, vcc, %5, %6, %1\n\tv_mad_u64_u32 %2, vcc, %5, %6, %2\n\tv_mad_u64_u32 %3, vcc, %5, %6, %3\n\tv_mad_u64_u32 %4, vcc, %5, %6, %4"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(b), "s"(sc) : "vcc");
    }
    uint64_t sink = a0 + a1 + a2 + a3 + a4;
    if (sink == 0x123456789abcdefull) out[0] = sink;
}
int main() {
    uint64_t* d; CHECK(hipMalloc(&d, 4096));
    for (int warm = 0; warm < 20; warm++) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d, 123u);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[2] = {"mad, VGPR x VGPR", "mad, VGPR x SGPR"};
    for (int mode = 0; mode < 2; mode++) {
        printf("%-18s:", names[mode]);
        for (int w : {1, 2, 4}) {
            float ms = 0;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256 * w), dim3(256), 0, 0, d, 123u);
                else hipLaunchKernelGGL(k<1>, dim3(256 * w), dim3(256), 0, 0, d, 123u);
                CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("  W=%d %.2f ns/instr/SIMD", w, ms * 1e6 / ((double)ITERS * 10 * w));
        }
        printf("\n");
    }
    return 0;
}
