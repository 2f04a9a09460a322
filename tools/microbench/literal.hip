// Does a 32-bit literal operand (8-byte encoding) change the issue cost of simple VOP2 ops on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITERS = 20000;
template <int MODE> __global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t seed) {
    uint32_t f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, m = seed | 0x3ffffff;
    uint32_t sm = __builtin_amdgcn_readfirstlane(seed | 0x3ffffff);
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) asm volatile("v_and_b32 %0, 0x3ffffff, %0\n\tv_and_b32 %1, 0x3ffffff, %1\n\tv_and_b32 %2, 0x3ffffff, %2\n\tv_and_b32 %3, 0x3ffffff, %3\n\tv_and_b32 %4, 0x3ffffff, %4\n\t"
                                    "v_add_u32 %0, 0x7fffffe, %0\n\tv_add_u32 %1, 0x7fffffe, %1\n\tv_add_u32 %2, 0x7fffffe, %2\n\tv_add_u32 %3, 0x7fffffe, %3\n\tv_add_u32 %4, 0x7fffffe, %4"
                                    : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4));
        else if (MODE == 1) asm volatile("v_and_b32 %0, %5, %0\n\tv_and_b32 %1, %5, %1\n\tv_and_b32 %2, %5, %2\n\tv_and_b32 %3, %5, %3\n\tv_and_b32 %4, %5, %4\n\t"
                                    "v_add_u32 %0, %5, %0\n\tv_add_u32 %1, %5, %1\n\tv_add_u32 %2, %5, %2\n\tv_add_u32 %3, %5, %3\n\tv_add_u32 %4, %5, %4"
                                    : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4) : "v"(m));
        else asm volatile("v_and_b32 %0, %5, %0\n\tv_and_b32 %1, %5, %1\n\tv_and_b32 %2, %5, %2\n\tv_and_b32 %3, %5, %3\n\tv_and_b32 %4, %5, %4\n\t"
                                    "v_add_u32 %0, %5, %0\n\tv_add_u32 %1, %5, %1\n\tv_add_u32 %2, %5, %2\n\tv_add_u32 %3, %5, %3\n\tv_add_u32 %4, %5, %4"
                                    : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4) : "s"(sm));
    }
    if (f0 + f1 + f2 + f3 + f4 == 0x12345678u) out[0] = f0;
}
int main() {
    uint64_t* d; CHECK(hipMalloc(&d, 4096));
    const char* names[3] = {"literal operand", "VGPR operand", "SGPR operand"};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; mode++) {
        printf("%-18s:", names[mode]);
        for (int w : {1, 2, 4}) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256 * w), dim3(256), 0, 0, d, 123u);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256 * w), dim3(256), 0, 0, d, 123u);
                else hipLaunchKernelGGL(k<2>, dim3(256 * w), dim3(256), 0, 0, d, 123u);
                CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("  W=%d %.2f ns/instr/SIMD", w, ms * 1e6 / ((double)ITERS * 10 * w));
        }
        printf("\n");
    }
    return 0;
}
