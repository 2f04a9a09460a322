// True per-SIMD VALU throughput (kernel wall time, all CUs, W waves per SIMD): because the SIMD serves the
// oldest wave first, per-wave timers overstate multi-wave throughput; kernel time does not.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITERS = 20000;
// MODE 0: 10 mads (5 chains x2) ; 1: 10 v_add_u32 (5 chains) ; 2: 5 mads + 5 adds ; 3: 10 lshl_add_u64 ; 4: 10 v_and/v_lshrrev mix; 5: 5 mad + 5 lshrrev_b64
template <int MODE> __global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t seed) {
    uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 9;
    uint32_t b = seed | 1, c = seed ^ 0x55, e0 = seed * 3, f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4;
    for (int it = 0; it < ITERS; ++it) {
#define MAD5 asm volatile("v_mad_u64_u32 %0, vcc, %5, %6, %0\n\tv_mad_u64_u32 %1, vcc, %5, %7, %1\n\tv_mad_u64_u32 %2, vcc, %6, %7, %2\n\tv_mad_u64_u32 %3, vcc, %5, %5, %3\n\tv_mad_u64_u32 %4, vcc, %6, %6, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(b), "v"(c), "v"(e0) : "vcc");
#define ADD5 asm volatile("v_add_u32 %0, %0, %5\n\tv_add_u32 %1, %1, %5\n\tv_add_u32 %2, %2, %5\n\tv_add_u32 %3, %3, %5\n\tv_add_u32 %4, %4, %5" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4) : "v"(b));
#define LSA5 asm volatile("v_lshl_add_u64 %0, %0, 0, %5\n\tv_lshl_add_u64 %1, %1, 0, %5\n\tv_lshl_add_u64 %2, %2, 0, %5\n\tv_lshl_add_u64 %3, %3, 0, %5\n\tv_lshl_add_u64 %4, %4, 0, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(a0));
#define BIT5 asm volatile("v_and_b32 %0, %0, %5\n\tv_lshrrev_b32 %1, 1, %1\n\tv_xor_b32 %2, %2, %5\n\tv_lshlrev_b32 %3, 1, %3\n\tv_sub_u32 %4, %4, %5" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4) : "v"(b));
#define SHR5 asm volatile("v_lshrrev_b64 %0, 1, %0\n\tv_lshrrev_b64 %1, 1, %1\n\tv_lshrrev_b64 %2, 1, %2\n\tv_lshrrev_b64 %3, 1, %3\n\tv_lshrrev_b64 %4, 1, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4));
        if (MODE == 0) { MAD5 MAD5 } else if (MODE == 1) { ADD5 ADD5 } else if (MODE == 2) { MAD5 ADD5 } else if (MODE == 3) { LSA5 LSA5 } else if (MODE == 4) { BIT5 BIT5 } else { MAD5 SHR5 }
    }
    uint64_t sink = a0 + a1 + a2 + a3 + a4 + f0 + f1 + f2 + f3 + f4;
    if (sink == 0x123456789abcdefull) out[0] = sink;
}
int main() {
    uint64_t* d; CHECK(hipMalloc(&d, 4096));
    const char* names[6] = {"v_mad_u64_u32 x10", "v_add_u32 x10", "5 mad + 5 add", "v_lshl_add_u64 x10", "and/shr/xor/shl/sub x10", "5 mad + 5 lshrrev_b64"};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 6; mode++) {
        printf("%-26s:", names[mode]);
        for (int w : {1, 2, 4, 8}) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(e0));
                switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(256 * w), dim3(256), 0, 0, d, 123u); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(256 * w), dim3(256), 0, 0, d, 123u); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(256 * w), dim3(256), 0, 0, d, 123u); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(256 * w), dim3(256), 0, 0, d, 123u); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(256 * w), dim3(256), 0, 0, d, 123u); break;
                default: hipLaunchKernelGGL(k<5>, dim3(256 * w), dim3(256), 0, 0, d, 123u); break;
                }
                CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            double ns_per_instr_per_simd = ms * 1e6 / ((double)ITERS * 10 * w);
            printf("  W=%d %.2f ns/instr/SIMD", w, ns_per_instr_per_simd);
        }
        printf("   (x clock GHz = cycles)\n");
    }
    return 0;
}
