// Full-chip check: does the v_mad_u64_u32 / v_add_u32 issue rate measured on one CU hold when all
// 256 CUs run, and what clock does the chip hold?  (s_memtime = shader cycles, s_memrealtime = 100 MHz.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITERS = 20000;
template <int MIX> __global__ void k(uint64_t* out, uint32_t seed) {
    uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 9;
    uint32_t b = seed | 1, c = seed ^ 0x55, e0 = seed * 3, f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    uint64_t t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    for (int it = 0; it < ITERS; ++it) {
        asm volatile("v_mad_u64_u32 %0, vcc, %5, %6, %0\n\tv_mad_u64_u32 %1, vcc, %5, %7, %1\n\tv_mad_u64_u32 %2, vcc, %6, %7, %2\n\tv_mad_u64_u32 %3, vcc, %5, %5, %3\n\tv_mad_u64_u32 %4, vcc, %6, %6, %4"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(b), "v"(c), "v"(e0) : "vcc");
        if (MIX) asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_and_b32 %2, %2, %4\n\tv_lshrrev_b32 %3, 1, %3\n\tv_add_u32 %0, %0, %1"
                              : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b));
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    uint64_t sink = a0 + a1 + a2 + a3 + a4 + f0 + f1 + f2 + f3;
    if (sink == 0x123456789abcdefull) out[1 << 20] = sink;
    if ((threadIdx.x & 63) == 0) { size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64; out[2 * w] = t1 - t0; out[2 * w + 1] = r1 - r0; }
}
int main() {
    uint64_t* d; CHECK(hipMalloc(&d, 8ull << 21));
    for (int mix = 0; mix < 2; mix++)
        for (int blocks : {1, 256, 512, 1024})
            for (int wps : {1}) {
                int threads = 256;
                std::vector<uint64_t> h(2 * blocks * 4);
                for (int rep = 0; rep < 3; rep++) {
                    if (mix) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, d, 123u); else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, d, 123u);
                    CHECK(hipDeviceSynchronize());
                }
                CHECK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
                std::vector<double> cyc, mhz;
                for (size_t w = 0; w < h.size() / 2; w++) { cyc.push_back((double)h[2 * w]); mhz.push_back(h[2 * w] / (h[2 * w + 1] / 100.0)); }
                std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
                int per_iter = mix ? 10 : 5;
                printf("mix=%d blocks=%4d (x256 thr): median %.2f cyc/instr/wave, clock median %.0f MHz (min %.0f max %.0f)\n", mix, blocks,
                       cyc[cyc.size() / 2] / ((double)ITERS * per_iter), mhz[mhz.size() / 2], mhz.front(), mhz.back());
            }
    return 0;
}
