// Does VALU throughput at high occupancy depend on the size of the loop body (instruction cache / fetch)?
// Same work (dependent fe2_mul chain, 67 VGPRs) with the loop body unrolled U times.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include "../../fourq_amd/csrc/fp127.hip.h"
using namespace fq;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int TOTAL = 1024;   // fe2_mul pairs per wave
template <int U> __global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t seed) {
    Fe2<1> x, y;
    for (int i = 0; i < 5; i++) { x.re.l[i] = (seed * (i + 1) + threadIdx.x) & LIMB_MASK; x.im.l[i] = (seed * (i + 7) + threadIdx.x * 3) & LIMB_MASK;
                                  y.re.l[i] = (seed * (i + 3) ^ threadIdx.x) & LIMB_MASK; y.im.l[i] = (seed * (i + 11) + 5) & LIMB_MASK; }
    uint64_t t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll 1
    for (int it = 0; it < TOTAL / U; ++it) {
#pragma unroll
        for (int u = 0; u < U; u++) { x = fe2_mul(x, y); y = fe2_mul(y, x); }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    uint32_t sink = 0;
    for (int i = 0; i < 5; i++) sink += x.re.l[i] + x.im.l[i] + y.re.l[i] + y.im.l[i];
    if (sink == 0x12345678u) out[1 << 20] = sink;
    if ((threadIdx.x & 63) == 0) { size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64; out[w] = t1 - t0; }
}
template <int U> void run(uint64_t* d) {
    for (int blocks : {256, 1024, 1792}) {
        std::vector<uint64_t> h(blocks * 4);
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k<U>, dim3(blocks), dim3(256), 0, 0, d, 123u); CHECK(hipDeviceSynchronize()); }
        CHECK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        double per = (double)h[h.size() / 2] / (2 * TOTAL);
        printf("unroll %3d (~%5.1f KB body) waves/SIMD=%d: %7.1f cycles/fe2_mul/wave -> %6.1f /SIMD\n", U, U * 2 * 155 * 7.6 / 1024, blocks / 256, per, per / (blocks / 256));
    }
}
int main() {
    uint64_t* d; CHECK(hipMalloc(&d, 8ull << 21));
    run<1>(d); run<4>(d); run<8>(d); run<16>(d); run<32>(d); run<64>(d);
    return 0;
}
