// Does gfx950 spend less issue time on a wave64 VALU instruction when only the low 32 lanes are active?
// If it did, 2^16 elements could run as 2048 half-filled waves (2 per SIMD) instead of 1024 full ones (1 per SIMD)
// and gain the multi-wave issue rates.  Kernel wall time, every SIMD busy:  hipcc -O2 --offload-arch=gfx950 -o halfwave halfwave.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4000;
template <int KIND> __global__ __launch_bounds__(64) void k(uint64_t* out, uint32_t seed, int active) {
    const int lane = threadIdx.x;
    if (lane >= active) return;                     // EXEC = low `active` lanes from here on
    uint32_t a0 = seed + lane, a1 = seed * 3 + lane, a2 = seed * 5 + lane, a3 = seed * 7 + lane, b = seed | 1;
    uint64_t c0 = lane, c1 = lane + 1, c2 = lane + 2, c3 = lane + 3, c4 = lane + 4;
    uint32_t d0 = lane, d1 = lane * 3, d2 = lane * 5, d3 = lane * 7, d4 = lane * 9;
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
        if (KIND == 0) {                            // 10 multiply-adds, 5 independent chains
#pragma unroll
            for (int r = 0; r < 2; r++) {
                c0 += (uint64_t)a0 * b; c1 += (uint64_t)a1 * b; c2 += (uint64_t)a2 * b; c3 += (uint64_t)a3 * b; c4 += (uint64_t)a0 * a1;
                asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4));
            }
        } else if (KIND == 1) {                     // 10 plain 32-bit adds
#pragma unroll
            for (int r = 0; r < 2; r++) {
                d0 += a0; d1 += a1; d2 += a2; d3 += a3; d4 += b;
                asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4));
            }
        } else {                                    // the path's mix: 5 multiply-adds + 5 simple ops
            c0 += (uint64_t)a0 * b; c1 += (uint64_t)a1 * b; c2 += (uint64_t)a2 * b; c3 += (uint64_t)a3 * b; c4 += (uint64_t)a0 * a1;
            d0 += a0; d1 &= a1; d2 += a2; d3 ^= a3; d4 += b;
            asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4));
        }
    }
    uint64_t s = c0 + c1 + c2 + c3 + c4 + d0 + d1 + d2 + d3 + d4;
    if (s == 0x123456789abcdefull) out[0] = s;
}
int main() {
    uint64_t* d; CHECK(hipMalloc(&d, 4096));
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int simds = p.multiProcessorCount * 4;
    const char* names[3] = { "10 x v_mad_u64_u32", "10 x v_add_u32", "5 mad + 5 simple" };
    for (int kind = 0; kind < 3; kind++)
        for (int active : { 64, 32, 16 })
            for (int w : { 1, 2, 4 }) {
                hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); float ms = 0, best = 1e9;
                for (int rep = 0; rep < 4; rep++) {
                    CHECK(hipEventRecord(e0));
                    if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(simds * w), dim3(64), 0, 0, d, 12345u, active);
                    else if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(simds * w), dim3(64), 0, 0, d, 12345u, active);
                    else hipLaunchKernelGGL(k<2>, dim3(simds * w), dim3(64), 0, 0, d, 12345u, active);
                    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep && ms < best) best = ms;
                }
                printf("%-20s active lanes %2d  waves/SIMD %d : %7.3f ms -> %5.2f ns per wave-instruction per SIMD, %6.3f ns per ACTIVE-LANE-instruction x 64\n",
                       names[kind], active, w, best, best * 1e6 / (10.0 * ITERS * w), best * 1e6 / (10.0 * ITERS * w) * 64 / active);
            }
    return 0;
}
