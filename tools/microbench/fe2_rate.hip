// Cycles per GF(p^2) multiplication / squaring of fp127.hip.h as compiled, by waves per SIMD, all CUs busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include "../../fourq_amd/csrc/curve.hip.h"
using namespace fq;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITERS = 200;
template <int MODE> __global__ __launch_bounds__(256, MODE >= 3 ? 4 : 1) void k(uint64_t* out, uint32_t seed) {
    __shared__ __attribute__((aligned(16))) u32 lds_table[8 * 52];
    if (MODE >= 4) { for (int i = threadIdx.x; i < 8 * 52; i += 256) lds_table[i] = (seed * (i + 1)) & LIMB_MASK; __syncthreads(); }
    uint64_t digits = seed * 0x9E3779B97F4A7C15ull + threadIdx.x * 0xD1B54A32D192ED03ull;
    Fe2<1> x, y;
    for (int i = 0; i < 5; i++) { x.re.l[i] = (seed * (i + 1) + threadIdx.x) & LIMB_MASK; x.im.l[i] = (seed * (i + 7) + threadIdx.x * 3) & LIMB_MASK;
                                  y.re.l[i] = (seed * (i + 3) ^ threadIdx.x) & LIMB_MASK; y.im.l[i] = (seed * (i + 11) + 5) & LIMB_MASK; }
    uint64_t t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) { x = fe2_mul(x, y); y = fe2_mul(y, x); x = fe2_mul(x, y); y = fe2_mul(y, x); }
        else if (MODE == 1) { x = fe2_sqr(x); y = fe2_sqr(y); x = fe2_sqr(x); y = fe2_sqr(y); }
        else if (MODE == 3) {   // one ladder step: DBL + ADD with a register-resident table entry
            R1 Q; Q.X = x; Q.Y = y; Q.Z = fe2_carry(fe2_add(x, y));
            Q = dbl(Q.X, Q.Y, Q.Z);
            R2s t; t.N = x; t.D = y; t.E = Q.Z; t.F = widen<2>(Q.X);
            Q = add(Q, t);
            x = Q.X; y = fe2_carry(fe2_add(Q.Y, Q.Z));
        }
        else if (MODE == 4 || MODE == 5) {   // ladder step with the table entry gathered from LDS (4) / from a global table (5)
            R1 Q; Q.X = x; Q.Y = y; Q.Z = fe2_carry(fe2_add(x, y));
            u32 dg = (u32)(digits >> (it & 31)) & 7, neg = (u32)((digits >> (32 + (it & 31))) & 1) - 1u;
            Q = dbl(Q.X, Q.Y, Q.Z);
            Q = (MODE == 4) ? add_table(Q, lds_table + dg * 52, neg) : add_table(Q, (const u32*)(out + 4096) + ((blockIdx.x * 256 + threadIdx.x) * 8 + dg) * 48, neg);
            x = Q.X; y = fe2_carry(fe2_add(Q.Y, Q.Z));
        }
        else { Fe2<1> a = fe2_mul(x, y), b = fe2_sqr(x), c = fe2_sqr(y), d = fe2_mul(y, fe2_add(x, y)); x = fe2_carry(fe2_sub(fe2_add(a, b), c)); y = fe2_carry(fe2_add(d, a)); }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    uint32_t sink = 0;
    for (int i = 0; i < 5; i++) sink += x.re.l[i] + x.im.l[i] + y.re.l[i] + y.im.l[i];
    if (sink == 0x12345678u) out[1 << 20] = sink;
    if ((threadIdx.x & 63) == 0) { size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64; out[2 * w] = t1 - t0; out[2 * w + 1] = r1 - r0; }
}
int main() {
    uint64_t* d; CHECK(hipMalloc(&d, 1200ull << 20)); CHECK(hipMemset(d, 1, 1200ull << 20));
    const char* names[6] = {"4 x fe2_mul (dependent)", "4 x fe2_sqr (2 chains)", "2 mul + 2 sqr + mul (mixed)", "DBL + ADD (one ladder step)", "DBL + add_table(LDS)", "DBL + add_table(global)"};
    for (int mode : {0, 4})
        for (int blocks : {256, 512, 768, 1024, 1280, 1536, 2048, 4096}) {
            std::vector<uint64_t> h(2 * blocks * 4);
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, 123u);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, 123u);
                else if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, 123u);
                else if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, d, 123u);
                else if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, 123u);
                else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, 123u);
                CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            CHECK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> cyc, mhz;
            for (size_t w = 0; w < h.size() / 2; w++) { cyc.push_back((double)h[2 * w]); mhz.push_back(h[2 * w] / (h[2 * w + 1] / 100.0)); }
            std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
            double per_iter = cyc[cyc.size() / 2] / ITERS;
            int wps = blocks / 256;
            printf("%-26s blocks=%4d (%.1f/CU): %8.0f cycles/iter/wave ; kernel %.3f ms -> %.1f ns per wave-iter per SIMD ; clock %.0f MHz\n", names[mode], blocks, blocks / 256.0, per_iter, ms, ms * 1e6 / ((double)ITERS * blocks * 4 / 1024.0), mhz[mhz.size() / 2]);
            (void)wps;
        }
    return 0;
}
