// Residency census: which CU/SIMD does each wave run on, and when?  (HW_REG_HW_ID + s_memrealtime)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#include <algorithm>
#include "../../fourq_amd/csrc/curve.hip.h"
using namespace fq;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <int VG> __global__ __launch_bounds__(256) void k(uint64_t* out, int iters) {
    // keep VG VGPRs live: an array of accumulators updated in a loop
    uint32_t acc[VG];
#pragma unroll
    for (int i = 0; i < VG; i++) acc[i] = threadIdx.x * (i + 1);
    uint64_t r0, r1; uint32_t hwid, xcc;
    asm volatile("s_memrealtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(hwid), "=s"(xcc) :: "memory");
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < VG; i++) acc[i] = acc[i] * 1664525u + acc[(i + 1) % VG];
    }
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1) :: "memory");
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < VG; i++) s += acc[i];
    if (s == 0x1234567u) out[1 << 22] = s;
    if ((threadIdx.x & 63) == 0) { size_t w = (size_t)blockIdx.x * 4 + threadIdx.x / 64; out[4 * w] = r0; out[4 * w + 1] = r1; out[4 * w + 2] = hwid; out[4 * w + 3] = xcc; }
}
__global__ __launch_bounds__(256, 4) void kl(uint64_t* out, int iters) {
    __shared__ __attribute__((aligned(16))) u32 lds_table[8 * 52];
    for (int i = threadIdx.x; i < 8 * 52; i += 256) lds_table[i] = (123u * (i + 1)) & LIMB_MASK;
    __syncthreads();
    uint64_t digits = 123u * 0x9E3779B97F4A7C15ull + threadIdx.x * 0xD1B54A32D192ED03ull;
    Fe2<1> x, y;
    for (int i = 0; i < 5; i++) { x.re.l[i] = (77u * (i + 1) + threadIdx.x) & LIMB_MASK; x.im.l[i] = (77u * (i + 7) + threadIdx.x * 3) & LIMB_MASK;
                                  y.re.l[i] = (77u * (i + 3) ^ threadIdx.x) & LIMB_MASK; y.im.l[i] = (77u * (i + 11) + 5) & LIMB_MASK; }
    uint64_t r0, r1; uint32_t hwid, xcc;
    asm volatile("s_memrealtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(hwid), "=s"(xcc) :: "memory");
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        R1 Q; Q.X = x; Q.Y = y; Q.Z = fe2_carry(fe2_add(x, y));
        u32 dg = (u32)(digits >> (it & 31)) & 7, neg = (u32)((digits >> (32 + (it & 31))) & 1) - 1u;
        Q = dbl(Q.X, Q.Y, Q.Z);
        Q = add_table(Q, lds_table + dg * 52, neg);
        x = Q.X; y = fe2_carry(fe2_add(Q.Y, Q.Z));
    }
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1) :: "memory");
    uint32_t s = 0;
    for (int i = 0; i < 5; i++) s += x.re.l[i] + x.im.l[i] + y.re.l[i] + y.im.l[i];
    if (s == 0x1234567u) out[1 << 22] = s;
    if ((threadIdx.x & 63) == 0) { size_t w = (size_t)blockIdx.x * 4 + threadIdx.x / 64; out[4 * w] = r0; out[4 * w + 1] = r1; out[4 * w + 2] = hwid; out[4 * w + 3] = xcc; }
}
template <int VG> void run(uint64_t* d, int blocks) {
    if (VG == 0) { hipLaunchKernelGGL(kl, dim3(blocks), dim3(256), 0, 0, d, 64); CHECK(hipDeviceSynchronize()); hipLaunchKernelGGL(kl, dim3(blocks), dim3(256), 0, 0, d, 64); CHECK(hipDeviceSynchronize()); }
    else {
    hipLaunchKernelGGL(k<(VG ? VG : 1)>, dim3(blocks), dim3(256), 0, 0, d, 3000); CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k<(VG ? VG : 1)>, dim3(blocks), dim3(256), 0, 0, d, 3000); CHECK(hipDeviceSynchronize()); }
    std::vector<uint64_t> h(16 * blocks); CHECK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
    // key = (xcc, se, cu, simd)
    std::map<uint64_t, std::vector<std::pair<uint64_t,uint64_t>>> bysimd;
    uint64_t tmin = ~0ull, tmax = 0;
    for (int w = 0; w < blocks * 4; w++) {
        uint64_t r0 = h[4*w], r1 = h[4*w+1]; uint32_t id = (uint32_t)h[4*w+2], xcc = (uint32_t)h[4*w+3] & 0xf;
        uint32_t simd = (id >> 4) & 3, cu = (id >> 8) & 15, sh = (id >> 12) & 1, se = (id >> 13) & 7;
        uint64_t key = ((uint64_t)xcc << 24) | (se << 16) | (sh << 12) | (cu << 4) | simd;
        bysimd[key].push_back({r0, r1}); tmin = std::min(tmin, r0); tmax = std::max(tmax, r1);
    }
    // max concurrency per SIMD and distribution of waves per SIMD
    std::map<int,int> histo_waves, histo_conc;
    for (auto& kv : bysimd) {
        auto& v = kv.second; histo_waves[(int)v.size()]++;
        std::vector<std::pair<uint64_t,int>> ev; for (auto& p : v) { ev.push_back({p.first, +1}); ev.push_back({p.second, -1}); }
        std::sort(ev.begin(), ev.end()); int c = 0, m = 0; for (auto& e : ev) { c += e.second; m = std::max(m, c); }
        histo_conc[m]++;
    }
    double avg_life = 0; for (auto& kv : bysimd) for (auto& p : kv.second) avg_life += (p.second - p.first) / 100.0; avg_life /= (blocks * 4);
    printf("VG=%3d blocks=%5d: distinct SIMDs=%zu  span=%.1f us avg wave life %.1f us | waves per SIMD:", VG, blocks, bysimd.size(), (tmax - tmin) / 100.0, avg_life);
    for (auto& kv : histo_waves) printf(" %dx%d", kv.first, kv.second);
    printf(" | max concurrent per SIMD:");
    for (auto& kv : histo_conc) printf(" %dx%d", kv.first, kv.second);
    printf("\n");
    if (blocks == 512 || blocks == 1024) {
        int shown = 0;
        for (auto& kv : bysimd) { if (shown++ % 150) continue; printf("   simd %08llx:", (unsigned long long)kv.first); auto v = kv.second; std::sort(v.begin(), v.end());
            for (auto& p : v) printf(" [%.0f..%.0f]", (p.first - tmin) / 100.0, (p.second - tmin) / 100.0); printf("\n"); }
    }
}
int main() {
    uint64_t* d; CHECK(hipMalloc(&d, 8ull << 23));
    for (int blocks : {256, 512, 768, 1024, 2048}) { run<0>(d, blocks); }
    return 0;
}
