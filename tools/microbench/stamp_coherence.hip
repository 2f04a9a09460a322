// Are s_memtime / s_memrealtime the SAME counters for every wave of an XCD (and across XCDs)?  If so, the clock a stretch of work ran at
// can be had from two tiny stamp kernels on the work's own stream, one before and one after it, with no probe resident beside it (a
// resident probe wave displaces a workgroup of a kernel that fills the register file: cfg3 ran 30 % slower under one, profiles/r06_clock_probe.txt).
// Kernel S: every block (one wave) records {xcc_id, memtime, memrealtime}.  Launch S, then `work` busy kernels, then S again; per XCD the
// spread of (memtime - memrealtime * f) over the blocks of one launch says whether the counters are shared; the per-XCD delta gives the clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <cmath>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ __launch_bounds__(64) void stamp(uint64_t* out) {
    uint64_t t, r;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "=s"(r) :: "memory");
    uint32_t xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);      // HW_REG_XCC_ID, bits 3:0
    uint32_t hwid = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);     // HW_REG_HW_ID
    if (threadIdx.x == 0) { out[4 * blockIdx.x] = xcc & 15; out[4 * blockIdx.x + 1] = t; out[4 * blockIdx.x + 2] = r; out[4 * blockIdx.x + 3] = hwid; }
}
__global__ void busy(uint64_t* out, int iters) {
    uint64_t a = threadIdx.x + 1, b = blockIdx.x + 3;
    for (int i = 0; i < iters; i++) { a = a * b + i; b = b * a + 7; }
    if (a == 0x1234567 && b == 99) out[0] = a;
}
// which hardware unit owns a memtime counter?  many blocks, grouped by ever coarser keys of HW_ID: the finest key whose groups are coherent
// (spread of memtime - 24 * memrealtime within a few thousand cycles) names the counter's owner
static void owner_test() {
    constexpr int B = 4096;
    uint64_t* d; CHECK(hipMalloc(&d, B * 32));
    hipLaunchKernelGGL(stamp, dim3(B), dim3(64), 0, 0, d);
    CHECK(hipDeviceSynchronize());
    std::vector<uint64_t> h(4 * B);
    CHECK(hipMemcpy(h.data(), d, B * 32, hipMemcpyDeviceToHost));
    struct Key { const char* name; uint32_t mask; };
    // HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
    const Key keys[] = { {"xcc", 0}, {"xcc+se", 0xE000}, {"xcc+se+sh", 0xF000}, {"xcc+se+sh+cu", 0xFF00}, {"xcc+se+sh+cu+simd", 0xFF30} };
    for (const Key& k : keys) {
        std::vector<std::pair<uint64_t, double>> v;
        for (int b = 0; b < B; b++) {
            uint64_t key = (h[4 * b] << 32) | (h[4 * b + 3] & k.mask);
            v.push_back({key, (double)(int64_t)(h[4 * b + 1] - h[1]) - 24.0 * (double)(int64_t)(h[4 * b + 2] - h[2])});
        }
        std::sort(v.begin(), v.end());
        size_t groups = 0, coherent = 0; double worst = 0;
        for (size_t i = 0; i < v.size();) {
            size_t j = i; double lo = v[i].second, hi = v[i].second;
            while (j < v.size() && v[j].first == v[i].first) { lo = std::min(lo, v[j].second); hi = std::max(hi, v[j].second); j++; }
            groups++; if (hi - lo < 20000) coherent++; worst = std::max(worst, hi - lo);
            i = j;
        }
        printf("owner test, key %-18s: %4zu groups, %4zu coherent (spread < 20 000 cycles), worst spread %.0f\n", k.name, groups, coherent, worst);
    }
}
int main() {
    owner_test();
    constexpr int B = 64;
    uint64_t *d0, *d1, *sink;
    CHECK(hipMalloc(&d0, B * 32)); CHECK(hipMalloc(&d1, B * 32)); CHECK(hipMalloc(&sink, 64));
    hipStream_t s; CHECK(hipStreamCreate(&s));
    for (int rep = 0; rep < 4; rep++) {
        const int launches = rep == 0 ? 0 : 20 * rep;
        hipLaunchKernelGGL(stamp, dim3(B), dim3(64), 0, s, d0);
        for (int i = 0; i < launches; i++) hipLaunchKernelGGL(busy, dim3(4096), dim3(256), 0, s, sink, 20000);
        hipLaunchKernelGGL(stamp, dim3(B), dim3(64), 0, s, d1);
        CHECK(hipStreamSynchronize(s));
        std::vector<uint64_t> h0(4 * B), h1(4 * B);
        CHECK(hipMemcpy(h0.data(), d0, B * 32, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h1.data(), d1, B * 32, hipMemcpyDeviceToHost));
        printf("rep %d: %d busy launches between the stamps\n", rep, launches);
        for (int x = 0; x < 8; x++) {
            // within one launch: spread of memtime among this XCD's blocks after removing the realtime skew at ~24 cycles per tick
            double lo = 1e30, hi = -1e30; int cnt = 0; uint64_t t0 = 0, r0 = 0, t1 = 0, r1 = 0;
            for (int b = 0; b < B; b++) if ((int)h0[4 * b] == x) {
                if (!cnt) { t0 = h0[4 * b + 1]; r0 = h0[4 * b + 2]; }
                double v = (double)(int64_t)(h0[4 * b + 1] - t0) - 24.0 * (double)(int64_t)(h0[4 * b + 2] - r0);
                lo = std::min(lo, v); hi = std::max(hi, v); cnt++;
            }
            int cnt1 = 0;
            for (int b = 0; b < B; b++) if ((int)h1[4 * b] == x && !cnt1++) { t1 = h1[4 * b + 1]; r1 = h1[4 * b + 2]; }
            if (!cnt || !cnt1) { printf("  xcc %d: %d / %d blocks\n", x, cnt, cnt1); continue; }
            printf("  xcc %d: %2d blocks, in-launch memtime spread (skew removed) %.0f cycles; realtime spread in launch: first block r=%llu; delta %.3f ms -> %.1f MHz\n",
                   x, cnt, hi - lo, (unsigned long long)r0, (double)(r1 - r0) / 1e5, (double)(t1 - t0) / ((double)(r1 - r0) / 100.0));
        }
        // cross-XCD: are memrealtime values comparable?  min / max of r over the first launch
        uint64_t rmin = ~0ull, rmax = 0, tmin = ~0ull, tmax = 0;
        for (int b = 0; b < B; b++) { rmin = std::min(rmin, h0[4 * b + 2]); rmax = std::max(rmax, h0[4 * b + 2]); tmin = std::min(tmin, h0[4 * b + 1]); tmax = std::max(tmax, h0[4 * b + 1]); }
        printf("  all blocks of the first stamp launch: memrealtime spread %llu ticks (10 ns), memtime spread %llu cycles\n", (unsigned long long)(rmax - rmin), (unsigned long long)(tmax - tmin));
        printf("  blockIdx -> xcc of the first 16 blocks:"); for (int b = 0; b < 16; b++) printf(" %d", (int)h0[4 * b]); printf("\n");
    }
    return 0;
}
