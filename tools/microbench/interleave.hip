// Lone-wave issue cost of a mad/simple-op mix as a function of how finely the two kinds alternate.
// Every kernel executes the same 20 multiply-adds (5 independent chains) and 20 simple ops per iteration;
// only the block size of the alternation differs.  W = waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITERS = 10000;
#define M(i) "v_mad_u64_u32 %" #i ", vcc, %10, %11, %" #i "\n\t"
#define S(i) "v_add_u32 %" #i ", %" #i ", %10\n\t"
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4) : "v"(b), "v"(c) : "vcc"
template <int BLK> __global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t seed) {
    uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 9;
    uint32_t b = seed | 1, c = seed ^ 0x55, f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4;
    for (int it = 0; it < ITERS; ++it) {
        if (BLK == 1) {
            asm volatile(M(0) S(5) M(1) S(6) M(2) S(7) M(3) S(8) M(4) S(9) M(0) S(5) M(1) S(6) M(2) S(7) M(3) S(8) M(4) S(9)
                         M(0) S(5) M(1) S(6) M(2) S(7) M(3) S(8) M(4) S(9) M(0) S(5) M(1) S(6) M(2) S(7) M(3) S(8) M(4) S(9) OPS);
        } else if (BLK == 2) {
            asm volatile(M(0) M(1) S(5) S(6) M(2) M(3) S(7) S(8) M(4) M(0) S(9) S(5) M(1) M(2) S(6) S(7) M(3) M(4) S(8) S(9)
                         M(0) M(1) S(5) S(6) M(2) M(3) S(7) S(8) M(4) M(0) S(9) S(5) M(1) M(2) S(6) S(7) M(3) M(4) S(8) S(9) OPS);
        } else if (BLK == 5) {
            asm volatile(M(0) M(1) M(2) M(3) M(4) S(5) S(6) S(7) S(8) S(9) M(0) M(1) M(2) M(3) M(4) S(5) S(6) S(7) S(8) S(9)
                         M(0) M(1) M(2) M(3) M(4) S(5) S(6) S(7) S(8) S(9) M(0) M(1) M(2) M(3) M(4) S(5) S(6) S(7) S(8) S(9) OPS);
        } else if (BLK == 10) {
            asm volatile(M(0) M(1) M(2) M(3) M(4) M(0) M(1) M(2) M(3) M(4) S(5) S(6) S(7) S(8) S(9) S(5) S(6) S(7) S(8) S(9)
                         M(0) M(1) M(2) M(3) M(4) M(0) M(1) M(2) M(3) M(4) S(5) S(6) S(7) S(8) S(9) S(5) S(6) S(7) S(8) S(9) OPS);
        } else if (BLK == 20) {
            asm volatile(M(0) M(1) M(2) M(3) M(4) M(0) M(1) M(2) M(3) M(4) M(0) M(1) M(2) M(3) M(4) M(0) M(1) M(2) M(3) M(4)
                         S(5) S(6) S(7) S(8) S(9) S(5) S(6) S(7) S(8) S(9) S(5) S(6) S(7) S(8) S(9) S(5) S(6) S(7) S(8) S(9) OPS);
        } else if (BLK == 100) {   // 2 mads : 1 simple, finely alternated (30 instrs)
            asm volatile(M(0) M(1) S(5) M(2) M(3) S(6) M(4) M(0) S(7) M(1) M(2) S(8) M(3) M(4) S(9)
                         M(0) M(1) S(5) M(2) M(3) S(6) M(4) M(0) S(7) M(1) M(2) S(8) M(3) M(4) S(9) OPS);
        } else {                   // BLK == 101: 2 chains only (dependent every other mad), alternated 1:1
            asm volatile(M(0) S(5) M(1) S(6) M(0) S(7) M(1) S(8) M(0) S(9) M(1) S(5) M(0) S(6) M(1) S(7) M(0) S(8) M(1) S(9)
                         M(0) S(5) M(1) S(6) M(0) S(7) M(1) S(8) M(0) S(9) M(1) S(5) M(0) S(6) M(1) S(7) M(0) S(8) M(1) S(9) OPS);
        }
    }
    uint64_t sink = a0 + a1 + a2 + a3 + a4 + f0 + f1 + f2 + f3 + f4;
    if (sink == 0x123456789abcdefull) out[0] = sink;
}
template <int BLK> void run(uint64_t* d, const char* name, int instrs) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-44s:", name);
    for (int w : {1, 2, 4}) {
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k<BLK>, dim3(256 * w), dim3(256), 0, 0, d, 123u);
            CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("  W=%d %.2f ns/instr/SIMD", w, ms * 1e6 / ((double)ITERS * instrs * w));
    }
    printf("\n");
}
int main() {
    uint64_t* d; CHECK(hipMalloc(&d, 4096));
    run<1>(d, "20 mad + 20 add, alternating singly", 40);
    run<2>(d, "20 mad + 20 add, blocks of 2", 40);
    run<5>(d, "20 mad + 20 add, blocks of 5", 40);
    run<10>(d, "20 mad + 20 add, blocks of 10", 40);
    run<20>(d, "20 mad + 20 add, blocks of 20", 40);
    run<100>(d, "20 mad + 10 add, pattern MMS", 30);
    run<101>(d, "20 mad (2 chains) + 20 add, alternating singly", 40);
    return 0;
}
