// How much does the ORDER of multiply-adds and cheap ops matter to a lone wave (one wave per SIMD: the regime of the
// fused variable-base kernels)?  Each pattern is a fixed sequence of v_mad_i64_i32 (M, independent accumulator chains) and
// 32-bit VOP2 ops / 64-bit shifts (a), issued from asm volatile so that hipcc cannot reorder it; the kernel runs one wave
// per SIMD over the whole chip and the wall time per instruction is printed, with s_memtime cycles beside it.
//   hipcc -O2 --offload-arch=gfx950 -o issue_order issue_order.hip && ./issue_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITERS = 20000, REP = 8;   // 8 copies of the pattern per loop trip: the loop's own scalar instructions stay below 3 %
#define BODY(text) { BODY1(text) BODY1(text) BODY1(text) BODY1(text) BODY1(text) BODY1(text) BODY1(text) BODY1(text) }

#define M(acc) "v_mad_i64_i32 %" #acc ", vcc, %8, %9, %" #acc "\n\t"
#define A(r)   "v_add_u32 %" #r ", %" #r ", %8\n\t"
#define S(acc) "v_ashrrev_i64 %" #acc ", 1, %" #acc "\n\t"
#define N0     "s_nop 0\n\t"
#define BODY1(text) asm volatile(text : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b), "v"(c) : "vcc");

template <int MODE> __global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t seed, uint64_t* cycles) {
    int64_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    uint32_t b = seed | 1, c = seed ^ 0x55, f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        // every pattern holds 12 instructions (s_nop not counted)
        if (MODE == 0) BODY(M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3))                 // 12 M, 4 chains
        if (MODE == 1) BODY(A(4) A(5) A(6) A(7) A(4) A(5) A(6) A(7) A(4) A(5) A(6) A(7))                 // 12 a
        if (MODE == 2) BODY(M(0) M(1) M(2) M(3) M(0) M(1) A(4) A(5) A(6) A(7) A(4) A(5))                 // 6 M then 6 a
        if (MODE == 3) BODY(M(0) A(4) M(1) A(5) M(2) A(6) M(3) A(7) M(0) A(4) M(1) A(5))                 // alternating
        if (MODE == 4) BODY(M(0) M(1) A(4) A(5) M(2) M(3) A(6) A(7) M(0) M(1) A(4) A(5))                 // pairs
        if (MODE == 5) BODY(M(0) M(1) M(2) A(4) M(3) M(0) M(1) A(5) M(2) M(3) M(0) A(6))                 // 9 M : 3 a, spread
        if (MODE == 6) BODY(M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) M(0) A(4) A(5) A(6))                 // 9 M : 3 a, grouped
        if (MODE == 7) BODY(M(0) M(1) A(4) M(2) M(3) A(5) M(0) A(6) M(1) M(2) A(7) M(3))                 // 8 M : 4 a (7 : 5 is the ladder's mix), spread
        if (MODE == 8) BODY(M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) A(4) A(5) A(6) A(7))                 // 8 M : 4 a, grouped
        if (MODE == 9) BODY(M(0) M(1) N0 M(0) M(1) N0 M(0) M(1) N0 M(0) M(1) N0 M(0) M(1) N0 M(0) M(1) N0)  // 2 chains + the s_nop hipcc inserts
        if (MODE == 10) BODY(M(0) M(1) M(2) M(0) M(1) M(2) M(0) M(1) M(2) M(0) M(1) M(2))                // 3 chains, no nop needed
        if (MODE == 11) BODY(M(0) S(1) M(2) S(3) M(0) S(1) M(2) S(3) M(0) S(1) M(2) S(3))                // M alternating with 64-bit shifts
        if (MODE == 12) BODY(M(0) A(4) N0 M(0) A(5) N0 M(0) A(6) N0 M(0) A(7) N0 M(0) A(4) N0 M(0) A(5) N0)    // ONE chain: a cheap op and an s_nop between dependent M
        if (MODE == 14) { BODY(A(4) A(5) A(6) A(7) A(4) A(5) A(6) A(7) A(4) A(5) A(6) A(7)) BODY(A(4) A(5) A(6) A(7) A(4) A(5) A(6) A(7) A(4) A(5) A(6) A(7)) }   // 16 copies per trip: isolates the loop's own cost
        if (MODE == 15) { BODY(M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3)) BODY(M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3)) }
        if (MODE == 13) BODY(M(0) A(4) A(5) M(0) A(6) A(7) M(0) A(4) A(5) M(0) A(6) A(7))                // one chain, two cheap ops between
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    uint64_t sink = a0 + a1 + a2 + a3 + f0 + f1 + f2 + f3;
    if (sink == 0x123456789abcdefull) out[0] = sink;
    if (threadIdx.x % 64 == 0) { uint32_t w = blockIdx.x * 4 + threadIdx.x / 64; cycles[2 * w] = t1 - t0; cycles[2 * w + 1] = r1 - r0; }
}

template <int MODE> static void run(const char* name, uint64_t* d, uint64_t* cyc) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, d, 123u, cyc);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    static uint64_t h[2048]; CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    uint64_t cmin = ~0ull, cmax = 0, rmin = ~0ull, rmax = 0;
    for (int w = 0; w < 1024; w++) { cmin = h[2*w] < cmin ? h[2*w] : cmin; cmax = h[2*w] > cmax ? h[2*w] : cmax; rmin = h[2*w+1] < rmin ? h[2*w+1] : rmin; rmax = h[2*w+1] > rmax ? h[2*w+1] : rmax; }
    const double n = (double)ITERS * REP * 12;
    printf("%-48s %.3f ns/instr  s_memtime/instr min %.2f max %.2f  realtime(100 MHz) min %.3f max %.3f ms  -> clock %.2f..%.2f GHz  (kernel %.3f ms)\n", name, ms * 1e6 / n,
           cmin / n, cmax / n, rmin * 1e-5, rmax * 1e-5, cmin / (rmin * 10.0) , cmax / (rmax * 10.0), ms);
}
int main() {
    uint64_t *d, *cyc; CHECK(hipMalloc(&d, 4096)); CHECK(hipMalloc(&cyc, 2048 * 8));
    for (int pass = 0; pass < 2; pass++) {
        run<0>("12 M (4 chains)", d, cyc);
        run<1>("12 a", d, cyc);
        run<2>("6 M then 6 a", d, cyc);
        run<3>("M a alternating", d, cyc);
        run<4>("MM aa pairs", d, cyc);
        run<5>("9 M : 3 a spread (MMMa)", d, cyc);
        run<6>("9 M : 3 a grouped", d, cyc);
        run<7>("8 M : 4 a spread (MMa)", d, cyc);
        run<8>("8 M : 4 a grouped", d, cyc);
        run<9>("2 chains with s_nop 0 after each pair (12 M)", d, cyc);
        run<10>("3 chains (12 M)", d, cyc);
        run<11>("M alternating with v_ashrrev_i64", d, cyc);
        run<12>("one chain: M a nop M a nop", d, cyc);
        run<13>("one chain: M a a M a a", d, cyc);
        run<14>("12 a, 16 copies per trip (figures are per 2 instr)", d, cyc);
        run<15>("12 M, 16 copies per trip (figures are per 2 instr)", d, cyc);
        printf("\n");
    }
    return 0;
}
