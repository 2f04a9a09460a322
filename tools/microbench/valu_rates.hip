// Instruction-rate microbenchmark for the integer/FP64 VALU ops the FourQ field layer can be
// built from (gfx950).  One workgroup on one CU; W waves per SIMD; every wave runs the same
// unrolled block of independent (or dependent) instructions and stamps s_memtime around it.
// Prints shader cycles per wave-instruction as seen by ONE wave and per SIMD (aggregate).
//
//   hipcc -O2 --offload-arch=gfx950 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 2000;
constexpr int UNROLL = 16;   // instructions per loop body

// Each op: 4 independent chains (ILP=4) or 1 dependent chain (ILP=1) of the same instruction.
#define DEFINE_KERNEL(NAME, DECL, INIT, BODY_I4, BODY_D1, SINK)                                   \
    template <int ILP> __global__ void k_##NAME(uint64_t* out, uint32_t seed) {                    \
        DECL; INIT;                                                                                 \
        uint64_t t0, t1;                                                                            \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory"); \
        for (int it = 0; it < ITERS; ++it) {                                                        \
            if (ILP == 4) { BODY_I4 BODY_I4 BODY_I4 BODY_I4 } else { BODY_D1 BODY_D1 BODY_D1 BODY_D1 } \
        }                                                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");               \
        uint64_t sink = SINK;                                                                       \
        if (sink == 0x123456789abcdefull) out[4096] = sink;                                          \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
    }

// ---- 32-bit integer
#define I32_DECL uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, b = seed | 1, c = seed ^ 0x55
#define OP3(INS, X) asm volatile(INS " %0, %0, %1" : "+v"(X) : "v"(b));
DEFINE_KERNEL(add_u32, I32_DECL, , OP3("v_add_u32", a0) OP3("v_add_u32", a1) OP3("v_add_u32", a2) OP3("v_add_u32", a3),
              OP3("v_add_u32", a0) OP3("v_add_u32", a0) OP3("v_add_u32", a0) OP3("v_add_u32", a0), a0 + a1 + a2 + a3)
DEFINE_KERNEL(mul_lo_u32, I32_DECL, , OP3("v_mul_lo_u32", a0) OP3("v_mul_lo_u32", a1) OP3("v_mul_lo_u32", a2) OP3("v_mul_lo_u32", a3),
              OP3("v_mul_lo_u32", a0) OP3("v_mul_lo_u32", a0) OP3("v_mul_lo_u32", a0) OP3("v_mul_lo_u32", a0), a0 + a1 + a2 + a3)
DEFINE_KERNEL(mul_hi_u32, I32_DECL, , OP3("v_mul_hi_u32", a0) OP3("v_mul_hi_u32", a1) OP3("v_mul_hi_u32", a2) OP3("v_mul_hi_u32", a3),
              OP3("v_mul_hi_u32", a0) OP3("v_mul_hi_u32", a0) OP3("v_mul_hi_u32", a0) OP3("v_mul_hi_u32", a0), a0 + a1 + a2 + a3)
DEFINE_KERNEL(mul_u32_u24, I32_DECL, , OP3("v_mul_u32_u24", a0) OP3("v_mul_u32_u24", a1) OP3("v_mul_u32_u24", a2) OP3("v_mul_u32_u24", a3),
              OP3("v_mul_u32_u24", a0) OP3("v_mul_u32_u24", a0) OP3("v_mul_u32_u24", a0) OP3("v_mul_u32_u24", a0), a0 + a1 + a2 + a3)
DEFINE_KERNEL(mul_hi_u32_u24, I32_DECL, , OP3("v_mul_hi_u32_u24", a0) OP3("v_mul_hi_u32_u24", a1) OP3("v_mul_hi_u32_u24", a2) OP3("v_mul_hi_u32_u24", a3),
              OP3("v_mul_hi_u32_u24", a0) OP3("v_mul_hi_u32_u24", a0) OP3("v_mul_hi_u32_u24", a0) OP3("v_mul_hi_u32_u24", a0), a0 + a1 + a2 + a3)
#define OP4(INS, X) asm volatile(INS " %0, %1, %2, %0" : "+v"(X) : "v"(b), "v"(c));
DEFINE_KERNEL(mad_u32_u24, I32_DECL, , OP4("v_mad_u32_u24", a0) OP4("v_mad_u32_u24", a1) OP4("v_mad_u32_u24", a2) OP4("v_mad_u32_u24", a3),
              OP4("v_mad_u32_u24", a0) OP4("v_mad_u32_u24", a0) OP4("v_mad_u32_u24", a0) OP4("v_mad_u32_u24", a0), a0 + a1 + a2 + a3)
DEFINE_KERNEL(alignbit, I32_DECL, , OP4("v_alignbit_b32", a0) OP4("v_alignbit_b32", a1) OP4("v_alignbit_b32", a2) OP4("v_alignbit_b32", a3),
              OP4("v_alignbit_b32", a0) OP4("v_alignbit_b32", a0) OP4("v_alignbit_b32", a0) OP4("v_alignbit_b32", a0), a0 + a1 + a2 + a3)
DEFINE_KERNEL(and_or, I32_DECL, , OP4("v_and_or_b32", a0) OP4("v_and_or_b32", a1) OP4("v_and_or_b32", a2) OP4("v_and_or_b32", a3),
              OP4("v_and_or_b32", a0) OP4("v_and_or_b32", a0) OP4("v_and_or_b32", a0) OP4("v_and_or_b32", a0), a0 + a1 + a2 + a3)
DEFINE_KERNEL(add3_u32, I32_DECL, , OP4("v_add3_u32", a0) OP4("v_add3_u32", a1) OP4("v_add3_u32", a2) OP4("v_add3_u32", a3),
              OP4("v_add3_u32", a0) OP4("v_add3_u32", a0) OP4("v_add3_u32", a0) OP4("v_add3_u32", a0), a0 + a1 + a2 + a3)
#define MOV(X, Y) asm volatile("v_mov_b32 %0, %1" : "=v"(X) : "v"(Y));
DEFINE_KERNEL(mov_b32, I32_DECL, , MOV(a0, a1) MOV(a1, a2) MOV(a2, a3) MOV(a3, b),
              MOV(a0, a0) MOV(a0, a0) MOV(a0, a0) MOV(a0, a0), a0 + a1 + a2 + a3)
// carry chain: v_add_co_u32 + v_addc_co_u32 pairs (vcc dependent)
#define ADDC(X) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(X) : "v"(b) : "vcc");
DEFINE_KERNEL(addco_addc_pair, I32_DECL, , ADDC(a0) ADDC(a1), ADDC(a0) ADDC(a0), a0 + a1 + a2 + a3)
#define CND(X) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(X) : "v"(b) : );
DEFINE_KERNEL(cndmask, I32_DECL, , CND(a0) CND(a1) CND(a2) CND(a3), CND(a0) CND(a0) CND(a0) CND(a0), a0 + a1 + a2 + a3)

// ---- 64-bit integer
#define I64_DECL uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3; uint32_t b = seed | 1, c = seed ^ 0x55; uint64_t d = a0 ^ 0x1234
#define MAD64(X) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(X) : "v"(b), "v"(c) : "vcc");
DEFINE_KERNEL(mad_u64_u32, I64_DECL, , MAD64(a0) MAD64(a1) MAD64(a2) MAD64(a3), MAD64(a0) MAD64(a0) MAD64(a0) MAD64(a0), a0 + a1 + a2 + a3)
#define LSHLADD64(X) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(X) : "v"(d));
DEFINE_KERNEL(lshl_add_u64, I64_DECL, , LSHLADD64(a0) LSHLADD64(a1) LSHLADD64(a2) LSHLADD64(a3),
              LSHLADD64(a0) LSHLADD64(a0) LSHLADD64(a0) LSHLADD64(a0), a0 + a1 + a2 + a3)
#define LSHR64(X) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(X));
DEFINE_KERNEL(lshrrev_b64, I64_DECL, , LSHR64(a0) LSHR64(a1) LSHR64(a2) LSHR64(a3), LSHR64(a0) LSHR64(a0) LSHR64(a0) LSHR64(a0), a0 + a1 + a2 + a3)
// mad_u64 followed by the mov needed to feed its high half as the next 32-bit carry
#define MADMOV(X, Y) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %1" : "=v"(X) : "v"(Y), "v"(b), "v"(c) : "vcc");
DEFINE_KERNEL(mad_u64_u32_srcdst_differ, I64_DECL, , MADMOV(a0, a1) MADMOV(a1, a2) MADMOV(a2, a3) MADMOV(a3, d),
              MADMOV(a0, a0) MADMOV(a0, a0) MADMOV(a0, a0) MADMOV(a0, a0), a0 + a1 + a2 + a3)

// ---- FP64
#define F64_DECL double a0 = 1.0 + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b = 1.0000001, c = 1e-9
#define FMA64(X) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(X) : "v"(b), "v"(c));
DEFINE_KERNEL(fma_f64, F64_DECL, , FMA64(a0) FMA64(a1) FMA64(a2) FMA64(a3), FMA64(a0) FMA64(a0) FMA64(a0) FMA64(a0),
              (uint64_t)__double_as_longlong(a0 + a1 + a2 + a3))
#define ADD64F(X) asm volatile("v_add_f64 %0, %0, %1" : "+v"(X) : "v"(c));
DEFINE_KERNEL(add_f64, F64_DECL, , ADD64F(a0) ADD64F(a1) ADD64F(a2) ADD64F(a3), ADD64F(a0) ADD64F(a0) ADD64F(a0) ADD64F(a0),
              (uint64_t)__double_as_longlong(a0 + a1 + a2 + a3))
#define MUL64F(X) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(X) : "v"(b));
DEFINE_KERNEL(mul_f64, F64_DECL, , MUL64F(a0) MUL64F(a1) MUL64F(a2) MUL64F(a3), MUL64F(a0) MUL64F(a0) MUL64F(a0) MUL64F(a0),
              (uint64_t)__double_as_longlong(a0 + a1 + a2 + a3))
// ---- FP32 reference point
#define F32_DECL float a0 = 1.0f + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b = 1.0000001f, c = 1e-9f
#define FMA32(X) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(X) : "v"(b), "v"(c));
DEFINE_KERNEL(fma_f32, F32_DECL, , FMA32(a0) FMA32(a1) FMA32(a2) FMA32(a3), FMA32(a0) FMA32(a0) FMA32(a0) FMA32(a0),
              (uint64_t)__float_as_uint(a0 + a1 + a2 + a3))
// ---- mixed: the shape of a schoolbook inner step: 1 mad + 2 movs + 1 64-bit add
#define MIXSTEP(X, Y) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %1\n\tv_mov_b32 %4, %5\n\tv_mov_b32 %5, %4" \
    : "=v"(X) : "v"(Y), "v"(b), "v"(c), "v"(e0), "v"(e1) : "vcc");
DEFINE_KERNEL(mix_mad_2mov, I64_DECL; uint32_t e0 = seed; uint32_t e1 = seed + 1, , MIXSTEP(a0, a1) MIXSTEP(a1, a2) MIXSTEP(a2, a3) MIXSTEP(a3, d),
              MIXSTEP(a0, a0) MIXSTEP(a0, a0) MIXSTEP(a0, a0) MIXSTEP(a0, a0), a0 + a1 + a2 + a3 + e0 + e1)

// v_bfi_b32 (mask select), VOP3 cndmask with an SGPR-pair condition, sub with borrow
DEFINE_KERNEL(bfi_b32, I32_DECL, , OP4("v_bfi_b32", a0) OP4("v_bfi_b32", a1) OP4("v_bfi_b32", a2) OP4("v_bfi_b32", a3),
              OP4("v_bfi_b32", a0) OP4("v_bfi_b32", a0) OP4("v_bfi_b32", a0) OP4("v_bfi_b32", a0), a0 + a1 + a2 + a3)
#define CND3(X) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(X) : "v"(b), "s"(cm));
DEFINE_KERNEL(cndmask_e64_sgpr, I32_DECL; uint64_t cm = __ballot(threadIdx.x & 1), , CND3(a0) CND3(a1) CND3(a2) CND3(a3), CND3(a0) CND3(a0) CND3(a0) CND3(a0), a0 + a1 + a2 + a3)
#define CNDV(X) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(X) : "v"(b));
DEFINE_KERNEL(cndmask_vcc_set, I32_DECL; asm volatile("v_cmp_gt_u32 vcc, %0, %1" :: "v"(a0), "v"(b) : "vcc"), , CNDV(a0) CNDV(a1) CNDV(a2) CNDV(a3), CNDV(a0) CNDV(a0) CNDV(a0) CNDV(a0), a0 + a1 + a2 + a3)
DEFINE_KERNEL(xor_b32, I32_DECL, , OP3("v_xor_b32", a0) OP3("v_xor_b32", a1) OP3("v_xor_b32", a2) OP3("v_xor_b32", a3),
              OP3("v_xor_b32", a0) OP3("v_xor_b32", a0) OP3("v_xor_b32", a0) OP3("v_xor_b32", a0), a0 + a1 + a2 + a3)
DEFINE_KERNEL(lshrrev_b32, I32_DECL, , OP3("v_lshrrev_b32", a0) OP3("v_lshrrev_b32", a1) OP3("v_lshrrev_b32", a2) OP3("v_lshrrev_b32", a3),
              OP3("v_lshrrev_b32", a0) OP3("v_lshrrev_b32", a0) OP3("v_lshrrev_b32", a0) OP3("v_lshrrev_b32", a0), a0 + a1 + a2 + a3)
#define ADDCO(X) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(X) : "v"(b) : "vcc");
DEFINE_KERNEL(add_co_u32, I32_DECL, , ADDCO(a0) ADDCO(a1) ADDCO(a2) ADDCO(a3), ADDCO(a0) ADDCO(a0) ADDCO(a0) ADDCO(a0), a0 + a1 + a2 + a3)
// 5 independent in-place mad chains (the radix-2^26 column accumulators)
#define MAD5 asm volatile("v_mad_u64_u32 %0, vcc, %5, %6, %0\n\tv_mad_u64_u32 %1, vcc, %5, %7, %1\n\tv_mad_u64_u32 %2, vcc, %6, %7, %2\n\tv_mad_u64_u32 %3, vcc, %5, %5, %3\n\tv_mad_u64_u32 %4, vcc, %6, %6, %4" \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(d) : "v"(b), "v"(c), "v"(e0) : "vcc");
DEFINE_KERNEL(mad_u64_u32_5chains, I64_DECL; uint32_t e0 = seed * 3, , MAD5 MAD5 MAD5 MAD5, MAD5 MAD5 MAD5 MAD5, a0 + a1 + a2 + a3 + d)

__global__ void k_calib(uint64_t* out) {
    uint64_t t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    uint32_t x = threadIdx.x;
    for (int i = 0; i < 200000; ++i) asm volatile("v_add_u32 %0, %0, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, %0, %0" : "+v"(x));
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    if (x == 0x12345) out[100] = x;
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

struct Row { std::string name; int per_body; };

template <typename K>
void run(const char* name, K kern4, K kern1, int instr_per_macro, uint64_t* d_out) {
    // instr count per loop iteration: 4 macro groups, each BODY has (ILP4: 4, D1: 4) macro invocations
    for (int ilp : {4, 1}) {
        printf("%-28s ILP=%d :", name, ilp);
        for (int wps : {1, 2, 4, 8}) {
            int threads = 64 * 4 * wps;
            std::vector<uint64_t> h(threads / 64);
            for (int rep = 0; rep < 2; ++rep) {
                if (ilp == 4) hipLaunchKernelGGL(kern4, dim3(1), dim3(threads), 0, 0, d_out, 12345u);
                else hipLaunchKernelGGL(kern1, dim3(1), dim3(threads), 0, 0, d_out, 12345u);
                CHECK(hipDeviceSynchronize());
            }
            CHECK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            double cyc = (double)h[h.size() / 2];
            double n_instr = (double)ITERS * 16 * instr_per_macro;   // per wave
            double per_wave = cyc / n_instr;                          // cycles between a wave's instrs
            printf("  W=%d: %6.2f cyc/instr/wave -> %5.2f cyc/instr/SIMD", wps, per_wave, per_wave / wps);
        }
        printf("\n");
    }
}

#define RUN(NAME, N) run(#NAME, k_##NAME<4>, k_##NAME<1>, N, d_out)

int main() {
    uint64_t* d_out;
    CHECK(hipMalloc(&d_out, 8 * 8192));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d kHz  (s_memtime ticks = shader cycles per MI355X_MICROARCH.md)\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    {   // calibrate the s_memtime tick against s_memrealtime (100 MHz) and the host wall clock
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_calib, dim3(1), dim3(64), 0, 0, d_out);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            uint64_t h[2]; CHECK(hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost));
            printf("calib: s_memtime delta=%llu  s_memrealtime delta=%llu (100 MHz -> %.3f ms)  event %.3f ms  => s_memtime = %.1f MHz; 800k dependent v_add_u32 -> %.2f ticks each\n",
                   (unsigned long long)h[0], (unsigned long long)h[1], h[1] / 1e5, ms, h[0] / (h[1] / 100.0), h[0] / 800000.0);
        }
    }
    RUN(fma_f32, 1);
    RUN(add_u32, 1);
    RUN(mov_b32, 1);
    RUN(add3_u32, 1);
    RUN(alignbit, 1);
    RUN(and_or, 1);
    RUN(cndmask, 1);
    RUN(addco_addc_pair, 1);   // ILP4 body has 2 pairs (=4 instr) and D1 has 2 pairs too -> count pairs*2
    RUN(mul_lo_u32, 1);
    RUN(mul_hi_u32, 1);
    RUN(mul_u32_u24, 1);
    RUN(mul_hi_u32_u24, 1);
    RUN(mad_u32_u24, 1);
    RUN(mad_u64_u32, 1);
    RUN(mad_u64_u32_srcdst_differ, 1);
    RUN(lshl_add_u64, 1);
    RUN(lshrrev_b64, 1);
    RUN(fma_f64, 1);
    RUN(add_f64, 1);
    RUN(mul_f64, 1);
    RUN(mix_mad_2mov, 3);
    RUN(bfi_b32, 1);
    RUN(cndmask_e64_sgpr, 1);
    RUN(cndmask_vcc_set, 1);
    RUN(xor_b32, 1);
    RUN(lshrrev_b32, 1);
    RUN(add_co_u32, 1);
    RUN(mad_u64_u32_5chains, 5);
    return 0;
}
