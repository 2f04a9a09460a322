// GF(p^2) multiplication with three GF(p) products (Karatsuba; fields.py:166-173 notes it is result-identical) against
// the four-product forms of fp127.hip.h, on the signed-limb flavour where column differences need no bias:
//     P0 = a0*b0, P1 = a1*b1, P2 = (a0+a1)*(b0+b1);   re = P0 - P1;   im = P2 - P0 - P1      (75 multiply-adds, 15 64-bit
//     column subtractions, 10 operand additions, one more times-8 operand) against 100 multiply-adds.
// Kernel wall time with every SIMD busy, 1 and 4 waves per SIMD; the three variants must end on the same residues.
//     hipcc -O3 --offload-arch=gfx950 -std=c++17 -o karatsuba karatsuba.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define FQ_CHAIN 1
#include "../../fourq_amd/csrc/curve.hip.h"
using namespace fq;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int A, int B> FQ_DEV Fe2<1> fe2_mul_kara(const Fe2<A>& a, const Fe2<B>& b) {
    static_assert(cols_ok_signed((u64)4 * A * B), "column overflow");
    static_assert(fits8_signed<2 * B>(), "8*(b0+b1) does not fit a signed 32-bit operand");
    u32 sa[5], sb[5], b0x8[5], b1x8[5], sbx8[5];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        sa[i] = a.re.l[i] + a.im.l[i]; sb[i] = b.re.l[i] + b.im.l[i];
        b0x8[i] = b.re.l[i] << 3; b1x8[i] = b.im.l[i] << 3; sbx8[i] = sb[i] << 3;
    }
    i64 re = 0, im = 0;
    u32 lr[5], li[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        i64 p0 = 0, p1 = 0, p2 = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int j = k - i;
            p0 += smul(a.re.l[i], j >= 0 ? b.re.l[j >= 0 ? j : 0] : b0x8[j >= 0 ? 0 : j + 5]);
            p1 += smul(a.im.l[i], j >= 0 ? b.im.l[j >= 0 ? j : 0] : b1x8[j >= 0 ? 0 : j + 5]);
            p2 += smul(sa[i], j >= 0 ? sb[j >= 0 ? j : 0] : sbx8[j >= 0 ? 0 : j + 5]);
        }
        re += p0 - p1;
        im += p2 - p0 - p1;
        lr[k] = (u32)re & LIMB_MASK; re >>= LIMB_BITS;
        li[k] = (u32)im & LIMB_MASK; im >>= LIMB_BITS;
    }
    Fe2<1> r;
    r.re = fe_finish_signed(lr[0], lr[1], lr[2], lr[3], lr[4], re);
    r.im = fe_finish_signed(li[0], li[1], li[2], li[3], li[4], im);
    return r;
}

constexpr int ITERS = 400;
template <int KIND> __global__ __launch_bounds__(256, 4) void k(uint64_t* out, uint32_t seed) {
    Fe2<1> x, y;
    const u32 t = blockIdx.x * 256 + threadIdx.x;
    for (int i = 0; i < 5; i++) { x.re.l[i] = (seed * (i + 1) + t) & LIMB_MASK; x.im.l[i] = (seed * (i + 7) + t * 3) & LIMB_MASK;
                                  y.re.l[i] = (seed * (i + 3) ^ t) & LIMB_MASK; y.im.l[i] = (seed * (i + 11) + 5) & LIMB_MASK; }
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
        if (KIND == 0) { x = fe2_mul_chain(x, y); y = fe2_mul_chain(y, x); x = fe2_mul_chain(x, y); y = fe2_mul_chain(y, x); }
        else if (KIND == 1) { x = fe2_mul_signed(x, y); y = fe2_mul_signed(y, x); x = fe2_mul_signed(x, y); y = fe2_mul_signed(y, x); }
        else { x = fe2_mul_kara(x, y); y = fe2_mul_kara(y, x); x = fe2_mul_kara(x, y); y = fe2_mul_kara(y, x); }
    }
    if (KIND != 0) { x = fe2_unsign(x); y = fe2_unsign(y); }
    if (t < 64) { u64 w[8]; store_fe2(w, x); store_fe2(w + 4, y); for (int i = 0; i < 8; i++) out[(size_t)KIND * 512 + t * 8 + i] = w[i]; }
}
int main() {
    uint64_t* d; CHECK(hipMalloc(&d, 3 * 512 * 8));
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const char* names[3] = { "4 products, unsigned, chained carries", "4 products, signed limbs", "3 products (Karatsuba), signed limbs" };
    for (int w : { 1, 4 })
        for (int kind = 0; kind < 3; kind++) {
            const int blocks = p.multiProcessorCount * w;
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); float ms = 0, best = 1e9;
            for (int rep = 0; rep < 5; rep++) {
                CHECK(hipEventRecord(e0));
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, 123u);
                else if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, 123u);
                else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, 123u);
                CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("%-40s waves/SIMD %d : %7.3f ms -> %6.1f ns per GF(p^2) product per wave per SIMD\n", names[kind], w, best, best * 1e6 / (4.0 * ITERS * w));
        }
    uint64_t h[3 * 512];
    CHECK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    int same = 1;
    for (int i = 0; i < 512; i++) same &= (h[i] == h[512 + i]) && (h[i] == h[1024 + i]);
    printf("the three variants end on the same residues: %s\n", same ? "yes" : "NO");
    return same ? 0 : 1;
}
