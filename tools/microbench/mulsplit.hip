// GF(p^2) product on signed limbs, two ways of handling the wrap-around columns (2^130 == 8):
//   (1) fp127.hip.h's fe2_mul_signed: the second operand pre-multiplied by 8 (10 shifts), one accumulation chain per component;
//   (2) the wrap-around products of a column summed on their own and folded in with one v_lshl_add_u64 (carry + (hi << 3)):
//       8 folds instead of 10 shifts, and four chains in flight instead of two (fewer s_nop between dependent multiply-adds).
// A chain of dependent products and four independent chains side by side (the ladder's situation), 1 and 4 waves per SIMD.
//     hipcc -O3 --offload-arch=gfx950 -std=c++17 -o mulsplit mulsplit.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define FQ_CHAIN 1
#include "../../fourq_amd/csrc/curve.hip.h"
using namespace fq;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define OPQ(x) asm("" : "+v"(x))

template <int A, int B> FQ_DEV Fe2<1> fe2_mul_split(const Fe2<A>& a, const Fe2<B>& b) {
    static_assert(cols_ok_signed((u64)2 * A * B), "column overflow");
    Fe<A> na1 = fe_neg_signed(a.im);
    i64 re = 0, im = 0;
    u32 lr[5], li[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        if (k < 4) {
            i64 hr = 0, hi = 0;
#pragma unroll
            for (int i = k + 1; i < 5; i++) {
                const int j = k + 5 - i;
                hr += smul(a.re.l[i], b.re.l[j]); OPQ(hr);
                hi += smul(a.re.l[i], b.im.l[j]); OPQ(hi);
                hr += smul(na1.l[i], b.im.l[j]); OPQ(hr);
                hi += smul(a.im.l[i], b.re.l[j]); OPQ(hi);
            }
            re += hr * 8; im += hi * 8;
        }
#pragma unroll
        for (int i = 0; i <= k; i++) {
            const int j = k - i;
            re += smul(a.re.l[i], b.re.l[j]); OPQ(re);
            im += smul(a.re.l[i], b.im.l[j]); OPQ(im);
            re += smul(na1.l[i], b.im.l[j]); OPQ(re);
            im += smul(a.im.l[i], b.re.l[j]); OPQ(im);
        }
        lr[k] = (u32)re & LIMB_MASK; re >>= LIMB_BITS;
        li[k] = (u32)im & LIMB_MASK; im >>= LIMB_BITS;
    }
    Fe2<1> r;
    r.re = fe_finish_signed(lr[0], lr[1], lr[2], lr[3], lr[4], re);
    r.im = fe_finish_signed(li[0], li[1], li[2], li[3], li[4], im);
    return r;
}

constexpr int ITERS = 400;
// KIND 0/1: one dependent chain (shipped / split); KIND 2/3: four independent chains side by side (shipped / split)
template <int KIND> __global__ __launch_bounds__(256, 4) void k(uint64_t* out, uint32_t seed) {
    Fe2<1> x[4], y[4];
    const u32 t = blockIdx.x * 256 + threadIdx.x;
    for (int c = 0; c < 4; c++)
        for (int i = 0; i < 5; i++) { x[c].re.l[i] = (seed * (i + 1 + c) + t) & LIMB_MASK; x[c].im.l[i] = (seed * (i + 7) + t * 3 + c) & LIMB_MASK;
                                      y[c].re.l[i] = (seed * (i + 3) ^ (t + c)) & LIMB_MASK; y[c].im.l[i] = (seed * (i + 11) + 5 + c) & LIMB_MASK; }
    constexpr int CH = KIND >= 2 ? 4 : 1;
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int rep = 0; rep < (KIND >= 2 ? 1 : 4); rep++) {
#pragma unroll
            for (int c = 0; c < CH; c++) x[c] = (KIND & 1) ? fe2_mul_split(x[c], y[c]) : fe2_mul_signed(x[c], y[c]);
#pragma unroll
            for (int c = 0; c < CH; c++) { Fe2<1> tmp = x[c]; x[c] = y[c]; y[c] = tmp; }
        }
    }
    u64 acc[8] = {0};
    for (int c = 0; c < CH; c++) { u64 w[8]; store_fe2(w, fe2_unsign(x[c])); for (int i = 0; i < 4; i++) acc[i] ^= w[i]; store_fe2(w, fe2_unsign(y[c])); for (int i = 0; i < 4; i++) acc[4 + i] ^= w[i]; }
    if (t < 64) for (int i = 0; i < 8; i++) out[(size_t)KIND * 512 + t * 8 + i] = acc[i];
}
int main() {
    uint64_t* d; CHECK(hipMalloc(&d, 4 * 512 * 8));
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const char* names[4] = { "dependent chain, shipped (b x 8)", "dependent chain, split wrap-around columns", "4 chains side by side, shipped", "4 chains side by side, split" };
    for (int w : { 1, 4 })
        for (int kind = 0; kind < 4; kind++) {
            const int blocks = p.multiProcessorCount * w;
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); float ms = 0, best = 1e9;
            for (int rep = 0; rep < 5; rep++) {
                CHECK(hipEventRecord(e0));
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, 123u);
                else if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, 123u);
                else if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, 123u);
                else hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, 123u);
                CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("%-46s waves/SIMD %d : %7.3f ms -> %6.1f ns per GF(p^2) product per wave per SIMD\n", names[kind], w, best, best * 1e6 / (4.0 * ITERS * w));
        }
    uint64_t h[4 * 512];
    CHECK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    bool same = true;
    for (int i = 0; i < 512; i++) same = same && h[i] == h[512 + i] && h[1024 + i] == h[1536 + i];
    printf("residues equal: %s\n", same ? "yes" : "NO");
    return same ? 0 : 1;
}
