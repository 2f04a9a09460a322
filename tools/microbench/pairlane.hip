// Two lanes per element: the only lever against one-wave-per-SIMD at 2^16 elements that DESIGN.md had
// costed but never measured.  One ladder step (DBL + ADD of a table entry, the DAG of curve4q.py:138-171) is run
//   A  as shipped: one lane per element, dbl<2> + add_entry<2> of curve.hip.h (signed limbs, chained carries), ONE wave per SIMD;
//   B  pair-lane:  lane 2k holds the real parts, lane 2k+1 the imaginary parts of every GF(p^2) value of element k; a product
//      r = u*v + w*z of 50 multiply-adds per lane gives the even lane its real and the odd lane its imaginary part, the partner's
//      halves arrive by DPP quad_perm moves (no LDS, no waits), additions cost half; TWO waves per SIMD for the same elements.
//   B2 as B with the two products of a lane accumulated in two independent chains (one more 64-bit add per column).
// Same elements, same number of steps, kernel wall time with every SIMD busy; B must end on the residues of A.
//     hipcc -O3 --offload-arch=gfx950 -std=c++17 -o pairlane pairlane.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define FQ_CHAIN 1
#include "../../fourq_amd/csrc/curve.hip.h"
using namespace fq;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int STEPS = 64, REPS = 8;           // REPS ladders of 64 steps per launch

// ---- pair-lane field: this lane's half (5 signed limbs) of a GF(p^2) value -------------------------------------------
struct PF { u32 l[5]; };
constexpr int DPP_SWAP = 0xB1, DPP_EVEN = 0xA0, DPP_ODD = 0xF5;      // quad_perm [1,0,3,2], [0,0,2,2], [1,1,3,3]
template <int CTRL> FQ_DEV PF dpp(const PF& a) {
    PF r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.l[i] = (u32)__builtin_amdgcn_mov_dpp((int)a.l[i], CTRL, 0xF, 0xF, true);
    return r;
}
FQ_DEV PF padd(const PF& a, const PF& b) { PF r; for (int i = 0; i < 5; i++) r.l[i] = a.l[i] + b.l[i]; return r; }
FQ_DEV PF psub(const PF& a, const PF& b) { PF r; for (int i = 0; i < 5; i++) r.l[i] = a.l[i] - b.l[i]; return r; }
FQ_DEV PF pcneg(const PF& a, u32 m) { PF r; for (int i = 0; i < 5; i++) r.l[i] = (a.l[i] ^ m) - m; return r; }
#define PL_OPAQUE(x) asm("" : "+v"(x))
// r = u*v + w*z on signed limbs, carries chained (the column loop of fe2_mul_signed for ONE component)
template <bool TWO_CHAINS> FQ_DEV PF mac2(const PF& u, const PF& v, const PF& w, const PF& z) {
    u32 v8[5], z8[5];
#pragma unroll
    for (int i = 0; i < 5; i++) { v8[i] = v.l[i] << 3; z8[i] = z.l[i] << 3; }
    u32 l[5];
    i64 acc = 0;
#pragma unroll
    for (int K = 0; K < 5; K++) {
        i64 side = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int j = K - i;
            const u32 q0 = j >= 0 ? v.l[j >= 0 ? j : 0] : v8[j >= 0 ? 0 : j + 5];
            const u32 q1 = j >= 0 ? z.l[j >= 0 ? j : 0] : z8[j >= 0 ? 0 : j + 5];
            acc += smul(u.l[i], q0); PL_OPAQUE(acc);
            if (TWO_CHAINS) { side += smul(w.l[i], q1); PL_OPAQUE(side); }
            else { acc += smul(w.l[i], q1); PL_OPAQUE(acc); }
        }
        if (TWO_CHAINS) acc += side;
        l[K] = (u32)acc & LIMB_MASK; acc >>= LIMB_BITS;
    }
    Fe<1> f = fe_finish_signed(l[0], l[1], l[2], l[3], l[4], acc);
    PF r; for (int i = 0; i < 5; i++) r.l[i] = f.l[i];
    return r;
}
// even lane: re = a_re*b_re - a_im*b_im ; odd lane: im = a_im*b_re + a_re*b_im.   With `mine` = this lane's half:
//   u = a_mine, v = b_re (both lanes), w = +-a_other (minus on even lanes), z = b_im (both lanes)  -- wait: odd lane needs
//   a_im*b_re + a_re*b_im = a_mine*b_re + a_other*b_im: the same shape with the sign dropped.
template <bool TWO> FQ_DEV PF pmul(const PF& a, const PF& b, u32 even) {
    const PF w = pcneg(dpp<DPP_SWAP>(a), even);
    return mac2<TWO>(a, dpp<DPP_EVEN>(b), w, dpp<DPP_ODD>(b));
}
// even lane: re = (a_re + a_im)(a_re - a_im) ; odd lane: im = (2 a_re) a_im:   one product u*v per lane.
// WIDE: the operand has bound 2 (DBL's (X+Y)^2), 8*v would not fit a signed 32-bit operand: the wrap-around factor 8 is split
// as (4u)(2v), as wrap_operands_signed does in fp127.hip.h.
template <bool WIDE = false> FQ_DEV PF psqr(const PF& a, u32 even) {
    const PF o = dpp<DPP_SWAP>(a);
    PF u, v;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        u.l[i] = o.l[i] + __builtin_amdgcn_bitop3_b32(even, a.l[i], o.l[i], 0xCA);      // even: a + o ; odd: 2 o
        v.l[i] = a.l[i] - (o.l[i] & even);                                              // even: a - o ; odd: a
    }
    u32 v8[5], uw[5];
#pragma unroll
    for (int i = 0; i < 5; i++) { v8[i] = v.l[i] << (WIDE ? 1 : 3); uw[i] = WIDE ? u.l[i] << 2 : u.l[i]; }
    u32 l[5];
    i64 acc = 0;
#pragma unroll
    for (int K = 0; K < 5; K++) {
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int j = K - i;
            acc += smul(j >= 0 ? u.l[i] : uw[i], j >= 0 ? v.l[j >= 0 ? j : 0] : v8[j >= 0 ? 0 : j + 5]); PL_OPAQUE(acc);
        }
        l[K] = (u32)acc & LIMB_MASK; acc >>= LIMB_BITS;
    }
    Fe<1> f = fe_finish_signed(l[0], l[1], l[2], l[3], l[4], acc);
    PF r; for (int i = 0; i < 5; i++) r.l[i] = f.l[i];
    return r;
}
struct PPoint { PF X, Y, Z, Ta, Tb; };
template <bool TWO> FQ_DEV PPoint pstep(const PPoint& q, const PF& tN, const PF& tD, const PF& tE, const PF& tF, u32 neg, u32 even) {
    // DBL (curve4q.py:138-152, the signed DAG of curve.hip.h)
    PF A = psqr(q.X, even), B = psqr(q.Y, even);
    PF C = psqr(q.Z, even); C = padd(C, C);
    PF D = padd(A, B);
    PF E = psub(psqr<true>(padd(q.X, q.Y), even), D);
    PF F = psub(B, A);
    PF G = psub(C, F);
    PF X = pmul<TWO>(G, E, even), Z = pmul<TWO>(G, F, even), Y = pmul<TWO>(D, F, even);
    // ADD (curve4q.py:155-175) of the entry (N, D exchanged by masked selects, -F by a conditional negation)
    PF sN, sD;
#pragma unroll
    for (int i = 0; i < 5; i++) { sN.l[i] = __builtin_amdgcn_bitop3_b32(neg, tD.l[i], tN.l[i], 0xCA); sD.l[i] = __builtin_amdgcn_bitop3_b32(neg, tN.l[i], tD.l[i], 0xCA); }
    PF T = pmul<TWO>(E, D, even);
    PF N1 = padd(X, Y), D1 = psub(Y, X);
    PF a = pmul<TWO>(D1, sD, even), b = pmul<TWO>(N1, sN, even);
    PF c = pmul<TWO>(pcneg(tF, neg), T, even), d = pmul<TWO>(tE, Z, even);
    PF e = psub(b, a), f = psub(d, c), g = padd(d, c), h = padd(b, a);
    PPoint r;
    r.X = pmul<TWO>(e, f, even); r.Z = pmul<TWO>(g, f, even); r.Y = pmul<TWO>(g, h, even);
    r.Ta = e; r.Tb = h;
    return r;
}

FQ_DEV Fe2<1> seed_fe2(u32 seed, u32 elem, u32 k) {
    Fe2<1> x;
    for (int i = 0; i < 5; i++) { x.re.l[i] = (seed * (2 * i + 1 + k) + elem * (k + 3)) & LIMB_MASK; x.im.l[i] = (seed * (2 * i + 2 + 7 * k) ^ (elem * 5 + k)) & LIMB_MASK; }
    return x;
}
// A: the shipped step, one lane per element
__global__ __launch_bounds__(256, 1) void single_kernel(uint64_t* out, uint32_t seed) {
    const u32 elem = blockIdx.x * 256 + threadIdx.x;
    EntryRegs t; t.N = seed_fe2(seed, elem, 1); t.D = seed_fe2(seed, elem, 2); t.E = seed_fe2(seed, elem, 3); t.F = seed_fe2(seed, elem, 4);
    R1 Q; Q.X = seed_fe2(seed, elem, 5); Q.Y = seed_fe2(seed, elem, 6); Q.Z = seed_fe2(seed, elem, 7); Q.Ta = widen<4>(Q.X); Q.Tb = widen<2>(Q.Y);
#pragma unroll 1
    for (int it = 0; it < STEPS * REPS; ++it) {
        const u32 neg = 0u - ((elem >> (it & 15)) & 1);
        Q = dbl<2>(Q.X, Q.Y, Q.Z);
        Q = add_entry<2>(Q, t, neg);
    }
    if (elem < 64) { u64 w[12]; store_fe2(w, fe2_unsign(Q.X)); store_fe2(w + 4, fe2_unsign(Q.Y)); store_fe2(w + 8, fe2_unsign(Q.Z)); for (int i = 0; i < 12; i++) out[elem * 12 + i] = w[i]; }
}
// B: two lanes per element
template <bool TWO> __global__ __launch_bounds__(256, 2) void pair_kernel(uint64_t* out, uint32_t seed) {
    const u32 lane = blockIdx.x * 256 + threadIdx.x, elem = lane >> 1, odd = lane & 1, even = odd - 1u;
    auto half = [&](u32 k) { const Fe2<1> x = seed_fe2(seed, elem, k); PF r; for (int i = 0; i < 5; i++) r.l[i] = odd ? x.im.l[i] : x.re.l[i]; return r; };
    const PF tN = half(1), tD = half(2), tE = half(3), tF = half(4);
    PPoint Q; Q.X = half(5); Q.Y = half(6); Q.Z = half(7); Q.Ta = Q.X; Q.Tb = Q.Y;
#pragma unroll 1
    for (int it = 0; it < STEPS * REPS; ++it) {
        const u32 neg = 0u - ((elem >> (it & 15)) & 1);
        Q = pstep<TWO>(Q, tN, tD, tE, tF, neg, even);
    }
    if (elem < 64) {                                            // reassemble (re, im) through memory: canonical halves
        const PF* c[3] = { &Q.X, &Q.Y, &Q.Z };
        for (int k = 0; k < 3; k++) {
            Fe<1> f; for (int i = 0; i < 5; i++) f.l[i] = c[k]->l[i];
            u64 lo, hi; fe_canon(fe_unsign(f), lo, hi);
            out[elem * 12 + 4 * k + 2 * odd] = lo; out[elem * 12 + 4 * k + 2 * odd + 1] = hi;
        }
    }
}
int main() {
    uint64_t* d; CHECK(hipMalloc(&d, 3 * 768 * 8));
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const char* names[3] = { "A  one lane per element, 1 wave/SIMD (shipped step)", "B  two lanes per element, 2 waves/SIMD", "B2 two lanes, two accumulation chains per lane" };
    float times[3];
    for (int kind = 0; kind < 3; kind++) {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); float ms = 0, best = 1e9;
        for (int rep = 0; rep < 6; rep++) {
            CHECK(hipEventRecord(e0));
            if (kind == 0) hipLaunchKernelGGL(single_kernel, dim3(cus), dim3(256), 0, 0, d, 12345u);
            else if (kind == 1) hipLaunchKernelGGL(pair_kernel<false>, dim3(2 * cus), dim3(256), 0, 0, d + 768, 12345u);
            else hipLaunchKernelGGL(pair_kernel<true>, dim3(2 * cus), dim3(256), 0, 0, d + 1536, 12345u);
            CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        times[kind] = best;
        printf("%-58s %8.3f ms for %d elements x %d steps -> %7.1f ns per step per 64 elements\n", names[kind], best, cus * 256, STEPS * REPS,
               best * 1e6 / (STEPS * REPS));
    }
    printf("pair-lane / shipped: B %.3f, B2 %.3f  (< 1 would be a gain)\n", times[1] / times[0], times[2] / times[0]);
    {   // latency view: HALF the elements (one wave per SIMD of pair lanes): what a batch of <= 32 768 elements would pay per step
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); float ms = 0, best = 1e9;
        for (int rep = 0; rep < 6; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(pair_kernel<false>, dim3(cus), dim3(256), 0, 0, d + 768, 12345u);
            CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        printf("B at ONE wave per SIMD (32 768 elements): %8.3f ms -> %7.1f ns per step, %.3f of the shipped step's latency\n", best, best * 1e6 / (STEPS * REPS), best / times[0]);
    }
    uint64_t h[3 * 768];
    CHECK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    int same = 1;
    for (int i = 0; i < 768; i++) same &= (h[i] == h[768 + i]) && (h[i] == h[1536 + i]);
    printf("the three variants end on the same residues: %s\n", same ? "yes" : "NO");
    return same ? 0 : 1;
}
