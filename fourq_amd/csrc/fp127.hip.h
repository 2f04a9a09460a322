// GF(p) and GF(p^2) arithmetic for p = 2^127 - 1 on gfx950 (MI355X) VALU.
//
// Replaces, on the device, the reference's big-int field layer:
//   GFp.add/sub/mul/sqr/neg/inv      bifurcation/fourq impl/fields.py:29-106
//   GFp2.add/sub/mul/sqr/neg/conj/inv impl/fields.py:156-199
//
// Representation (chosen from tools/microbench measurements on MI355X, profiles/true_rates_r01.txt:
// per SIMD a wave64 v_mad_u64_u32 takes 4 cycles, every other 64-bit or carry-propagating op ~3.7, a
// plain 32-bit VOP2 op ~2; so carries, not multiplies, are what to avoid): five 26-bit limbs in 32-bit registers, radix 2^26,
// 130 bits, 2^130 == 8 (mod p).  Limbs are kept LAZILY: adds and subtractions are five plain
// v_add_u32 / v_sub_u32 with no carry propagation; only multiplications normalise.  Every element
// type carries a compile-time bound B meaning "each limb <= B * UNIT" (UNIT = 2^26 + 2^15), and
// every multiplication static_asserts that its 64-bit column accumulators cannot overflow, so the
// laziness is verified by the compiler, not by hope.
//
// A GF(p) product is 25 v_mad_u64_u32 accumulating into 64-bit columns (the wrap-around terms use
// the second operand pre-multiplied by 8); a GF(p^2) product accumulates its two GF(p) products per
// component into the same columns (a0*b0 + (-a1)*b1 and a0*b1 + a1*b0): 100 multiply-adds and two
// carry passes (per limb one 64-bit shift, one mask, one 64-bit add).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fq {

typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef int64_t i64;

#define FQ_DEV __device__ __forceinline__

// FQ_CHAIN=1: GF(p^2) products keep each column's carry inside the multiply-add chain (opaque partial sums stop
// hipcc from re-associating it into a 64-bit addition per limb): fewer instructions, less freedom for the
// scheduler.  Pays in every kernel but the fused variable-base ones (kernels.hip.h).
#ifndef FQ_CHAIN
#define FQ_CHAIN 0
#endif
// FQ_MUL_ASM=1: in a translation unit with FQ_CHAIN=0 the GF(p^2) products and squares of the unsigned flavour are the generated
// instruction streams FQ_ASM_MULU / FQ_ASM_SQRU (ladder_asm.hip.h, tools/asmgen/gen_ladder_step.py): no fences, no hazard nops.
#ifndef FQ_MUL_ASM
#define FQ_MUL_ASM 1
#endif

constexpr u32 LIMB_BITS = 26;
constexpr u32 LIMB_MASK = (1u << LIMB_BITS) - 1;
constexpr u64 UNIT = (1ull << 26) + (1ull << 15);   // bound of a limb right after normalisation

// limbs of K * (2^130 - 8) == 0 (mod p) in redundant form: used as the bias of subtractions
FQ_DEV constexpr u32 bias_limb(int k, int i) { return (u32)k * (i == 0 ? (LIMB_MASK - 7) : LIMB_MASK); }

template <int B> struct Fe {   // GF(p) element, every limb <= B * UNIT
    u32 l[5];
};
template <int B> struct Fe2 {  // GF(p^2) element re + im*i
    Fe<B> re, im;
};

// ---- bound widening (free) -----------------------------------------------------------------
template <int B2, int B> FQ_DEV Fe<B2> widen(const Fe<B>& a) {
    static_assert(B2 >= B, "cannot narrow a bound without a carry pass");
    Fe<B2> r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.l[i] = a.l[i];
    return r;
}
template <int B2, int B> FQ_DEV Fe2<B2> widen(const Fe2<B>& a) {
    Fe2<B2> r; r.re = widen<B2>(a.re); r.im = widen<B2>(a.im); return r;
}

// ---- add / sub / neg (lazy) ----------------------------------------------------------------
template <int A, int B> FQ_DEV Fe<A + B> fe_add(const Fe<A>& a, const Fe<B>& b) {
    static_assert((u64)(A + B) * UNIT < (1ull << 32), "limb overflow");
    Fe<A + B> r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
// a - b computed as a + ((B+1)*(2^130-8) - b): limbwise non-negative because
// (B+1)*(2^26-8) >= B*UNIT for every B used here (checked below).
template <int A, int B> FQ_DEV Fe<A + B + 1> fe_sub(const Fe<A>& a, const Fe<B>& b) {
    static_assert((u64)(A + B + 1) * UNIT < (1ull << 32), "limb overflow");
    static_assert((u64)(B + 1) * (LIMB_MASK - 7) >= (u64)B * UNIT, "bias too small");
    Fe<A + B + 1> r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.l[i] = a.l[i] + (bias_limb(B + 1, i) - b.l[i]);
    return r;
}
template <int B> FQ_DEV Fe<B + 1> fe_neg(const Fe<B>& b) {
    static_assert((u64)(B + 1) * (LIMB_MASK - 7) >= (u64)B * UNIT, "bias too small");
    Fe<B + 1> r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.l[i] = bias_limb(B + 1, i) - b.l[i];
    return r;
}
template <int B> FQ_DEV Fe<2 * B> fe_dbl(const Fe<B>& a) { return fe_add(a, a); }

// 32-bit carry pass: any bound -> bound 1 (15 cheap VALU ops).
template <int B> FQ_DEV Fe<1> fe_carry(const Fe<B>& a) {
    static_assert((u64)B * UNIT + (1ull << 20) < (1ull << 32), "limb overflow");
    Fe<1> r;
    u32 t1 = a.l[1] + (a.l[0] >> LIMB_BITS);
    u32 t2 = a.l[2] + (t1 >> LIMB_BITS);
    u32 t3 = a.l[3] + (t2 >> LIMB_BITS);
    u32 t4 = a.l[4] + (t3 >> LIMB_BITS);
    u32 t0 = (a.l[0] & LIMB_MASK) + ((t4 >> LIMB_BITS) << 3);   // < 2^26 + 2^9
    r.l[0] = t0 & LIMB_MASK;
    r.l[1] = (t1 & LIMB_MASK) + (t0 >> LIMB_BITS);
    r.l[2] = t2 & LIMB_MASK;
    r.l[3] = t3 & LIMB_MASK;
    r.l[4] = t4 & LIMB_MASK;
    return r;
}
template <int B> FQ_DEV Fe2<1> fe2_carry(const Fe2<B>& a) {
    Fe2<1> r; r.re = fe_carry(a.re); r.im = fe_carry(a.im); return r;
}

// ---- column multiply-accumulate -------------------------------------------------------------------
// Column k of a product is  sum_{i+j == k (mod 5)} a_i * (i+j < 5 ? b_j : 8*b_j).  Columns are
// written in order with column k+1 starting from (column k >> 26); hipcc re-associates that into
// five independent in-place v_mad_u64_u32 chains plus one v_lshl_add_u64 per limb for the carry,
// which keeps the multiply-adds free of dependences on the carry chain (better for a lone wave).
template <int B> FQ_DEV void times8(u32 out[5], const Fe<B>& b) {
    static_assert((u64)8 * B * UNIT < (1ull << 32), "8*b does not fit 32 bits");
#pragma unroll
    for (int i = 0; i < 5; i++) out[i] = b.l[i] << 3;
}
// column bound: `weighted` = sum over accumulated products of A*B (in UNIT^2), times 5 terms, times 8
// for the wrap-around factor, plus the incoming carry (< 2^40)
constexpr bool cols_ok(u64 weighted) {
    return weighted * 5 * 8 <= ((~0ull - (1ull << 41)) / (UNIT * UNIT));
}
template <int K> FQ_DEV u64 col_mac(u64 acc, const u32 a[5], const u32 b[5], const u32 b8[5]) {
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const int j = K - i;
        acc += (u64)a[i] * (j >= 0 ? b[j] : b8[j + 5]);   // v_mad_u64_u32, accumulating in place
    }
    return acc;
}
// five limbs (< 2^26 each) + the carry out of column 4 -> bound-1 element (2^130 == 8)
FQ_DEV Fe<1> fe_finish(u32 l0, u32 l1, u32 l2, u32 l3, u32 l4, u64 top) {
    Fe<1> r;
    u64 w = (top << 3) + l0;                 // top < 2^38 -> w < 2^42
    r.l[0] = (u32)w & LIMB_MASK;
    r.l[1] = l1 + (u32)(w >> LIMB_BITS);     // <= 2^26 - 1 + 2^15 = UNIT - 1
    r.l[2] = l2; r.l[3] = l3; r.l[4] = l4;
    return r;
}
// sum of up to two GF(p) products  a*b + c*d  (pass c == nullptr for a single product)
template <bool TWO> FQ_DEV Fe<1> fe_mac2(const u32 a[5], const u32 b[5], const u32 b8[5], const u32 c[5], const u32 d[5], const u32 d8[5]) {
    u32 l[5];
    u64 acc = 0;
#define FQ_COL(K)                                                   \
    acc = col_mac<K>(acc, a, b, b8);                                \
    if (TWO) acc = col_mac<K>(acc, c, d, d8);                       \
    l[K] = (u32)acc & LIMB_MASK; acc >>= LIMB_BITS;
    FQ_COL(0) FQ_COL(1) FQ_COL(2) FQ_COL(3) FQ_COL(4)
#undef FQ_COL
    return fe_finish(l[0], l[1], l[2], l[3], l[4], acc);
}

// ---- GF(p) multiplication ----------------------------------------------------------------------
template <int A, int B> FQ_DEV Fe<1> fe_mul(const Fe<A>& a, const Fe<B>& b) {   // fields.py:42-45
    static_assert(cols_ok((u64)A * B), "column overflow");
    u32 b8[5];
    times8(b8, b);
    return fe_mac2<false>(a.l, b.l, b8, nullptr, nullptr, nullptr);
}
template <int A> FQ_DEV Fe<1> fe_sqr(const Fe<A>& a) {                          // fields.py:48-51
    static_assert(cols_ok((u64)A * A), "column overflow");
    static_assert((u64)8 * A * UNIT < (1ull << 32), "8*a does not fit 32 bits");
    // 15 products: squares once, cross terms through a doubled operand; e = 8a for the wrap-around
    const u32 *x = a.l;
    u32 d[5], e[5];
#pragma unroll
    for (int i = 0; i < 5; i++) { d[i] = x[i] << 1; e[i] = x[i] << 3; }
    u32 l0, l1, l2, l3, l4;
    u64 acc = (u64)x[0] * x[0] + (u64)d[1] * e[4] + (u64)d[2] * e[3];
    l0 = (u32)acc & LIMB_MASK; acc >>= LIMB_BITS;
    acc += (u64)d[0] * x[1] + (u64)d[2] * e[4] + (u64)x[3] * e[3];
    l1 = (u32)acc & LIMB_MASK; acc >>= LIMB_BITS;
    acc += (u64)d[0] * x[2] + (u64)x[1] * x[1] + (u64)d[3] * e[4];
    l2 = (u32)acc & LIMB_MASK; acc >>= LIMB_BITS;
    acc += (u64)d[0] * x[3] + (u64)d[1] * x[2] + (u64)x[4] * e[4];
    l3 = (u32)acc & LIMB_MASK; acc >>= LIMB_BITS;
    acc += (u64)d[0] * x[4] + (u64)d[1] * x[3] + (u64)x[2] * x[2];
    l4 = (u32)acc & LIMB_MASK; acc >>= LIMB_BITS;
    return fe_finish(l0, l1, l2, l3, l4, acc);
}

// ---- GF(p^2) -----------------------------------------------------------------------------------
template <int A, int B> FQ_DEV Fe2<A + B> fe2_add(const Fe2<A>& a, const Fe2<B>& b) {   // fields.py:157-159
    Fe2<A + B> r; r.re = fe_add(a.re, b.re); r.im = fe_add(a.im, b.im); return r;
}
template <int A, int B> FQ_DEV Fe2<A + B + 1> fe2_sub(const Fe2<A>& a, const Fe2<B>& b) {   // fields.py:162-164
    Fe2<A + B + 1> r; r.re = fe_sub(a.re, b.re); r.im = fe_sub(a.im, b.im); return r;
}
template <int B> FQ_DEV Fe2<B + 1> fe2_neg(const Fe2<B>& a) {                              // fields.py:184-186
    Fe2<B + 1> r; r.re = fe_neg(a.re); r.im = fe_neg(a.im); return r;
}
template <int B> FQ_DEV Fe2<B + 1> fe2_conj(const Fe2<B>& a) {                             // fields.py:189-191
    Fe2<B + 1> r; r.re = widen<B + 1>(a.re); r.im = fe_neg(a.im); return r;
}
template <int B> FQ_DEV Fe2<2 * B> fe2_dbl(const Fe2<B>& a) { return fe2_add(a, a); }

// (a0 + a1 i)(b0 + b1 i) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) i                  fields.py:167-173
template <int A, int B> FQ_DEV Fe2<1> fe2_mul_chain(const Fe2<A>& a, const Fe2<B>& b) {
    static_assert(cols_ok((u64)(2 * A + 1) * B), "column overflow");
    u32 b0x8[5], b1x8[5];
    times8(b0x8, b.re);
    times8(b1x8, b.im);
    Fe<A + 1> na1 = fe_neg(a.im);
    // re and im columns advance together; each column's accumulator starts from the previous column's carry.
    // The empty asm statements make the partial sums opaque so that hipcc keeps the carry inside the
    // multiply-add chain instead of re-associating it into a separate 64-bit addition per limb.
    u64 re = 0, im = 0;
    u32 lr[5], li[5];
#define FQ_OPAQUE(x) asm("" : "+v"(x))
#define FQ_COL2(K)                                                                          \
    _Pragma("unroll") for (int i = 0; i < 5; i++) {                                         \
        const int j = K - i;                                                                \
        const u32 q0 = j >= 0 ? b.re.l[j >= 0 ? j : 0] : b0x8[j >= 0 ? 0 : j + 5];           \
        const u32 q1 = j >= 0 ? b.im.l[j >= 0 ? j : 0] : b1x8[j >= 0 ? 0 : j + 5];           \
        re += (u64)a.re.l[i] * q0; FQ_OPAQUE(re);                                           \
        im += (u64)a.re.l[i] * q1; FQ_OPAQUE(im);                                           \
        re += (u64)na1.l[i] * q1; FQ_OPAQUE(re);                                            \
        im += (u64)a.im.l[i] * q0; FQ_OPAQUE(im);                                           \
    }                                                                                       \
    lr[K] = (u32)re & LIMB_MASK; re >>= LIMB_BITS;                                          \
    li[K] = (u32)im & LIMB_MASK; im >>= LIMB_BITS;
    FQ_COL2(0) FQ_COL2(1) FQ_COL2(2) FQ_COL2(3) FQ_COL2(4)
#undef FQ_COL2
#undef FQ_OPAQUE
    Fe2<1> r;
    r.re = fe_finish(lr[0], lr[1], lr[2], lr[3], lr[4], re);
    r.im = fe_finish(li[0], li[1], li[2], li[3], li[4], im);
    return r;
}
template <int A, int B> FQ_DEV Fe2<1> fe2_mul_plain(const Fe2<A>& a, const Fe2<B>& b) {
    static_assert(cols_ok((u64)(2 * A + 1) * B), "column overflow");
    u32 b0x8[5], b1x8[5];
    times8(b0x8, b.re);
    times8(b1x8, b.im);
    Fe<A + 1> na1 = fe_neg(a.im);
    Fe2<1> r;
    r.re = fe_mac2<true>(a.re.l, b.re.l, b0x8, na1.l, b.im.l, b1x8);
    r.im = fe_mac2<true>(a.re.l, b.im.l, b1x8, a.im.l, b.re.l, b0x8);
    return r;
}
// MODE selects the variant per call site (0 plain, 1 chained carries, 2 chained carries on SIGNED limbs, see the
// signed flavour at the end of this file); the plain name follows the translation unit's default
template <int A, int B> FQ_DEV Fe2<1> fe2_mul_signed(const Fe2<A>& a, const Fe2<B>& b);
template <int A> FQ_DEV Fe2<1> fe2_sqr_signed(const Fe2<A>& a);
template <int A, int B> FQ_DEV Fe2<1> fe2_mul_asm(const Fe2<A>& a, const Fe2<B>& b);      // ladder_asm.hip.h
template <int A> FQ_DEV Fe2<1> fe2_sqr_asm(const Fe2<A>& a);
template <int MODE, int A, int B> FQ_DEV Fe2<1> fe2_mulx(const Fe2<A>& a, const Fe2<B>& b) {
    if constexpr (MODE == 3) return fe2_mul_asm(a, b);
    else if constexpr (MODE == 2) return fe2_mul_signed(a, b);
    else if constexpr (MODE == 1) return fe2_mul_chain(a, b);
    else return fe2_mul_plain(a, b);
}
constexpr int FE2_DEFAULT_MODE = (FQ_CHAIN != 0) ? 1 : ((FQ_MUL_ASM != 0) ? 3 : 0);
template <int A, int B> FQ_DEV Fe2<1> fe2_mul(const Fe2<A>& a, const Fe2<B>& b) { return fe2_mulx<FE2_DEFAULT_MODE>(a, b); }
// (a0 + a1 i)^2 = (a0 + a1)(a0 - a1) + (2 a0 a1) i                               fields.py:176-181
template <int A> FQ_DEV Fe2<1> fe2_sqr_chain(const Fe2<A>& a) {
    Fe<2 * A> s = fe_add(a.re, a.im);
    Fe<2 * A + 1> d = fe_sub(a.re, a.im);
    Fe<2 * A> t = fe_dbl(a.re);
    static_assert(cols_ok((u64)(2 * A + 1) * (2 * A)), "column overflow");
    u32 s8[5], i8[5];
    times8(s8, s);
    times8(i8, a.im);
    // re = d * s and im = t * a.im advance together, carries chained as in fe2_mul
    u64 re = 0, im = 0;
    u32 lr[5], li[5];
#define FQ_OPAQUE(x) asm("" : "+v"(x))
#define FQ_COL2(K)                                                                          \
    _Pragma("unroll") for (int i = 0; i < 5; i++) {                                         \
        const int j = K - i;                                                                \
        const u32 q0 = j >= 0 ? s.l[j >= 0 ? j : 0] : s8[j >= 0 ? 0 : j + 5];               \
        const u32 q1 = j >= 0 ? a.im.l[j >= 0 ? j : 0] : i8[j >= 0 ? 0 : j + 5];            \
        re += (u64)d.l[i] * q0; FQ_OPAQUE(re);                                              \
        im += (u64)t.l[i] * q1; FQ_OPAQUE(im);                                              \
    }                                                                                       \
    lr[K] = (u32)re & LIMB_MASK; re >>= LIMB_BITS;                                          \
    li[K] = (u32)im & LIMB_MASK; im >>= LIMB_BITS;
    FQ_COL2(0) FQ_COL2(1) FQ_COL2(2) FQ_COL2(3) FQ_COL2(4)
#undef FQ_COL2
#undef FQ_OPAQUE
    Fe2<1> r;
    r.re = fe_finish(lr[0], lr[1], lr[2], lr[3], lr[4], re);
    r.im = fe_finish(li[0], li[1], li[2], li[3], li[4], im);
    return r;
}

template <int A> FQ_DEV Fe2<1> fe2_sqr_plain(const Fe2<A>& a) {
    Fe<2 * A> s = fe_add(a.re, a.im);
    Fe<2 * A + 1> d = fe_sub(a.re, a.im);
    Fe<2 * A> t = fe_dbl(a.re);
    Fe2<1> r;
    r.re = fe_mul(d, s);
    r.im = fe_mul(t, a.im);
    return r;
}

template <int MODE, int A> FQ_DEV Fe2<1> fe2_sqrx(const Fe2<A>& a) {
    if constexpr (MODE == 3) return fe2_sqr_asm(a);
    else if constexpr (MODE == 2) return fe2_sqr_signed(a);
    else if constexpr (MODE == 1) return fe2_sqr_chain(a);
    else return fe2_sqr_plain(a);
}
template <int A> FQ_DEV Fe2<1> fe2_sqr(const Fe2<A>& a) { return fe2_sqrx<FE2_DEFAULT_MODE>(a); }

// ---- canonical form, packing --------------------------------------------------------------------
// 128-bit little-endian container (what the C ABI carries) <-> limbs.
FQ_DEV Fe<1> fe_unpack(u64 lo, u64 hi) {   // any 128-bit value; the residue class is what counts
    Fe<1> r;
    r.l[0] = (u32)lo & LIMB_MASK;
    r.l[1] = (u32)(lo >> 26) & LIMB_MASK;
    r.l[2] = (u32)((lo >> 52) | (hi << 12)) & LIMB_MASK;
    r.l[3] = (u32)(hi >> 14) & LIMB_MASK;
    r.l[4] = (u32)(hi >> 40);              // 24 bits
    return r;
}
// Fully reduced representative in [0, p): what the reference's `% p1271` returns.
template <int B> FQ_DEV void fe_canon(const Fe<B>& a, u64& lo, u64& hi) {
    Fe<1> t = fe_carry(a);                 // limbs tight, value < 2^130 + 2^41
    // fold bits >= 127 (limb 4 holds bits 104..129): 2^127 == 1
    u32 l0 = t.l[0] + (t.l[4] >> 23), l1 = t.l[1], l2 = t.l[2], l3 = t.l[3], l4 = t.l[4] & 0x7fffff;
    l1 += l0 >> 26; l0 &= LIMB_MASK;
    l2 += l1 >> 26; l1 &= LIMB_MASK;
    l3 += l2 >> 26; l2 &= LIMB_MASK;
    l4 += l3 >> 26; l3 &= LIMB_MASK;       // value < 2^127 + 2^5
    l0 += l4 >> 23; l4 &= 0x7fffff;        // second fold: now value <= 2^127 - 1 + small, limbs may ripple once
    l1 += l0 >> 26; l0 &= LIMB_MASK;
    l2 += l1 >> 26; l1 &= LIMB_MASK;
    l3 += l2 >> 26; l2 &= LIMB_MASK;
    l4 += l3 >> 26; l3 &= LIMB_MASK;       // value in [0, 2^127)
    // p itself is the non-canonical zero
    bool is_p = (l0 & l1 & l2 & l3) == LIMB_MASK && l4 == 0x7fffff;
    u32 keep = is_p ? 0u : ~0u;
    l0 &= keep; l1 &= keep; l2 &= keep; l3 &= keep; l4 &= keep;
    lo = (u64)l0 | ((u64)l1 << 26) | ((u64)l2 << 52);
    hi = ((u64)l2 >> 12) | ((u64)l3 << 14) | ((u64)l4 << 40);
}
template <int B> FQ_DEV bool fe_is_zero(const Fe<B>& a) {
    u64 lo, hi; fe_canon(a, lo, hi); return (lo | hi) == 0;
}
template <int A, int B> FQ_DEV bool fe_equal(const Fe<A>& a, const Fe<B>& b) {
    u64 alo, ahi, blo, bhi; fe_canon(a, alo, ahi); fe_canon(b, blo, bhi);
    return alo == blo && ahi == bhi;
}
template <int A, int B> FQ_DEV bool fe2_equal(const Fe2<A>& a, const Fe2<B>& b) {
    const bool re_eq = fe_equal(a.re, b.re), im_eq = fe_equal(a.im, b.im);   // both evaluated: no branch
    return re_eq && im_eq;
}

// x if mask == ~0 else y (mask must be 0 or ~0): the branch-free GFp.select of fields.py:59-64
template <int B> FQ_DEV Fe<B> fe_select(u32 mask, const Fe<B>& x, const Fe<B>& y) {
    Fe<B> r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.l[i] = y.l[i] ^ (mask & (x.l[i] ^ y.l[i]));
    return r;
}
template <int B> FQ_DEV Fe2<B> fe2_select(u32 mask, const Fe2<B>& x, const Fe2<B>& y) {   // fields.py:236-238
    Fe2<B> r; r.re = fe_select(mask, x.re, y.re); r.im = fe_select(mask, x.im, y.im); return r;
}

// the same selection as ONE instruction per limb: gfx950's v_bitop3_b32 evaluates any three-input boolean function, here
// (mask & x) | (~mask & y) (truth table 0xCA).  Written through the builtin because hipcc, given the xor form above for
// a PAIR of selections (x, y) -> (y, x), shares the x ^ y between them and spends three instructions per limb pair instead of two.
template <int B> FQ_DEV Fe2<B> fe2_bitselect(u32 mask, const Fe2<B>& x, const Fe2<B>& y) {
    Fe2<B> r;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        r.re.l[i] = __builtin_amdgcn_bitop3_b32(mask, x.re.l[i], y.re.l[i], 0xCA);
        r.im.l[i] = __builtin_amdgcn_bitop3_b32(mask, x.im.l[i], y.im.l[i], 0xCA);
    }
    return r;
}

// -x if mask == ~0 else x, branch-free in two cheap ops per limb: bias - x == (bias + 1) + ~x (mod 2^32)
template <int B> FQ_DEV Fe<B + 1> fe_cneg(const Fe<B>& x, u32 mask) {
    static_assert((u64)(B + 1) * (LIMB_MASK - 7) >= (u64)B * UNIT, "bias too small");
    Fe<B + 1> r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.l[i] = (x.l[i] ^ mask) + ((bias_limb(B + 1, i) + 1u) & mask);
    return r;
}
template <int B> FQ_DEV Fe2<B + 1> fe2_cneg(const Fe2<B>& x, u32 mask) {
    Fe2<B + 1> r; r.re = fe_cneg(x.re, mask); r.im = fe_cneg(x.im, mask); return r;
}

// ---- inversion: x^(2^127-3) by the reference's fixed chain (fields.py:66-106) -----------------
FQ_DEV Fe<1> fe_sqr_n(Fe<1> x, int n) {
#pragma unroll 1
    for (int i = 0; i < n; i++) x = fe_sqr(x);
    return x;
}
// the chain itself, inlined where the call would cost more than its code: a caller with many live values (normalize_kernel)
// otherwise spills them around the call (656 bytes of scratch per lane at K = 8)
FQ_DEV Fe<1> fe_inv_inline(Fe<1> x) {
    Fe<1> x2 = fe_mul(x, fe_sqr(x));             // 2^2 - 1
    Fe<1> x4 = fe_mul(x2, fe_sqr_n(x2, 2));      // 2^4 - 1
    Fe<1> x8 = fe_mul(x4, fe_sqr_n(x4, 4));      // 2^8 - 1
    Fe<1> x16 = fe_mul(x8, fe_sqr_n(x8, 8));     // 2^16 - 1
    Fe<1> x32 = fe_mul(x16, fe_sqr_n(x16, 16));  // 2^32 - 1
    Fe<1> t = fe_mul(fe_sqr_n(x32, 32), x32);    // 2^64 - 1
    t = fe_mul(fe_sqr_n(t, 32), x32);            // 2^96 - 1
    t = fe_mul(fe_sqr_n(t, 16), x16);            // 2^112 - 1
    t = fe_mul(fe_sqr_n(t, 8), x8);              // 2^120 - 1
    t = fe_mul(fe_sqr_n(t, 4), x4);              // 2^124 - 1
    t = fe_mul(fe_sqr(t), x);                    // 2^125 - 1
    return fe_mul(fe_sqr_n(t, 2), x);            // 2^127 - 3
}
__device__ __noinline__ Fe<1> fe_inv(Fe<1> x) { return fe_inv_inline(x); }
// x^(2^125 - 1) = 1/sqrt(x) for squares (fields.py:108-122); any addition chain gives the same residue
__device__ __noinline__ Fe<1> fe_invsqrt(Fe<1> x) {
    Fe<1> x2 = fe_mul(x, fe_sqr(x));             // 2^2 - 1
    Fe<1> x4 = fe_mul(x2, fe_sqr_n(x2, 2));      // 2^4 - 1
    Fe<1> x5 = fe_mul(x, fe_sqr(x4));            // 2^5 - 1
    Fe<1> acc = x5, cur = x5;
#pragma unroll 1
    for (int i = 0; i < 24; i++) {               // 2^(5(i+2)) - 1, as the reference's loop
        cur = fe_sqr_n(cur, 5);
        acc = fe_mul(cur, acc);
    }
    return acc;                                  // 2^125 - 1
}
FQ_DEV Fe<1> fe_half() { Fe<1> h; h.l[0] = h.l[1] = h.l[2] = h.l[3] = 0; h.l[4] = 1u << 22; return h; }   // 2^126 (fields.py:16)

FQ_DEV Fe2<1> fe2_half_const() { Fe2<1> h; h.re = fe_half(); for (int i = 0; i < 5; i++) h.im.l[i] = 0; return h; }   // 2^126 = 1/2

// conj(a) / (a0^2 + a1^2)                                                       fields.py:193-199
template <int B> FQ_DEV Fe2<1> fe2_inv(const Fe2<B>& a) {
    Fe<1> n = fe_inv(fe_carry(fe_add(fe_sqr(a.re), fe_sqr(a.im))));
    Fe2<1> r;
    r.re = fe_mul(n, a.re);
    r.im = fe_mul(n, fe_neg(a.im));
    return r;
}


// ---- signed flavour (ladder phase only) ---------------------------------------------------------------------
// Inside the ladders the limbs are read as SIGNED 32-bit values with |limb| <= B * UNIT: a subtraction is one
// v_sub_u32 per limb (no bias), products are v_mad_i64_i32 with arithmetic carry shifts -- 5 % fewer instructions
// per ladder step.  Values enter the ladder with non-negative limbs (table entries, the start point) and leave it
// through fe_unsign(), so the two flavours never mix in one operation.  hipcc must not learn that a masked limb is
// non-negative: it would turn that operand's sign extension into a zero extension and then has no single
// instruction for the mixed product (it emits two multiply-adds and fix-ups); FQ_SIGN_UNKNOWN hides it.
#define FQ_SIGN_UNKNOWN(x) asm("" : "+v"(x))
FQ_DEV i64 smul(u32 x, u32 y) { return (i64)(i32)x * (i64)(i32)y; }        // with the accumulation: one v_mad_i64_i32
constexpr bool cols_ok_signed(u64 weighted) {
    return weighted * 5 * 8 <= (((1ull << 63) - (1ull << 41)) / (UNIT * UNIT));
}
template <int B> constexpr bool fits8_signed() { return (u64)8 * B * UNIT < (1ull << 31); }
template <int A, int B> FQ_DEV Fe<A + B> fe_sub_signed(const Fe<A>& a, const Fe<B>& b) {
    static_assert((u64)(A + B) * UNIT < (1ull << 31), "limb overflow");
    Fe<A + B> r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.l[i] = a.l[i] - b.l[i];
    return r;
}
template <int B> FQ_DEV Fe<B> fe_neg_signed(const Fe<B>& b) {
    Fe<B> r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.l[i] = 0u - b.l[i];
    return r;
}
template <int B> FQ_DEV Fe<B> fe_cneg_signed(const Fe<B>& x, u32 mask) {   // -x == ~x + 1
    Fe<B> r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.l[i] = (x.l[i] ^ mask) - mask;
    return r;
}
template <int A, int B> FQ_DEV Fe2<A + B> fe2_sub_signed(const Fe2<A>& a, const Fe2<B>& b) {
    Fe2<A + B> r; r.re = fe_sub_signed(a.re, b.re); r.im = fe_sub_signed(a.im, b.im); return r;
}
template <int B> FQ_DEV Fe2<B> fe2_cneg_signed(const Fe2<B>& x, u32 mask) {
    Fe2<B> r; r.re = fe_cneg_signed(x.re, mask); r.im = fe_cneg_signed(x.im, mask); return r;
}
// five limbs (in [0, 2^26) each) + the signed carry out of column 4 -> bound-1 element (2^130 == 8)
FQ_DEV Fe<1> fe_finish_signed(u32 l0, u32 l1, u32 l2, u32 l3, u32 l4, i64 top) {
    Fe<1> r;
    i64 w = top * 8 + (i64)l0;               // |top| < 2^37 -> |w| < 2^41
    r.l[0] = (u32)w & LIMB_MASK;
    r.l[1] = l1 + (u32)(w >> LIMB_BITS);     // magnitude <= UNIT - 1
    r.l[2] = l2; r.l[3] = l3; r.l[4] = l4;
    FQ_SIGN_UNKNOWN(r.l[0]); FQ_SIGN_UNKNOWN(r.l[2]); FQ_SIGN_UNKNOWN(r.l[3]); FQ_SIGN_UNKNOWN(r.l[4]);
    return r;
}
// The wrap-around factor 8 as (aw, bw) with aw_i * bw_j == 8 * a_i * b_j: normally (a, 8b); when 8b would not fit a
// signed 32-bit operand, (4a, 2b) -- needed only by the squaring of a bound-2 element (DBL's (X+Y)^2).
template <int A, int B> FQ_DEV void wrap_operands_signed(u32 aw[5], u32 bw[5], const Fe<A>& a, const Fe<B>& b) {
    if constexpr (fits8_signed<B>()) {
#pragma unroll
        for (int i = 0; i < 5; i++) { aw[i] = a.l[i]; bw[i] = b.l[i] << 3; }
    } else {
        static_assert((u64)4 * A * UNIT < (1ull << 31) && (u64)2 * B * UNIT < (1ull << 31), "no split of the factor 8 fits");
#pragma unroll
        for (int i = 0; i < 5; i++) { aw[i] = a.l[i] << 2; bw[i] = b.l[i] << 1; }
    }
}
#define FQ_OPAQUE(x) asm("" : "+v"(x))
template <int A, int B> FQ_DEV Fe2<1> fe2_mul_signed(const Fe2<A>& a, const Fe2<B>& b) {
    static_assert(cols_ok_signed((u64)2 * A * B), "column overflow");
    static_assert(fits8_signed<B>(), "8*b does not fit a signed 32-bit operand: swap the operands");
    u32 b0x8[5], b1x8[5];
#pragma unroll
    for (int i = 0; i < 5; i++) { b0x8[i] = b.re.l[i] << 3; b1x8[i] = b.im.l[i] << 3; }
    Fe<A> na1 = fe_neg_signed(a.im);
    i64 re = 0, im = 0;
    u32 lr[5], li[5];
#define FQ_COL2(K)                                                                          \
    _Pragma("unroll") for (int i = 0; i < 5; i++) {                                         \
        const int j = K - i;                                                                \
        const u32 q0 = j >= 0 ? b.re.l[j >= 0 ? j : 0] : b0x8[j >= 0 ? 0 : j + 5];           \
        const u32 q1 = j >= 0 ? b.im.l[j >= 0 ? j : 0] : b1x8[j >= 0 ? 0 : j + 5];           \
        re += smul(a.re.l[i], q0); FQ_OPAQUE(re);                                           \
        im += smul(a.re.l[i], q1); FQ_OPAQUE(im);                                           \
        re += smul(na1.l[i], q1); FQ_OPAQUE(re);                                            \
        im += smul(a.im.l[i], q0); FQ_OPAQUE(im);                                           \
    }                                                                                       \
    lr[K] = (u32)re & LIMB_MASK; re >>= LIMB_BITS;                                          \
    li[K] = (u32)im & LIMB_MASK; im >>= LIMB_BITS;
    FQ_COL2(0) FQ_COL2(1) FQ_COL2(2) FQ_COL2(3) FQ_COL2(4)
#undef FQ_COL2
    Fe2<1> r;
    r.re = fe_finish_signed(lr[0], lr[1], lr[2], lr[3], lr[4], re);
    r.im = fe_finish_signed(li[0], li[1], li[2], li[3], li[4], im);
    return r;
}
template <int A> FQ_DEV Fe2<1> fe2_sqr_signed(const Fe2<A>& a) {
    Fe<2 * A> s = fe_add(a.re, a.im);
    Fe<2 * A> d = fe_sub_signed(a.re, a.im);
    Fe<2 * A> t = fe_dbl(a.re);
    static_assert((u64)2 * A * UNIT < (1ull << 31), "limb overflow");
    static_assert(cols_ok_signed((u64)(2 * A) * (2 * A)), "column overflow");
    u32 dw[5], s8[5], tw[5], i8[5];
    wrap_operands_signed(dw, s8, d, s);
    wrap_operands_signed(tw, i8, t, a.im);
    i64 re = 0, im = 0;
    u32 lr[5], li[5];
#define FQ_COL2(K)                                                                          \
    _Pragma("unroll") for (int i = 0; i < 5; i++) {                                         \
        const int j = K - i;                                                                \
        const u32 q0 = j >= 0 ? s.l[j >= 0 ? j : 0] : s8[j >= 0 ? 0 : j + 5];               \
        const u32 q1 = j >= 0 ? a.im.l[j >= 0 ? j : 0] : i8[j >= 0 ? 0 : j + 5];            \
        re += smul(j >= 0 ? d.l[i] : dw[i], q0); FQ_OPAQUE(re);                             \
        im += smul(j >= 0 ? t.l[i] : tw[i], q1); FQ_OPAQUE(im);                             \
    }                                                                                       \
    lr[K] = (u32)re & LIMB_MASK; re >>= LIMB_BITS;                                          \
    li[K] = (u32)im & LIMB_MASK; im >>= LIMB_BITS;
    FQ_COL2(0) FQ_COL2(1) FQ_COL2(2) FQ_COL2(3) FQ_COL2(4)
#undef FQ_COL2
    Fe2<1> r;
    r.re = fe_finish_signed(lr[0], lr[1], lr[2], lr[3], lr[4], re);
    r.im = fe_finish_signed(li[0], li[1], li[2], li[3], li[4], im);
    return r;
}
#undef FQ_OPAQUE
// signed limbs of bound B -> non-negative bound-1 limbs of the same residue (leaves the signed flavour)
template <int B> FQ_DEV Fe<1> fe_unsign(const Fe<B>& a) {
    static_assert((u64)(B + 1) * (LIMB_MASK - 7) >= (u64)B * UNIT, "bias too small");
    static_assert((u64)(2 * B + 1) * UNIT + (1ull << 20) < (1ull << 32), "limb overflow");
    Fe<2 * B + 1> u;
#pragma unroll
    for (int i = 0; i < 5; i++) u.l[i] = a.l[i] + bias_limb(B + 1, i);
    return fe_carry(u);
}
// the same without the carry: non-negative limbs of bound 2B + 1 -- for a consumer that carries anyway (fe_canon in front of the final
// store: round 6, one carry chain per field element of the result instead of two)
template <int B> FQ_DEV Fe<2 * B + 1> fe_unsign_wide(const Fe<B>& a) {
    static_assert((u64)(B + 1) * (LIMB_MASK - 7) >= (u64)B * UNIT, "bias too small");
    static_assert((u64)(2 * B + 1) * UNIT + (1ull << 20) < (1ull << 32), "limb overflow");
    Fe<2 * B + 1> u;
#pragma unroll
    for (int i = 0; i < 5; i++) u.l[i] = a.l[i] + bias_limb(B + 1, i);
    return u;
}
template <int B> FQ_DEV Fe2<1> fe2_unsign(const Fe2<B>& a) {
    Fe2<1> r; r.re = fe_unsign(a.re); r.im = fe_unsign(a.im); return r;
}

}  // namespace fq
