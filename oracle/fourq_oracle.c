/* TEST INFRASTRUCTURE ONLY -- fast CPU oracle for the FourQ scalar-multiplication path.
 *
 * A plain-C restatement (2 x 64-bit limbs, unsigned __int128) of the reference algorithms in
 * bifurcation/fourq impl/fields.py and impl/curve4q.py; every function cites the reference
 * lines it follows.  It exists so that parity tests can check FULL batches (2^16 .. 2^20
 * elements) of the HIP engine in seconds.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; the product library never links or calls it.
 *
 * Parity status: PINNED -- tests/test_oracle_c.py checks it against the committed golden
 * vectors (reference KATs + reference-generated fixtures) and against the Python oracle.
 *
 * Array layouts (all little-endian u64 words):
 *   fp      2 words            fp2   4 words (re, im)
 *   scalar  4 words            affine 8 words (x, y)
 *   R1     20 words (X,Y,Z,Ta,Tb)     R2 16 words (X+Y, Y-X, 2Z, 2dT)     table 8 x 16 words
 *
 * Build: see oracle/Makefile  (gcc -O2 -fopenmp -shared -fPIC).
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

typedef struct { u64 lo, hi; } fp;          /* canonical, in [0, p), p = 2^127-1 (fields.py:5) */
typedef struct { fp re, im; } fp2;
typedef struct { fp2 X, Y, Z, Ta, Tb; } r1_t;
typedef struct { fp2 N, D, E, F; } r2_t;    /* also used for R3 (X+Y, Y-X, Z, T) */
typedef struct { fp2 X, Y, Z; } r4_t;

#define HI_MASK 0x7fffffffffffffffULL

/* ------------------------------------------------------------------ GF(p)  fields.py:29-57 */
static inline fp fp_norm128(u64 lo, u64 hi) {   /* any 128-bit value -> canonical residue */
    u64 c = hi >> 63;                           /* 2^127 == 1 */
    hi &= HI_MASK;
    u128 s = (u128)lo + c;
    lo = (u64)s; hi += (u64)(s >> 64);
    if (hi >> 63) { hi &= HI_MASK; s = (u128)lo + 1; lo = (u64)s; hi += (u64)(s >> 64); }
    if (lo == ~0ULL && hi == HI_MASK) { lo = 0; hi = 0; }
    fp r = { lo, hi }; return r;
}
static inline fp fp_add(fp a, fp b) {           /* fields.py:30-33 */
    u128 s = (u128)a.lo + b.lo;
    return fp_norm128((u64)s, a.hi + b.hi + (u64)(s >> 64));
}
static inline fp fp_neg(fp a) {                 /* fields.py:54-57 : p - a, with p - 0 -> 0 */
    return fp_norm128(~a.lo, HI_MASK - a.hi);   /* p - a = (all-ones127) - a, no borrow */
}
static inline fp fp_sub(fp a, fp b) { return fp_add(a, fp_neg(b)); }   /* fields.py:36-39 */
static inline fp fp_mul(fp a, fp b) {           /* fields.py:42-45 */
    u128 p00 = (u128)a.lo * b.lo, p01 = (u128)a.lo * b.hi, p10 = (u128)a.hi * b.lo, p11 = (u128)a.hi * b.hi;
    u64 w0 = (u64)p00;
    u128 t = (p00 >> 64) + (u64)p01 + (u64)p10;
    u64 w1 = (u64)t;
    t = (t >> 64) + (p01 >> 64) + (p10 >> 64) + (u64)p11;
    u64 w2 = (u64)t;
    u64 w3 = (u64)(t >> 64) + (u64)(p11 >> 64);
    /* value = L + 2^127 H, L = low 127 bits, H = bits 127..253 */
    u64 l_lo = w0, l_hi = w1 & HI_MASK;
    u64 h_lo = (w1 >> 63) | (w2 << 1), h_hi = (w2 >> 63) | (w3 << 1);
    u128 s = (u128)l_lo + h_lo;
    return fp_norm128((u64)s, l_hi + h_hi + (u64)(s >> 64));
}
static inline fp fp_sqr(fp a) { return fp_mul(a, a); }                 /* fields.py:48-51 */
static fp fp_sqr_n(fp a, int n) { while (n--) a = fp_sqr(a); return a; }
static fp fp_inv(fp x) {                        /* fields.py:67-106 : x^(2^127-3), fixed chain */
    fp x2 = fp_mul(x, fp_sqr(x));
    fp x4 = fp_mul(x2, fp_sqr_n(x2, 2));
    fp x8 = fp_mul(x4, fp_sqr_n(x4, 4));
    fp x16 = fp_mul(x8, fp_sqr_n(x8, 8));
    fp x32 = fp_mul(x16, fp_sqr_n(x16, 16));
    fp t = fp_mul(fp_sqr_n(x32, 32), x32);
    t = fp_mul(fp_sqr_n(t, 32), x32);
    t = fp_mul(fp_sqr_n(t, 16), x16);
    t = fp_mul(fp_sqr_n(t, 8), x8);
    t = fp_mul(fp_sqr_n(t, 4), x4);
    t = fp_mul(fp_sqr(t), x);
    return fp_mul(fp_sqr_n(t, 2), x);
}
static inline int fp_eq(fp a, fp b) { return a.lo == b.lo && a.hi == b.hi; }

/* --------------------------------------------------------------- GF(p^2)  fields.py:156-199 */
static inline fp2 f2_add(fp2 a, fp2 b) { fp2 r = { fp_add(a.re, b.re), fp_add(a.im, b.im) }; return r; }
static inline fp2 f2_sub(fp2 a, fp2 b) { fp2 r = { fp_sub(a.re, b.re), fp_sub(a.im, b.im) }; return r; }
static inline fp2 f2_neg(fp2 a) { fp2 r = { fp_neg(a.re), fp_neg(a.im) }; return r; }
static inline fp2 f2_conj(fp2 a) { fp2 r = { a.re, fp_neg(a.im) }; return r; }
static inline fp2 f2_mul(fp2 a, fp2 b) {        /* fields.py:167-173 */
    fp2 r = { fp_sub(fp_mul(a.re, b.re), fp_mul(a.im, b.im)), fp_add(fp_mul(a.re, b.im), fp_mul(a.im, b.re)) };
    return r;
}
static inline fp2 f2_sqr(fp2 a) {               /* fields.py:176-181 */
    fp t = fp_mul(a.re, a.im);
    fp2 r = { fp_sub(fp_sqr(a.re), fp_sqr(a.im)), fp_add(t, t) };
    return r;
}
static fp2 f2_inv(fp2 a) {                      /* fields.py:194-199 */
    fp n = fp_inv(fp_add(fp_sqr(a.re), fp_sqr(a.im)));
    fp2 r = { fp_mul(n, a.re), fp_mul(n, fp_neg(a.im)) };
    return r;
}
static inline int f2_eq(fp2 a, fp2 b) { return fp_eq(a.re, b.re) && fp_eq(a.im, b.im); }

#define FP(hi_, lo_) { lo_##ULL, hi_##ULL }
static const fp2 F2_ONE = { FP(0x0, 0x1), FP(0x0, 0x0) };
/* curve4q.py:9 */
static const fp2 CURVE_D = { FP(0xe4, 0x0000000000000142), FP(0x5e472f846657e0fc, 0xb3821488f1fc0c8d) };
/* curve4q.py:240-256 */
static const fp2 CTAU = { FP(0x1964de2c3afad20c, 0x74dcd57cebce74c3), FP(0x000000000000000c, 0x0000000000000012) };
static const fp2 CTAUDUAL = { FP(0x4aa740eb23058652, 0x9ecaa6d9decdf034), FP(0x7ffffffffffffff4, 0x0000000000000011) };
static const fp2 CPHI[10] = {
    { FP(0x0000000000000005, 0xfffffffffffffff7), FP(0x2553a0759182c329, 0x4f65536cef66f81a) },
    { FP(0x0000000000000005, 0x0000000000000007), FP(0x62c8caa0c50c62cf, 0x334d90e9e28296f9) },
    { FP(0x000000000000000f, 0x0000000000000015), FP(0x78df262b6c9b5c98, 0x2c2cb7154f1df391) },
    { FP(0x0000000000000002, 0x0000000000000003), FP(0x5084c6491d76342a, 0x92440457a7962ea4) },
    { FP(0x0000000000000003, 0x0000000000000003), FP(0x12440457a7962ea4, 0xa1098c923aec6855) },
    { FP(0x000000000000000a, 0x000000000000000f), FP(0x459195418a18c59e, 0x669b21d3c5052df3) },
    { FP(0x0000000000000012, 0x0000000000000018), FP(0x0b232a8314318b3c, 0xcd3643a78a0a5be7) },
    { FP(0x0000000000000018, 0x0000000000000023), FP(0x3963bc1c99e2ea1a, 0x66c183035f48781a) },
    { FP(0x00000000000000aa, 0x00000000000000f0), FP(0x1f529f860316cbe5, 0x44e251582b5d0ef0) },
    { FP(0x0000000000000870, 0x0000000000000bef), FP(0x0fd52e9cfe00375b, 0x014d3e48976e2505) },
};
static const fp2 CPSI[5] = {
    { FP(0x0, 0x0), FP(0x0, 0x0) },
    { FP(0x2af99e9a83d54a02, 0xedf07f4767e346ef), FP(0x00000000000000de, 0x000000000000013a) },
    { FP(0x00000000000000e4, 0x0000000000000143), FP(0x21b8d07b99a81f03, 0x4c7deb770e03f372) },
    { FP(0x0000000000000006, 0x0000000000000009), FP(0x4cb26f161d7d6906, 0x3a6e6abe75e73a61) },
    { FP(0x7ffffffffffffff9, 0xfffffffffffffff6), FP(0x334d90e9e28296f9, 0xc59195418a18c59e) },
};

/* ------------------------------------------------------------ group law  curve4q.py:100-175 */
static r2_t r1_to_r2(const r1_t *p) {           /* curve4q.py:109-116 */
    fp2 twod = f2_add(CURVE_D, CURVE_D);
    r2_t r = { f2_add(p->X, p->Y), f2_sub(p->Y, p->X), f2_add(p->Z, p->Z), f2_mul(twod, f2_mul(p->Ta, p->Tb)) };
    return r;
}
static r2_t r1_to_r3(const r1_t *p) {           /* curve4q.py:119-126 */
    r2_t r = { f2_add(p->X, p->Y), f2_sub(p->Y, p->X), p->Z, f2_mul(p->Ta, p->Tb) };
    return r;
}
static r1_t r2_to_r4(const r2_t *p) {           /* curve4q.py:129-135 (X,Y,Z only; Ta,Tb unset) */
    r1_t r; memset(&r, 0, sizeof r);
    r.X = f2_sub(p->N, p->D); r.Y = f2_add(p->D, p->N); r.Z = p->E;
    return r;
}
static r2_t r2_neg(const r2_t *p) {             /* curve4q.py:193-195 */
    r2_t r = { p->D, p->N, p->E, f2_neg(p->F) };
    return r;
}
static r1_t dbl(const r1_t *p) {                /* curve4q.py:138-152 */
    fp2 a = f2_sqr(p->X), b = f2_sqr(p->Y), zz = f2_sqr(p->Z);
    fp2 c = f2_add(zz, zz), d = f2_add(a, b);
    fp2 e = f2_sub(f2_sqr(f2_add(p->X, p->Y)), d), f = f2_sub(b, a), g = f2_sub(c, f);
    r1_t r = { f2_mul(e, g), f2_mul(d, f), f2_mul(f, g), e, d };
    return r;
}
static r1_t add_core(const r2_t *p3, const r2_t *q2) {   /* curve4q.py:155-171 */
    fp2 a = f2_mul(p3->D, q2->D), b = f2_mul(p3->N, q2->N);
    fp2 c = f2_mul(q2->F, p3->F), d = f2_mul(q2->E, p3->E);
    fp2 e = f2_sub(b, a), f = f2_sub(d, c), g = f2_add(d, c), h = f2_add(b, a);
    r1_t r = { f2_mul(e, f), f2_mul(g, h), f2_mul(f, g), e, h };
    return r;
}
static r1_t add(const r1_t *p, const r2_t *q2) {          /* curve4q.py:174-175 */
    r2_t p3 = r1_to_r3(p);
    return add_core(&p3, q2);
}

/* ------------------------------------------------------------ endomorphisms  curve4q.py:258-322 */
static r4_t tau(const r1_t *p) {                /* curve4q.py:258-267 */
    fp2 a = f2_sqr(p->X), b = f2_sqr(p->Y), c = f2_add(a, b), d = f2_sub(a, b), zz = f2_sqr(p->Z);
    r4_t r;
    r.X = f2_mul(f2_mul(f2_mul(CTAU, p->X), p->Y), d);
    r.Y = f2_neg(f2_mul(f2_add(f2_add(zz, zz), d), c));
    r.Z = f2_mul(c, d);
    return r;
}
static r1_t tau_dual(const r4_t *p) {           /* curve4q.py:269-280 */
    fp2 a = f2_sqr(p->X), b = f2_sqr(p->Y), c = f2_add(a, b), ta = f2_sub(b, a), zz = f2_sqr(p->Z);
    fp2 d = f2_sub(f2_add(zz, zz), ta);
    fp2 tb = f2_mul(f2_mul(CTAUDUAL, p->X), p->Y);
    r1_t r = { f2_mul(tb, c), f2_mul(ta, d), f2_mul(c, d), ta, tb };
    return r;
}
static r4_t upsilon(const r4_t *p) {            /* curve4q.py:282-302 */
    fp2 A = f2_mul(f2_mul(CPHI[0], p->X), p->Y), B = f2_mul(p->Y, p->Z), C = f2_sqr(p->Y), D = f2_sqr(p->Z);
    fp2 F = f2_sqr(D), G = f2_sqr(B), H = f2_sqr(C), I = f2_mul(CPHI[1], B);
    fp2 J = f2_add(C, f2_mul(CPHI[2], D));
    fp2 K = f2_add(f2_add(f2_mul(CPHI[8], G), H), f2_mul(CPHI[9], F));
    fp2 x2 = f2_mul(f2_add(I, J), f2_sub(I, J));
    fp2 L = f2_add(C, f2_mul(CPHI[4], D)), M = f2_mul(CPHI[3], B);
    fp2 Nn = f2_mul(f2_add(L, M), f2_sub(L, M));
    fp2 y2 = f2_add(f2_add(H, f2_mul(CPHI[6], G)), f2_mul(CPHI[7], F));
    r4_t r;
    r.X = f2_conj(f2_mul(f2_mul(A, K), x2));
    r.Y = f2_conj(f2_mul(f2_mul(f2_mul(CPHI[5], D), Nn), y2));
    r.Z = f2_conj(f2_mul(f2_mul(B, K), Nn));
    return r;
}
static r4_t chi(const r4_t *p) {                /* curve4q.py:304-316 */
    fp2 A = f2_conj(p->X), B = f2_conj(p->Y), C = f2_sqr(f2_conj(p->Z)), D = f2_sqr(A);
    fp2 G = f2_mul(B, f2_add(D, f2_mul(CPSI[2], C)));
    fp2 H = f2_neg(f2_add(D, f2_mul(CPSI[4], C)));
    r4_t r;
    r.X = f2_mul(f2_mul(f2_mul(CPSI[1], A), C), H);
    r.Y = f2_mul(G, f2_add(D, f2_mul(CPSI[3], C)));
    r.Z = f2_mul(G, H);
    return r;
}
static r1_t phi(const r1_t *p) { r4_t t = tau(p); t = upsilon(&t); return tau_dual(&t); }   /* curve4q.py:318 */
static r1_t psi(const r1_t *p) { r4_t t = tau(p); t = chi(&t); return tau_dual(&t); }       /* curve4q.py:321 */

/* ------------------------------------------------------------ tables  curve4q.py:179-185, :385-403 */
static void table_windowed(const r1_t *p, r2_t T[8]) {
    r1_t q = dbl(p);
    T[0] = r1_to_r2(p);
    for (int i = 1; i < 8; i++) { r1_t s = add(&q, &T[i - 1]); T[i] = r1_to_r2(&s); }
}
static void table_endo(const r1_t *p, r2_t T[8]) {
    r1_t q = phi(p), r = psi(p), s = psi(&q);
    r2_t q3 = r1_to_r3(&q), r3 = r1_to_r3(&r), s3 = r1_to_r3(&s);
    r1_t t;
    T[0] = r1_to_r2(p);
    t = add_core(&q3, &T[0]); T[1] = r1_to_r2(&t);
    t = add_core(&r3, &T[0]); T[2] = r1_to_r2(&t);
    t = add_core(&r3, &T[1]); T[3] = r1_to_r2(&t);
    for (int k = 0; k < 4; k++) { t = add_core(&s3, &T[k]); T[4 + k] = r1_to_r2(&t); }
}

/* ------------------------------------------------------------ recoding  curve4q.py:326-380 */
static const u64 ELL[4][4] = {   /* L1..L4, curve4q.py:331-334, little-endian words */
    { 0x259686e09d1a7d4fULL, 0xf75682ace6a6bd66ULL, 0xfc5bb5c5ea2be5dfULL, 0x7ULL },
    { 0xd1ba1d84dd627afbULL, 0x2bd235580f468d8dULL, 0x8fd4b04caa6c0f8aULL, 0x3ULL },
    { 0x9b291a33678c203cULL, 0xc42bd6c965dca902ULL, 0xd038bf8d0bffbaf6ULL, 0x0ULL },
    { 0x12e5666b77e7fdc0ULL, 0x81cbdc3714983d82ULL, 0x1b073877a22d8410ULL, 0x3ULL },
};
static const u64 BASIS[4][4] = { /* b1..b4 (two's complement), curve4q.py:326-329 */
    { 0x0906ff27e0a0a196ULL, (u64)-0x1363e862c22a2da0LL, 0x07426031ecc8030fULL, (u64)-0x084f739986b9e651LL },
    { 0x1d495bea84fcc2d4ULL, (u64)-0x1LL, 0x1ULL, 0x25dbc5bc8dd167d0ULL },
    { 0x17abad1d231f0302ULL, 0x02c4211ae388da51ULL, (u64)-0x2e4d21c98927c49fLL, 0x0a9e6f44c02ecd97ULL },
    { 0x136e340a9108c83fULL, 0x3122df2dc3e0ff32ULL, (u64)-0x068a49f02aa8a9b5LL, (u64)-0x18d5087896de0aeaLL },
};
static u64 mul_shift256(const u64 a[4], const u64 m[4]) {   /* ((a * m) >> 256) mod 2^64 */
    u64 w[8] = { 0 };
    for (int i = 0; i < 4; i++) {
        u64 carry = 0;
        for (int j = 0; j < 4; j++) {
            u128 t = (u128)a[i] * m[j] + w[i + j] + carry;
            w[i + j] = (u64)t; carry = (u64)(t >> 64);
        }
        w[i + 4] = carry;
    }
    return w[4];
}
static void decompose(const u64 m[4], u64 v[4]) {           /* curve4q.py:339-356, mod 2^64 */
    u64 t[4];
    for (int i = 0; i < 4; i++) t[i] = mul_shift256(ELL[i], m);
    u64 ac[4], acp[4];
    for (int i = 0; i < 4; i++) {
        u64 a = (i == 0) ? m[0] : 0;
        for (int j = 0; j < 4; j++) a -= t[j] * BASIS[j][i];
        u64 c = 5 * BASIS[1][i] - 3 * BASIS[2][i] + 2 * BASIS[3][i];
        ac[i] = a + c; acp[i] = a + c + BASIS[3][i];
    }
    u64 mask = (u64)0 - (ac[0] & 1);
    for (int i = 0; i < 4; i++) v[i] = acp[i] ^ (mask & (ac[i] ^ acp[i]));
}
static void recode(const u64 vin[4], uint8_t sgn[65], uint8_t dig[65]) {   /* curve4q.py:358-380 */
    u64 v[4] = { vin[0], vin[1], vin[2], vin[3] };
    for (int i = 0; i < 64; i++) {
        u64 b1 = (i + 1 < 64) ? (v[0] >> (i + 1)) & 1 : 0;
        unsigned d = 0;
        for (int j = 1; j < 4; j++) {
            u64 bj = v[j] & 1;
            d += (unsigned)bj << (j - 1);
            v[j] = (v[j] >> 1) + ((b1 | bj) ^ b1);
        }
        sgn[i] = (uint8_t)b1; dig[i] = (uint8_t)d;
    }
    dig[64] = (uint8_t)(v[1] + 2 * v[2] + 4 * v[3]);
    sgn[64] = 1;
}

static const u64 ORDER_N[4] = { 0x2fb2540ec7768ce7ULL, 0xdfbd004dfe0f7999ULL, 0xf05397829cbc14e5ULL, 0x0029cbc14e5e0a72ULL };

static int ge256(const u64 a[4], const u64 b[4]) {
    for (int i = 3; i >= 0; i--) { if (a[i] != b[i]) return a[i] > b[i]; }
    return 1;
}
static void sub256(u64 a[4], const u64 b[4]) {
    u64 br = 0;
    for (int i = 0; i < 4; i++) { u128 t = (u128)a[i] - b[i] - br; a[i] = (u64)t; br = (u64)(t >> 64) & 1; }
}
static void recode_windowed(const u64 m[4], uint8_t sgn[63], uint8_t ind[63]) {   /* curve4q.py:216-226 */
    u64 r[4] = { m[0], m[1], m[2], m[3] };
    for (int k = 10; k >= 0; k--) {             /* r = m mod N ; N < 2^246 so N << 10 < 2^256 */
        u64 s[4];
        for (int i = 3; i >= 0; i--) s[i] = (ORDER_N[i] << k) | ((k && i) ? ORDER_N[i - 1] >> (64 - k) : 0);
        if (ge256(r, s)) sub256(r, s);
    }
    if ((r[0] & 1) == 0) {                      /* make it odd: r += N */
        u64 c = 0;
        for (int i = 0; i < 4; i++) { u128 t = (u128)r[i] + ORDER_N[i] + c; r[i] = (u64)t; c = (u64)(t >> 64); }
    }
    int dg[63];
    for (int i = 0; i < 63; i++) {
        int d = (int)(r[0] & 31) - 16;
        dg[i] = d;
        /* r = (r - d) / 16 */
        if (d >= 0) { u64 b[4] = { (u64)d, 0, 0, 0 }; sub256(r, b); }
        else { u64 c = (u64)(-d); for (int k = 0; k < 4 && c; k++) { u128 t = (u128)r[k] + c; r[k] = (u64)t; c = (u64)(t >> 64); } }
        for (int k = 0; k < 4; k++) r[k] = (r[k] >> 4) | (k < 3 ? r[k + 1] << 60 : 0);
    }
    dg[62] = (int)r[0];                         /* curve4q.py:223 : what is left (always 1) */
    for (int i = 0; i < 63; i++) {
        int ad = dg[i] < 0 ? -dg[i] : dg[i];
        sgn[i] = dg[i] > 0; ind[i] = (uint8_t)(((ad - 1) / 2) & 7);
    }
}

/* ------------------------------------------------------------ scalar multiplication */
static r2_t pick(const r2_t T[8], unsigned idx, unsigned positive) {   /* selectpt(s, T[i], nT[i]) */
    return positive ? T[idx & 7] : r2_neg(&T[idx & 7]);
}
static r1_t mul_windowed(const u64 m[4], const r2_t T[8]) {           /* curve4q.py:228-235 */
    uint8_t sgn[63], ind[63];
    recode_windowed(m, sgn, ind);
    r2_t s = pick(T, ind[62], sgn[62]);
    r1_t q = r2_to_r4(&s);
    for (int i = 61; i >= 0; i--) {
        q = dbl(&q); q = dbl(&q); q = dbl(&q); q = dbl(&q);
        s = pick(T, ind[i], sgn[i]);
        q = add(&q, &s);
    }
    return q;
}
static r1_t mul_endo(const u64 m[4], const r2_t T[8]) {               /* curve4q.py:432-442 */
    u64 v[4]; uint8_t sgn[65], dig[65];
    decompose(m, v); recode(v, sgn, dig);
    r2_t s = pick(T, dig[64], sgn[64]);
    r1_t q = r2_to_r4(&s);
    for (int i = 63; i >= 0; i--) {
        q = dbl(&q);
        s = pick(T, dig[i], sgn[i]);
        q = add(&q, &s);
    }
    return q;
}

/* ------------------------------------------------------------ DH  curve4q.py:446-462 */
static int on_curve(fp2 x, fp2 y) {             /* curve4q.py:23-29 */
    fp2 x2 = f2_sqr(x), y2 = f2_sqr(y);
    return f2_eq(f2_sub(y2, x2), f2_add(F2_ONE, f2_mul(f2_mul(CURVE_D, x2), y2)));
}
static r1_t cofactor392(fp2 x, fp2 y) {         /* curve4q.py:450-455 */
    r1_t p0 = { x, y, F2_ONE, x, y };
    r2_t t0 = r1_to_r2(&p0);
    r1_t q = dbl(&p0); q = add(&q, &t0);
    q = dbl(&q); q = dbl(&q); q = dbl(&q); q = dbl(&q);
    q = add(&q, &t0);
    q = dbl(&q); q = dbl(&q); q = dbl(&q);
    return q;
}

/* ================================================================== exported batch entry points */
#define EXPORT __attribute__((visibility("default")))

static void load_r1(const u64 *w, r1_t *p) { memcpy(p, w, sizeof *p); }
static void store_r1(u64 *w, const r1_t *p) { memcpy(w, p, sizeof *p); }

EXPORT void fqo_table_windowed(const u64 *p_r1, u64 *table) {
    r1_t p; load_r1(p_r1, &p); r2_t T[8]; table_windowed(&p, T); memcpy(table, T, sizeof T);
}
EXPORT void fqo_table_endo(const u64 *p_r1, u64 *table) {
    r1_t p; load_r1(p_r1, &p); r2_t T[8]; table_endo(&p, T); memcpy(table, T, sizeof T);
}
/* kind: 0 = MUL_endo, 1 = MUL_windowed.  points==NULL -> fixed base with `table` (8x16 words). */
EXPORT void fqo_mul_batch(int kind, const u64 *scalars, const u64 *points_r1, const u64 *table, u64 *out_r1, size_t n) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; i++) {
        r2_t T[8];
        if (points_r1) {
            r1_t p; load_r1(points_r1 + 20 * i, &p);
            if (kind == 0) table_endo(&p, T); else table_windowed(&p, T);
        } else memcpy(T, table, sizeof T);
        r1_t q = kind == 0 ? mul_endo(scalars + 4 * i, T) : mul_windowed(scalars + 4 * i, T);
        store_r1(out_r1 + 20 * i, &q);
    }
}
/* status: 0 ok, 1 "Point not on curve", 2 "DH computation resulted in neutral point". */
EXPORT void fqo_dh_batch(int kind, const u64 *scalars, const u64 *points_affine, const u64 *table, u64 *out_affine,
                         uint8_t *status, size_t n) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; i++) {
        fp2 x, y;
        memcpy(&x, points_affine + 8 * i, sizeof x); memcpy(&y, points_affine + 8 * i + 4, sizeof y);
        memset(out_affine + 8 * i, 0, 64);
        if (!on_curve(x, y)) { status[i] = 1; continue; }
        r1_t q = cofactor392(x, y);
        r2_t T[8];
        if (table) memcpy(T, table, sizeof T);
        else if (kind == 0) table_endo(&q, T); else table_windowed(&q, T);
        q = kind == 0 ? mul_endo(scalars + 4 * i, T) : mul_windowed(scalars + 4 * i, T);
        fp2 zi = f2_inv(q.Z);
        fp2 ax = f2_mul(q.X, zi), ay = f2_mul(q.Y, zi);
        fp2 zero = { { 0, 0 }, { 0, 0 } };
        if (f2_eq(ax, zero) && f2_eq(ay, F2_ONE)) { status[i] = 2; continue; }
        memcpy(out_affine + 8 * i, &ax, sizeof ax); memcpy(out_affine + 8 * i + 4, &ay, sizeof ay);
        status[i] = 0;
    }
}
/* R1toAffine (curve4q.py:103-106) of every row: (X/Z, Y/Z), canonical. */
EXPORT void fqo_r1_to_affine_batch(const u64 *points_r1, u64 *out_affine, size_t n) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; i++) {
        r1_t q; load_r1(points_r1 + 20 * i, &q);
        fp2 zi = f2_inv(q.Z);
        fp2 ax = f2_mul(q.X, zi), ay = f2_mul(q.Y, zi);
        memcpy(out_affine + 8 * i, &ax, sizeof ax); memcpy(out_affine + 8 * i + 4, &ay, sizeof ay);
    }
}
EXPORT void fqo_decompose_batch(const u64 *scalars, u64 *out, size_t n) {
    for (size_t i = 0; i < n; i++) decompose(scalars + 4 * i, out + 4 * i);
}
EXPORT int fqo_num_threads(void) {
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
/* The environment variable OMP_NUM_THREADS is read once, when libgomp is first loaded (numpy / torch usually got
 * there first), so a caller that wants a given thread count sets it through the library. */
EXPORT void fqo_set_num_threads(int n) {
#ifdef _OPENMP
    extern void omp_set_num_threads(int);
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
