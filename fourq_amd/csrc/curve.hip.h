// FourQ group law, endomorphisms and look-up tables on top of fp127.hip.h.
//
// Device counterparts of (bifurcation/fourq, impl/curve4q.py):
//   PointOnCurve :23   AffineToR1 :100   R1toAffine :103   R1toR2 :109   R1toR3 :119   R2toR4 :129
//   DBL :138   ADD_core :155   ADD :174   table_windowed :179   tau :258   tau_dual :269
//   upsilon :282   chi :304   phi :318   psi :321   table_endo :385   the 392-cofactor chain :450-455
//
// Every function evaluates the SAME formula DAG as the reference, so that the un-normalised
// projective outputs (X, Y, Z, Ta, Tb) are the same residues the reference returns (parity level
// L2 of SURVEY.md section 8a28).  Only residue-preserving liberties are taken: 2*x is an addition
// instead of a multiplication by (2,0); the constant 2d is precomputed; values that the reference
// computes but never uses (chi's F) are not computed.
#pragma once
#include <type_traits>
#include "fp127.hip.h"

namespace fq {

// ---- constants (canonical limbs, radix 2^26) ---------------------------------------------------
// Generated from curve4q.py:9, :240-256 by tools/gen_constants.py; tests/test_gpu_primitives.py
// checks every one of them through the primitive ABI.
#include "constants.inc"

// ---- point representations --------------------------------------------------------------------
struct R1 {                  // (X, Y, Z, Ta, Tb), x = X/Z, y = Y/Z, T = Ta*Tb = XY/Z
    Fe2<1> X, Y, Z;
    Fe2<4> Ta;               // lazily bounded: DBL returns E (bound 4), ADD returns E (bound 3)
    Fe2<2> Tb;
};
struct R2 {                  // (X+Y, Y-X, 2Z, 2dT): a table entry, stored tight
    Fe2<1> N, D, E, F;
};
struct R2s {                 // a table entry after the sign selection (F possibly negated)
    Fe2<1> N, D, E;
    Fe2<2> F;
};
struct R3 {                  // (X+Y, Y-X, Z, T)
    Fe2<2> N;
    Fe2<3> D;
    Fe2<1> E, F;
};
template <int BX, int BY, int BZ> struct Proj {   // (X, Y, Z) on E or on the isogenous curve
    Fe2<BX> X;
    Fe2<BY> Y;
    Fe2<BZ> Z;
};

FQ_DEV R2s as_signed(const R2& t) {
    R2s r; r.N = t.N; r.D = t.D; r.E = t.E; r.F = widen<2>(t.F); return r;
}
// selectpt(s, T[i], nT[i]) with nT = R2neg(T) = (D, N, E, -F)         curve4q.py:193-206
// neg_mask = ~0 selects the negated entry, 0 the entry itself; branch-free.
FQ_DEV R2s r2_apply_sign(const R2& t, u32 neg_mask) {
    R2s r;
    r.N = fe2_select(neg_mask, t.D, t.N);
    r.D = fe2_select(neg_mask, t.N, t.D);
    r.E = t.E;
    r.F = fe2_select(neg_mask, fe2_neg(t.F), widen<2>(t.F));
    return r;
}

FQ_DEV R1 affine_to_r1(const Fe2<1>& x, const Fe2<1>& y) {            // curve4q.py:100-101
    R1 p; p.X = x; p.Y = y; p.Z = fe2_one(); p.Ta = widen<4>(x); p.Tb = widen<2>(y); return p;
}
FQ_DEV R2 r1_to_r2(const R1& p) {                                      // curve4q.py:109-116
    R2 r;
    r.N = fe2_carry(fe2_add(p.X, p.Y));
    r.D = fe2_carry(fe2_sub(p.Y, p.X));
    r.E = fe2_carry(fe2_dbl(p.Z));
    r.F = fe2_mul(fe2_mul(p.Ta, p.Tb), fe2_two_d());
    return r;
}
FQ_DEV R3 r1_to_r3(const R1& p) {                                      // curve4q.py:119-126
    R3 r;
    r.N = fe2_add(p.X, p.Y);
    r.D = fe2_sub(p.Y, p.X);
    r.E = p.Z;
    r.F = fe2_mul(p.Ta, p.Tb);
    return r;
}
FQ_DEV Proj<1, 1, 1> r2_to_r4(const R2s& t) {                          // curve4q.py:129-135 (the code, not the docstring)
    Proj<1, 1, 1> r;
    r.X = fe2_carry(fe2_sub(t.N, t.D));
    r.Y = fe2_carry(fe2_add(t.D, t.N));
    r.Z = t.E;
    return r;
}

// R1/R4 -> R1                                                          curve4q.py:138-152
// CH: 0 plain, 1 chained carries, 2 chained carries on signed limbs (fp127.hip.h; the ladders only)
template <int CH = (FQ_CHAIN != 0) ? 1 : 0> FQ_DEV R1 dbl(const Fe2<1>& X, const Fe2<1>& Y, const Fe2<1>& Z) {
    Fe2<1> A = fe2_sqrx<CH>(X);
    Fe2<1> B = fe2_sqrx<CH>(Y);
    Fe2<2> C = fe2_dbl(fe2_sqrx<CH>(Z));
    Fe2<2> D = fe2_add(A, B);
    if constexpr (CH == 2) {       // signed: differences carry no bias; G (bound 4) must be a FIRST operand (8*b fits 32 bits signed)
        Fe2<3> E = fe2_sub_signed(fe2_sqrx<CH>(fe2_add(X, Y)), D);
        Fe2<2> F = fe2_sub_signed(B, A);
        Fe2<4> G = fe2_sub_signed(C, F);
        R1 r;
        r.X = fe2_mulx<CH>(G, E);
        r.Z = fe2_mulx<CH>(G, F);
        r.Y = fe2_mulx<CH>(D, F);
        r.Ta = widen<4>(E);
        r.Tb = D;
        return r;
    } else {
        Fe2<4> E = fe2_sub(fe2_sqrx<CH>(fe2_add(X, Y)), D);
        Fe2<3> F = fe2_sub(B, A);
        Fe2<6> G = fe2_sub(C, F);
        R1 r;                  // operand roles chosen so that 8*G (second operand) and -F.im (first) are shared
        r.X = fe2_mulx<CH>(E, G);
        r.Z = fe2_mulx<CH>(F, G);
        r.Y = fe2_mulx<CH>(F, D);
        r.Ta = E;
        r.Tb = D;
        return r;
    }
}
FQ_DEV R1 dbl(const R1& p) { return dbl(p.X, p.Y, p.Z); }

// R3 + R2 -> R1                                                        curve4q.py:155-171
FQ_DEV R1 add_core(const R3& p, const R2s& q) {
    Fe2<1> A = fe2_mul(p.D, q.D);
    Fe2<1> B = fe2_mul(p.N, q.N);
    Fe2<1> C = fe2_mul(q.F, p.F);
    Fe2<1> D = fe2_mul(q.E, p.E);
    Fe2<3> E = fe2_sub(B, A);
    Fe2<3> F = fe2_sub(D, C);
    Fe2<2> G = fe2_add(D, C);
    Fe2<2> H = fe2_add(B, A);
    R1 r;
    r.X = fe2_mul(E, F);
    r.Y = fe2_mul(G, H);
    r.Z = fe2_mul(F, G);
    r.Ta = widen<4>(E);
    r.Tb = H;
    return r;
}
FQ_DEV R1 add(const R1& p, const R2s& q) { return add_core(r1_to_r3(p), q); }   // curve4q.py:174-175

// ---- endomorphisms ---------------------------------------------------------------------------
FQ_DEV Proj<1, 2, 1> tau(const Fe2<1>& X, const Fe2<1>& Y, const Fe2<1>& Z) {   // curve4q.py:258-267
    Fe2<1> A = fe2_sqr(X);
    Fe2<1> B = fe2_sqr(Y);
    Fe2<2> C = fe2_add(A, B);
    Fe2<3> D = fe2_sub(A, B);
    Proj<1, 2, 1> r;
    r.X = fe2_mul(D, fe2_mul(Y, fe2_mul(X, c_tau())));
    r.Y = fe2_neg(fe2_mul(fe2_add(fe2_dbl(fe2_sqr(Z)), D), C));
    r.Z = fe2_mul(D, C);
    return r;
}
FQ_DEV R1 tau_dual(const Fe2<2>& X, const Fe2<2>& Y, const Fe2<2>& Z) {         // curve4q.py:269-280
    Fe2<1> A = fe2_sqr(X);
    Fe2<1> B = fe2_sqr(Y);
    Fe2<2> C = fe2_add(A, B);
    Fe2<3> Ta = fe2_sub(B, A);
    Fe2<6> D = fe2_sub(fe2_dbl(fe2_sqr(Z)), Ta);
    Fe2<1> Tb = fe2_mul(Y, fe2_mul(X, c_taudual()));
    R1 r;
    r.X = fe2_mul(C, Tb);
    r.Y = fe2_mul(D, Ta);
    r.Z = fe2_mul(D, C);
    r.Ta = widen<4>(Ta);
    r.Tb = widen<2>(Tb);
    return r;
}
FQ_DEV Proj<2, 2, 2> upsilon(const Proj<1, 2, 1>& p) {                          // curve4q.py:282-302
    Fe2<1> A = fe2_mul(p.Y, fe2_mul(p.X, c_phi<0>()));
    Fe2<1> B = fe2_mul(p.Y, p.Z);
    Fe2<1> C = fe2_sqr(p.Y);
    Fe2<1> D = fe2_sqr(p.Z);
    Fe2<1> F = fe2_sqr(D);
    Fe2<1> G = fe2_sqr(B);
    Fe2<1> H = fe2_sqr(C);
    Fe2<1> I = fe2_mul(B, c_phi<1>());
    Fe2<2> J = fe2_add(C, fe2_mul(D, c_phi<2>()));
    Fe2<3> K = fe2_add(fe2_add(fe2_mul(G, c_phi<8>()), H), fe2_mul(F, c_phi<9>()));
    Fe2<1> x2 = fe2_mul(fe2_sub(I, J), fe2_add(I, J));
    Fe2<2> L = fe2_add(C, fe2_mul(D, c_phi<4>()));
    Fe2<1> M = fe2_mul(B, c_phi<3>());
    Fe2<1> Nn = fe2_mul(fe2_sub(L, M), fe2_add(L, M));
    Fe2<3> y2 = fe2_add(fe2_add(H, fe2_mul(G, c_phi<6>())), fe2_mul(F, c_phi<7>()));
    Proj<2, 2, 2> r;
    r.X = fe2_conj(fe2_mul(fe2_mul(K, A), x2));
    r.Y = fe2_conj(fe2_mul(y2, fe2_mul(fe2_mul(D, c_phi<5>()), Nn)));
    r.Z = fe2_conj(fe2_mul(fe2_mul(K, B), Nn));
    return r;
}
FQ_DEV Proj<1, 1, 1> chi(const Proj<1, 2, 1>& p) {                              // curve4q.py:304-316
    Fe2<2> A = fe2_conj(p.X);
    Fe2<3> B = fe2_conj(p.Y);
    Fe2<1> C = fe2_sqr(fe2_conj(p.Z));
    Fe2<1> D = fe2_sqr(A);
    Fe2<1> G = fe2_mul(B, fe2_add(D, fe2_mul(C, c_psi<2>())));
    Fe2<3> H = fe2_neg(fe2_add(D, fe2_mul(C, c_psi<4>())));
    Proj<1, 1, 1> r;
    r.X = fe2_mul(H, fe2_mul(fe2_mul(A, c_psi<1>()), C));
    r.Y = fe2_mul(fe2_add(D, fe2_mul(C, c_psi<3>())), G);
    r.Z = fe2_mul(H, G);
    return r;
}
FQ_DEV R1 phi(const R1& p) {                                                    // curve4q.py:318-319
    Proj<2, 2, 2> u = upsilon(tau(p.X, p.Y, p.Z));
    return tau_dual(u.X, u.Y, u.Z);
}
FQ_DEV R1 psi(const R1& p) {                                                    // curve4q.py:321-322
    Proj<1, 1, 1> c = chi(tau(p.X, p.Y, p.Z));
    return tau_dual(widen<2>(c.X), widen<2>(c.Y), widen<2>(c.Z));
}

// ---- membership, cofactor clearing, normalisation (the DH wrapper) ---------------------------
FQ_DEV bool point_on_curve(const Fe2<1>& x, const Fe2<1>& y) {                  // curve4q.py:23-29
    Fe2<1> x2 = fe2_sqr(x), y2 = fe2_sqr(y);
    Fe2<3> lhs = fe2_sub(y2, x2);
    Fe2<2> rhs = fe2_add(fe2_one(), fe2_mul(y2, fe2_mul(x2, c_d())));
    return fe2_equal(lhs, rhs);
}
FQ_DEV R1 clear_cofactor_392(const Fe2<1>& x, const Fe2<1>& y) {                // curve4q.py:450-455
    R1 p0 = affine_to_r1(x, y);
    R2s t0 = as_signed(r1_to_r2(p0));
    R1 q = add(dbl(p0), t0);                  // 3P
#pragma unroll 1
    for (int i = 0; i < 4; i++) q = dbl(q);   // 48P
    q = add(q, t0);                           // 49P
#pragma unroll 1
    for (int i = 0; i < 3; i++) q = dbl(q);   // 392P
    return q;
}
FQ_DEV void r1_to_affine(const R1& p, Fe2<1>& x, Fe2<1>& y) {                   // curve4q.py:103-106
    Fe2<1> zi = fe2_inv(p.Z);
    x = fe2_mul(p.X, zi);
    y = fe2_mul(p.Y, zi);
}

template <int B> FQ_DEV void store_fe2_words(u64* w, const Fe2<B>& a) {   // canonical 4 x u64
    fe_canon(a.re, w[0], w[1]); fe_canon(a.im, w[2], w[3]);
}
// ---- point compression -----------------------------------------------
// sign(X) of curve4q.py:33-39 on canonical words: top bit (bit 126) of x0, or of x1 when x0 == 0.
FQ_DEV u32 fe2_sign(const u64 w[4]) {
    return (w[0] | w[1]) ? (u32)(w[1] >> 62) & 1 : (u32)(w[3] >> 62) & 1;
}
// encode(X, Y) of curve4q.py:41-46: y0 | y1 little-endian, sign(x) in the top bit of the last byte.
FQ_DEV void point_encode(const Fe2<1>& x, const Fe2<1>& y, u64 out[4]) {
    u64 xw[4];
    store_fe2_words(xw, x);
    store_fe2_words(out, y);
    out[3] |= (u64)fe2_sign(xw) << 63;
}
constexpr int DECODE_OK = 0, DECODE_RESERVED = 1, DECODE_NOT_ON_CURVE = 2, DECODE_REF_ATTRIBUTE_ERROR = 3;
// decode(B) of curve4q.py:49-96.  Returns a DECODE_* status; (x, y) valid when DECODE_OK.
//   RESERVED            "Malformed point: reserved bit is not zero" (:53, :62)
//   NOT_ON_CURVE        "Point not on curve" (:94)
//   REF_ATTRIBUTE_ERROR the reference's t == 0 branch (:76-77) names a GFp.two that does not exist and raises
//                       AttributeError (e.g. for the encoding of the neutral point); reported, not "fixed"
FQ_DEV int point_decode(const u64 in[4], Fe2<1>& x, Fe2<1>& y) {
    const u64 HI = 0x7fffffffffffffffull;
    int st = DECODE_OK;
    if (in[1] >> 63) st = DECODE_RESERVED;
    const u32 s = (u32)(in[3] >> 63);
    const u64 y1hi = in[3] & HI;
    if ((in[0] == ~0ull && in[1] == HI) || (in[2] == ~0ull && y1hi == HI)) st = DECODE_RESERVED;   // y0 >= p or y1 >= p
    y.re = fe_unpack(in[0], in[1] & HI);
    y.im = fe_unpack(in[2], y1hi);
    Fe2<1> y2 = fe2_sqr(y);
    Fe2<3> u = fe2_sub(y2, fe2_one());
    Fe2<2> v = fe2_add(fe2_mul(y2, c_d()), fe2_one());
    Fe<1> u0 = fe_carry(u.re), u1 = fe_carry(u.im), v0 = fe_carry(v.re), v1 = fe_carry(v.im);
    Fe<1> t0 = fe_carry(fe_add(fe_mul(u0, v0), fe_mul(u1, v1)));
    Fe<1> t1 = fe_carry(fe_sub(fe_mul(u1, v0), fe_mul(u0, v1)));
    Fe<1> t2 = fe_carry(fe_add(fe_sqr(v0), fe_sqr(v1)));
    Fe<1> t3 = fe_carry(fe_add(fe_sqr(t0), fe_sqr(t1)));
    t3 = fe_mul(fe_invsqrt(t3), t3);
    Fe<1> t = fe_carry(fe_dbl(fe_add(t0, t3)));
    if (fe_is_zero(t) && st == DECODE_OK) st = DECODE_REF_ATTRIBUTE_ERROR;
    Fe<1> a = fe_invsqrt(fe_mul(t, fe_mul(t2, fe_sqr(t2))));
    Fe<1> at2 = fe_mul(a, t2);
    Fe<1> b = fe_mul(at2, t);
    Fe<1> x0 = fe_mul(b, fe_half());
    Fe<1> x1 = fe_mul(at2, t1);
    const u32 swap = fe_equal(t, fe_mul(t2, fe_sqr(b))) ? 0u : ~0u;
    x.re = fe_select(swap, x1, x0);
    x.im = fe_select(swap, x0, x1);
    u64 xw[4];
    store_fe2_words(xw, x);
    x = fe2_carry(fe2_cneg(x, fe2_sign(xw) != s ? ~0u : 0u));
    const u32 flip = point_on_curve(x, y) ? 0u : ~0u;                      // second candidate: conj(x)
    x.im = fe_carry(fe_cneg(x.im, flip));
    if (flip && !point_on_curve(x, y) && st == DECODE_OK) st = DECODE_NOT_ON_CURVE;
    return st;
}

// ---- memory forms ------------------------------------------------------------------------------
// C-ABI form: 128-bit little-endian words (fields packed, canonical on output).
FQ_DEV Fe2<1> load_fe2(const u64* w) {
    Fe2<1> r; r.re = fe_unpack(w[0], w[1]); r.im = fe_unpack(w[2], w[3]); return r;
}
template <int B> FQ_DEV void store_fe2(u64* w, const Fe2<B>& a) {
    fe_canon(a.re, w[0], w[1]); fe_canon(a.im, w[2], w[3]);
}
FQ_DEV R1 load_r1(const u64* w) {
    R1 p;
    p.X = load_fe2(w); p.Y = load_fe2(w + 4); p.Z = load_fe2(w + 8);
    p.Ta = widen<4>(load_fe2(w + 12)); p.Tb = widen<2>(load_fe2(w + 16));
    return p;
}
FQ_DEV void store_r1(u64* w, const R1& p) {
    store_fe2(w, p.X); store_fe2(w + 4, p.Y); store_fe2(w + 8, p.Z); store_fe2(w + 12, p.Ta); store_fe2(w + 16, p.Tb);
}
FQ_DEV R2 load_r2_packed(const u64* w) {
    R2 t; t.N = load_fe2(w); t.D = load_fe2(w + 4); t.E = load_fe2(w + 8); t.F = load_fe2(w + 12); return t;
}
FQ_DEV void store_r2_packed(u64* w, const R2& t) {
    store_fe2(w, t.N); store_fe2(w + 4, t.D); store_fe2(w + 8, t.E); store_fe2(w + 12, t.F);
}

// Working form of a table entry in HBM scratch or LDS: four coordinates (N, D, E, F) of 12 dwords
// each (10 tight limbs re|im + 2 pad, so that every coordinate is three aligned 16-byte accesses).
constexpr int COORD_U32 = 12;
constexpr int R2_LIMBS = 4 * COORD_U32;
template <typename P> FQ_DEV Fe2<1> load_fe2_limbs(const P* src) {   // src: 16-byte aligned u32 pointer
    const uint4* q = reinterpret_cast<const uint4*>(src);
    uint4 a = q[0], b = q[1], c = q[2];
    Fe2<1> r;
    r.re.l[0] = a.x; r.re.l[1] = a.y; r.re.l[2] = a.z; r.re.l[3] = a.w; r.re.l[4] = b.x;
    r.im.l[0] = b.y; r.im.l[1] = b.z; r.im.l[2] = b.w; r.im.l[3] = c.x; r.im.l[4] = c.y;
    return r;
}
template <typename P> FQ_DEV void store_fe2_limbs(P* dst, const Fe2<1>& v) {
    uint4* q = reinterpret_cast<uint4*>(dst);
    q[0] = make_uint4(v.re.l[0], v.re.l[1], v.re.l[2], v.re.l[3]);
    q[1] = make_uint4(v.re.l[4], v.im.l[0], v.im.l[1], v.im.l[2]);
    q[2] = make_uint4(v.im.l[3], v.im.l[4], 0u, 0u);
}
template <typename P> FQ_DEV R2 load_r2_limbs(const P* src) {
    R2 t;
    t.N = load_fe2_limbs(src); t.D = load_fe2_limbs(src + COORD_U32);
    t.E = load_fe2_limbs(src + 2 * COORD_U32); t.F = load_fe2_limbs(src + 3 * COORD_U32);
    return t;
}
template <typename P> FQ_DEV void store_r2_limbs(P* dst, const R2& t) {
    store_fe2_limbs(dst, t.N); store_fe2_limbs(dst + COORD_U32, t.D);
    store_fe2_limbs(dst + 2 * COORD_U32, t.E); store_fe2_limbs(dst + 3 * COORD_U32, t.F);
}

// Slot layouts of a per-lane table in HBM scratch.  A ladder step gathers one entry per lane, so the entry's size in
// 64-byte memory sectors is the traffic of the step:
//   LimbSlots    4 coordinates x 48 bytes, ready-to-use limbs: 192 bytes = 3 sectors (also the LDS layout)
//   PackedSlots  4 coordinates x 32 bytes, each GF(p) element as one 128-bit word (value < 2^128, not necessarily
//                canonical): 128 bytes = 2 sectors, at the price of 8 cheap ALU ops per element on every load.
// (160 bytes of bare limbs would still straddle 3 sectors.)  PARK_P / PARK_Q: where table_endo parks its working points.
struct LimbSlots {
    static constexpr int COORD = COORD_U32, ENTRY = R2_LIMBS, PARK_P = 8 * R2_LIMBS, PARK_Q = 8 * R2_LIMBS + 40, SLOT = 464;
    template <typename P> static FQ_DEV Fe2<1> load(const P* src) { return load_fe2_limbs(src); }
    template <typename P> static FQ_DEV void store(P* dst, const Fe2<1>& v) { store_fe2_limbs(dst, v); }
};
// N and D only (the fused kernels keep E and F in LDS, kernels.hip.h LdsEF, and park table_endo's working values there too):
// the twenty limbs of an entry back to back -- N.re, N.im, D.re, D.im -- 80 bytes = five aligned 16-byte words, 640-byte slots: the 8 192
// lanes of an XCD keep 5.2 MB of entries against its 4 MiB of L2 (round 3's 96-byte entries, each coordinate padded to 48 bytes for its
// own aligned loads: 6.3 MB, six loads per gather).  An entry is loaded and stored WHOLE (load_nd / store_nd below); there is no
// per-coordinate access to this layout.  PARK_P / PARK_Q exist for the builders that park in the slot (none does with this layout).
// FQ_ND_PACKED=1 (experiment, profiles/r06_gather_experiments.txt): the same slots with N and D as four 128-bit canonical words -- 64 bytes
// per entry instead of 80, a fifth less gather traffic, for four packs per stored entry and four unpacks per ladder step.
#ifndef FQ_ND_PACKED
#define FQ_ND_PACKED 0
#endif
struct NDSlots {
    static constexpr int COORD = FQ_ND_PACKED ? 8 : 10, ENTRY = FQ_ND_PACKED ? 16 : 20, PARK_P = 0, PARK_Q = 0, SLOT = 8 * ENTRY;
};
// tight limbs (after fe_carry) -> one 128-bit word: fold bits >= 127 (2^127 == 1), ripple once, concatenate
FQ_DEV uint4 fe_pack128(const Fe<1>& a) {
    u32 l0 = a.l[0] + (a.l[4] >> 23), l1 = a.l[1], l2 = a.l[2], l3 = a.l[3], l4 = a.l[4] & 0x7fffff;
    l1 += l0 >> LIMB_BITS; l0 &= LIMB_MASK;
    l2 += l1 >> LIMB_BITS; l1 &= LIMB_MASK;
    l3 += l2 >> LIMB_BITS; l2 &= LIMB_MASK;
    l4 += l3 >> LIMB_BITS; l3 &= LIMB_MASK;              // l4 <= 2^23: the value is < 2^128
    return make_uint4(l0 | (l1 << 26), (l1 >> 6) | (l2 << 20), (l2 >> 12) | (l3 << 14), (l3 >> 18) | (l4 << 8));
}
FQ_DEV Fe<1> fe_unpack128(const uint4& w) {
    Fe<1> r;
    r.l[0] = w.x & LIMB_MASK;
    r.l[1] = __builtin_amdgcn_alignbit(w.y, w.x, 26) & LIMB_MASK;
    r.l[2] = __builtin_amdgcn_alignbit(w.z, w.y, 20) & LIMB_MASK;
    r.l[3] = __builtin_amdgcn_alignbit(w.w, w.z, 14) & LIMB_MASK;
    r.l[4] = w.w >> 8;                                   // 24 bits
    return r;
}
struct PackedSlots {
    static constexpr int COORD = 8, ENTRY = 32, PARK_P = 8 * 32, PARK_Q = 8 * 32 + 32, SLOT = 320;
    template <typename P> static FQ_DEV Fe2<1> load(const P* src) {
        const uint4* q = reinterpret_cast<const uint4*>(src);
        Fe2<1> r; r.re = fe_unpack128(q[0]); r.im = fe_unpack128(q[1]); return r;
    }
    template <typename P> static FQ_DEV void store(P* dst, const Fe2<1>& v) {
        uint4* q = reinterpret_cast<uint4*>(dst);
        q[0] = fe_pack128(v.re); q[1] = fe_pack128(v.im);
    }
};
template <typename L, typename P> FQ_DEV R2 load_r2(const P* src) {
    R2 t;
    t.N = L::load(src); t.D = L::load(src + L::COORD); t.E = L::load(src + 2 * L::COORD); t.F = L::load(src + 3 * L::COORD);
    return t;
}
template <typename L, typename P> FQ_DEV void store_r2(P* dst, const R2& t) {
    L::store(dst, t.N); L::store(dst + L::COORD, t.D); L::store(dst + 2 * L::COORD, t.E); L::store(dst + 3 * L::COORD, t.F);
}

// N and D of one entry, whatever the layout: coordinate by coordinate where the layout has coordinates, five 16-byte words for NDSlots
template <typename L, typename P> FQ_DEV void load_nd(const P* entry, Fe2<1>& N, Fe2<1>& D) {
    if constexpr (std::is_same<L, NDSlots>::value && FQ_ND_PACKED) {
        const uint4* q = reinterpret_cast<const uint4*>(entry);
        const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
        N.re = fe_unpack128(a); N.im = fe_unpack128(b); D.re = fe_unpack128(c); D.im = fe_unpack128(d);
    } else if constexpr (std::is_same<L, NDSlots>::value) {
        const uint4* q = reinterpret_cast<const uint4*>(entry);
        const uint4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4];
        N.re.l[0] = a.x; N.re.l[1] = a.y; N.re.l[2] = a.z; N.re.l[3] = a.w; N.re.l[4] = b.x;
        N.im.l[0] = b.y; N.im.l[1] = b.z; N.im.l[2] = b.w; N.im.l[3] = c.x; N.im.l[4] = c.y;
        D.re.l[0] = c.z; D.re.l[1] = c.w; D.re.l[2] = d.x; D.re.l[3] = d.y; D.re.l[4] = d.z;
        D.im.l[0] = d.w; D.im.l[1] = e.x; D.im.l[2] = e.y; D.im.l[3] = e.z; D.im.l[4] = e.w;
    } else {
        N = L::load(entry); D = L::load(entry + L::COORD);
    }
}
template <typename L, typename P> FQ_DEV void store_nd(P* entry, const Fe2<1>& N, const Fe2<1>& D) {
    if constexpr (std::is_same<L, NDSlots>::value && FQ_ND_PACKED) {
        uint4* q = reinterpret_cast<uint4*>(entry);
        q[0] = fe_pack128(N.re); q[1] = fe_pack128(N.im); q[2] = fe_pack128(D.re); q[3] = fe_pack128(D.im);
    } else if constexpr (std::is_same<L, NDSlots>::value) {
        uint4* q = reinterpret_cast<uint4*>(entry);
        q[0] = make_uint4(N.re.l[0], N.re.l[1], N.re.l[2], N.re.l[3]);
        q[1] = make_uint4(N.re.l[4], N.im.l[0], N.im.l[1], N.im.l[2]);
        q[2] = make_uint4(N.im.l[3], N.im.l[4], D.re.l[0], D.re.l[1]);
        q[3] = make_uint4(D.re.l[2], D.re.l[3], D.re.l[4], D.im.l[0]);
        q[4] = make_uint4(D.im.l[1], D.im.l[2], D.im.l[3], D.im.l[4]);
    } else {
        L::store(entry, N); L::store(entry + L::COORD, D);
    }
}

// N and D of +-T for a table entry T: R2neg(T) = (D, N, E, -F) (curve4q.py:193-206).  Both coordinates are read from
// their own addresses whatever the sign is and exchanged by masked selects -- GFp2.select of fields.py:236-238, one
// v_bitop3_b32 per limb -- exactly as the reference's selectpt does: the sign of a digit never becomes an address.
// (Round 2 chose by address; the A/B of the two is profiles/r03_sign_select.txt, the switch tools/experiments/r03_sign_by_address.patch.)
template <typename L, typename P> FQ_DEV void load_signed_nd(const P* entry, u32 neg_mask, Fe2<1>& N, Fe2<1>& D) {
    Fe2<1> n, d;
    load_nd<L>(entry, n, d);
    N = fe2_bitselect(neg_mask, d, n);
    D = fe2_bitselect(neg_mask, n, d);
}
template <int CH, int A, int B> FQ_DEV auto fe2_subx(const Fe2<A>& a, const Fe2<B>& b) {
    if constexpr (CH == 2) return widen<A + B + 1>(fe2_sub_signed(a, b)); else return fe2_sub(a, b);
}
template <int CH, int B> FQ_DEV auto fe2_cnegx(const Fe2<B>& x, u32 mask) {
    if constexpr (CH == 2) return widen<B + 1>(fe2_cneg_signed(x, mask)); else return fe2_cneg(x, mask);
}
// Q + (+-T) for a table entry T read coordinate by coordinate from `entry` (HBM scratch or LDS):
// ADD(Q, selectpt(s, T, R2neg(T))) of curve4q.py:232-233, :440 with R2neg(T) = (D, N, E, -F).  The N/D swap
// of the negated entry is a masked select (load_signed_nd), -F is a two-op conditional negation, and E, F are
// loaded just before the products that consume them, which keeps the live set near 110 VGPRs.
template <int CH = (FQ_CHAIN != 0) ? 1 : 0, typename L = LimbSlots, typename P> FQ_DEV R1 add_table(const R1& q, const P* entry, u32 neg_mask) {
    Fe2<1> T = fe2_mulx<CH>(q.Ta, q.Tb);                          // R1toR3: curve4q.py:119-126
    Fe2<2> N1 = fe2_add(q.X, q.Y);
    Fe2<3> D1 = fe2_subx<CH>(q.Y, q.X);
    Fe2<1> tN, tD;
    load_signed_nd<L>(entry, neg_mask, tN, tD);
    Fe2<1> A = fe2_mulx<CH>(D1, tD);                               // ADD_core: curve4q.py:155-171
    Fe2<1> B = fe2_mulx<CH>(N1, tN);
    Fe2<1> C = fe2_mulx<CH>(fe2_cnegx<CH>(L::load(entry + 3 * L::COORD), neg_mask), T);
    Fe2<1> D = fe2_mulx<CH>(L::load(entry + 2 * L::COORD), q.Z);
    Fe2<3> E = fe2_subx<CH>(B, A);
    Fe2<3> F = fe2_subx<CH>(D, C);
    Fe2<2> G = fe2_add(D, C);
    Fe2<2> H = fe2_add(B, A);
    R1 r;                      // 8*F shared by X and Z, -G.im shared by Z and Y
    r.X = fe2_mulx<CH>(E, F);
    r.Z = fe2_mulx<CH>(G, F);
    r.Y = fe2_mulx<CH>(G, H);
    r.Ta = widen<4>(E);
    r.Tb = H;
    return r;
}
// The same addition with the entry's coordinates already in registers (as loaded: N, D not yet exchanged and F not yet
// negated for a negated entry): lets a lone wave issue its HBM gathers a whole doubling ahead of their use.
struct EntryRegs {
    Fe2<1> N, D, E, F;
};
// `ef` (kernels.hip.h: LdsEF / NoEF): where E and F come from -- the entry itself, or the lane's copy of them in LDS
template <typename L = LimbSlots, typename P, typename EF> FQ_DEV EntryRegs load_entry(const P* entry, u32 neg_mask, u32 digit, const EF& ef) {
    EntryRegs t;
    (void)neg_mask;
    load_nd<L>(entry, t.N, t.D);                                    // the sign is applied by add_entry, behind the doubling
    if constexpr (EF::ON) {
        t.E = ef.get(digit, 0); t.F = ef.get(digit, 1);
    } else {
        t.E = L::load(entry + 2 * L::COORD); t.F = L::load(entry + 3 * L::COORD);
    }
    return t;
}
template <int CH> FQ_DEV R1 add_entry(const R1& q, const EntryRegs& t, u32 neg_mask) {
    // The masked exchange of N and D must not be scheduled next to the gathers (hipcc does that when it may, and the lone
    // wave then sits out the gather latency at the top of every step: measured -7 % on the headline kernel).  The mask is
    // made to depend on the doubled point, so the twenty selects can only issue once the doubling has been computed.
    asm("" : "+v"(neg_mask) : "v"(q.X.re.l[0]), "v"(q.Y.re.l[0]), "v"(q.Z.re.l[0]));
    const Fe2<1> tN = fe2_bitselect(neg_mask, t.D, t.N), tD = fe2_bitselect(neg_mask, t.N, t.D);
    Fe2<1> T = fe2_mulx<CH>(q.Ta, q.Tb);
    Fe2<2> N1 = fe2_add(q.X, q.Y);
    Fe2<3> D1 = fe2_subx<CH>(q.Y, q.X);
    Fe2<1> A = fe2_mulx<CH>(D1, tD);
    Fe2<1> B = fe2_mulx<CH>(N1, tN);
    Fe2<1> C = fe2_mulx<CH>(fe2_cnegx<CH>(t.F, neg_mask), T);
    Fe2<1> D = fe2_mulx<CH>(t.E, q.Z);
    Fe2<3> E = fe2_subx<CH>(B, A);
    Fe2<3> F = fe2_subx<CH>(D, C);
    Fe2<2> G = fe2_add(D, C);
    Fe2<2> H = fe2_add(B, A);
    R1 r;
    r.X = fe2_mulx<CH>(E, F);
    r.Z = fe2_mulx<CH>(G, F);
    r.Y = fe2_mulx<CH>(G, H);
    r.Ta = widen<4>(E);
    r.Tb = H;
    return r;
}
// Q + (+-A) for a precomputed AFFINE point A = (x+y, y-x, 2d*x*y) read from `entry` (three coordinates of
// COORD_U32 dwords): ADD_core with the table point's 2Z = 2, i.e. D = 2*Z1 costs no multiplication.  Used by
// the fixed-base comb, not by the reference-shaped MUL_* paths.
// an opaque copy of every limb: the value is (re)defined in the current basic block as a 32-bit register
template <int B> FQ_DEV void fe2_here(Fe2<B>& x) {
#pragma unroll
    for (int i = 0; i < 5; i++) { FQ_SIGN_UNKNOWN(x.re.l[i]); FQ_SIGN_UNKNOWN(x.im.l[i]); }
}
template <int CH = (FQ_CHAIN != 0) ? 1 : 0, typename P> FQ_DEV R1 add_affine_table(const R1& q, const P* entry, u32 neg_mask) {
    // In the comb's loop Q reaches this addition through a join (the doubling is conditional), and hipcc sign-extends the limbs
    // of Ta and Tb on the far side of it: instruction selection, which works block by block, then sees 64-bit operands and
    // expands each of 90 products into v_mad_u64_u32 + 2 v_mul_lo_u32 + v_add3_u32.  Opaque copies keep the limbs 32-bit here.
    Fe2<4> Ta = q.Ta;
    Fe2<2> Tb = q.Tb;
    if constexpr (CH == 2) { fe2_here(Ta); fe2_here(Tb); }
    Fe2<1> T = fe2_mulx<CH>(Ta, Tb);
    Fe2<2> N1 = fe2_add(q.X, q.Y);
    Fe2<3> D1 = fe2_subx<CH>(q.Y, q.X);
    Fe2<1> tN, tD;
    load_signed_nd<LimbSlots>(entry, neg_mask, tN, tD);
    Fe2<1> A = fe2_mulx<CH>(D1, tD);
    Fe2<1> B = fe2_mulx<CH>(N1, tN);
    Fe2<1> C = fe2_mulx<CH>(fe2_cnegx<CH>(load_fe2_limbs(entry + 2 * COORD_U32), neg_mask), T);
    Fe2<2> D = fe2_dbl(q.Z);
    Fe2<3> E = fe2_subx<CH>(B, A);
    Fe2<3> G = fe2_add(D, C);
    Fe2<2> H = fe2_add(B, A);
    R1 r;
    if constexpr (CH == 2) {       // signed: D - C has bound 3 and fits the 8*b operand
        Fe2<3> F = fe2_sub_signed(D, C);
        r.X = fe2_mulx<CH>(E, F);
        r.Z = fe2_mulx<CH>(G, F);
    } else {
        Fe2<4> F = fe2_sub(D, C);
        r.X = fe2_mulx<CH>(E, F);
        r.Z = fe2_mulx<CH>(G, F);
    }
    r.Y = fe2_mulx<CH>(G, H);
    r.Ta = widen<4>(E);
    r.Tb = H;
    return r;
}
// the comb's starting point: +-A as an R1 point (Z = 1)
template <typename P> FQ_DEV R1 affine_table_start(const P* entry, u32 neg_mask) {
    Fe2<1> N, D;                                                                     // x+y, y-x of +-A
    load_signed_nd<LimbSlots>(entry, neg_mask, N, D);
    // x = (N - D)/2, y = (N + D)/2: keep the factor 2 projectively: (X, Y, Z) = (N - D, N + D, 2)
    R1 r;
    r.X = fe2_carry(fe2_sub(N, D));
    r.Y = fe2_carry(fe2_add(N, D));
    r.Z = fe2_carry(fe2_dbl(fe2_one()));
    r.Ta = widen<4>(r.X);                                   // Ta*Tb must equal X*Y/Z = (N-D)(N+D)/2
    r.Tb = widen<2>(fe2_mul(r.Y, fe2_half_const()));        // Tb = (N+D)/2
    return r;
}

// ---- constant-time selection (FOURQ_CT_SELECT; draft-ladd-cfrg-4q.md:753-758) --------------------------------
// "Implementations MUST ensure that ... memory addresses accessed do not depend on secret data."  The default ladders
// above use the digit as an address, exactly as the reference does (curve4q.py:232, :440: T[ind[i]]), and the sign
// by masked selects (load_signed_nd).  The sources below read EVERY entry of the table at every step and keep the wanted one with
// masks derived arithmetically from the digit (one v_and_or_b32 per limb and entry); the N/D swap of a negated entry
// is a masked XOR swap, -F a masked negation.  Same values, hence the same R1 tuples.
//   ScanMem    the table in LDS (all lanes read the same addresses: broadcasts) or in global memory
//   ScanSplit  a per-lane table: N, D in registers, E, F in the lane's LDS rows (fused variable-base kernels: no global
//              memory traffic in the ladder)
// One of 2^LOG values by a binary TREE of masked selects: neighbours by bit 0 of the digit, neighbouring winners by bit 1, and so on --
// ENTRIES - 1 selects per limb (v_bitop3_b32, truth table 0xCA) where a scan with one equality mask per entry takes ENTRIES
// compares and ENTRIES masked ORs.  The bit masks live in vector registers and are opaque to hipcc: from a visible `0 - (a == b)`
// it makes v_cmp / v_cndmask pairs, whose SGPR hazards cost the constant-time ladder step 80 s_nop besides (141 against 60).
// Round 3: 486 -> 280 selection instructions per fused step, constant-time cfg2 x1.29 -> see DESIGN.md section 10.
template <int LOG> struct DigitBits {
    u32 m[LOG];                                    // m[j] = ~0 where bit j of the digit is set
    FQ_DEV explicit DigitBits(u32 digit) {
#pragma unroll
        for (int j = 0; j < LOG; j++) { u32 x = 0u - ((digit >> j) & 1u); asm("" : "+v"(x)); m[j] = x; }
    }
};
constexpr int log2_of(int n) { return n <= 1 ? 0 : 1 + log2_of(n / 2); }
template <int N> FQ_DEV void pick(u32 r[N], u32 mask, const u32 hi[N], const u32 lo[N]) {      // mask ? hi : lo, limb by limb
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = __builtin_amdgcn_bitop3_b32(mask, hi[i], lo[i], 0xCA);
}
// The entries arrive in pairs (2P, 2P + 1), P = 0, 1, ...: the winner of a finished subtree waits in pend[] for its sibling, so at most
// LOG - 1 partial results are alive whatever ENTRIES is.  After the last pair `out` holds entry `digit`.
template <int N, int LOG> struct SelectTree {
    u32 pend[LOG > 1 ? LOG - 1 : 1][N];
    template <int P> FQ_DEV void feed(const DigitBits<LOG>& b, const u32 e0[N], const u32 e1[N], u32 out[N]) {
        u32 cur[N];
        pick<N>(cur, b.m[0], e1, e0);
        constexpr int UP = trailing_ones(P);       // subtrees this pair completes
#pragma unroll
        for (int j = 0; j < UP && j < LOG - 1; j++) {
            u32 nxt[N];
            pick<N>(nxt, b.m[j + 1], cur, pend[j]);
#pragma unroll
            for (int i = 0; i < N; i++) cur[i] = nxt[i];
        }
        if constexpr (UP >= LOG - 1) {
#pragma unroll
            for (int i = 0; i < N; i++) out[i] = cur[i];
        } else {
#pragma unroll
            for (int i = 0; i < N; i++) pend[UP][i] = cur[i];
        }
    }
    static constexpr int trailing_ones(int p) { return (p & 1) ? 1 + trailing_ones(p >> 1) : 0; }
};
FQ_DEV Fe2<1> fe2_from_limbs(const u32 w[10]) {
    Fe2<1> r;
#pragma unroll
    for (int i = 0; i < 5; i++) { r.re.l[i] = w[i]; r.im.l[i] = w[5 + i]; }
    return r;
}
// masked swap: (a, b) -> (b, a) where mask == ~0
FQ_DEV void fe2_cswap(Fe2<1>& a, Fe2<1>& b, u32 mask) {
#pragma unroll
    for (int i = 0; i < 5; i++) {
        u32 t = (a.re.l[i] ^ b.re.l[i]) & mask; a.re.l[i] ^= t; b.re.l[i] ^= t;
        u32 u = (a.im.l[i] ^ b.im.l[i]) & mask; a.im.l[i] ^= u; b.im.l[i] ^= u;
    }
}
template <int ENTRIES, typename TP> struct ScanMem {
    const TP* tbl;
    int stride;                                    // dwords between entries; coordinates are COORD_U32 apart
    // Two entries per round, rounds fenced: left alone the scheduler hoists all 3 * ENTRIES loads ahead of the masking
    // and the 128-VGPR kernels spill (measured: 1.3 KB of scratch per lane).
    template <int P, int LOG> FQ_DEV void pair(const DigitBits<LOG>& bits, SelectTree<10, LOG>& tree, int c, u32 out[10]) const {
        const uint4* q0 = reinterpret_cast<const uint4*>(tbl + (2 * P) * stride + c * COORD_U32);
        const uint4* q1 = reinterpret_cast<const uint4*>(tbl + (2 * P + 1) * stride + c * COORD_U32);
        const uint4 a0 = q0[0], b0 = q0[1], d0 = q0[2], a1 = q1[0], b1 = q1[1], d1 = q1[2];
        const u32 v0[10] = { a0.x, a0.y, a0.z, a0.w, b0.x, b0.y, b0.z, b0.w, d0.x, d0.y };
        const u32 v1[10] = { a1.x, a1.y, a1.z, a1.w, b1.x, b1.y, b1.z, b1.w, d1.x, d1.y };
        tree.template feed<P>(bits, v0, v1, out);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (2 * P + 2 < ENTRIES) pair<P + 1>(bits, tree, c, out);
    }
    static_assert(ENTRIES >= 2 && (ENTRIES & (ENTRIES - 1)) == 0, "a power of two of entries, scanned in pairs");
    using Bits = DigitBits<log2_of(ENTRIES)>;      // the digit's bit masks: formed once per step, used for every coordinate
    FQ_DEV Fe2<1> coord(const Bits& bits, int c) const {           // coordinate c of the entry the bits stand for
        SelectTree<10, log2_of(ENTRIES)> tree;
        u32 out[10];
        pair<0>(bits, tree, c, out);
        return fe2_from_limbs(out);
    }
};
// The fused constant-time kernels: N and D of the lane's eight entries in registers (loaded once from the HBM slot), E and F
// scanned where the fused kernels keep them anyway, in the lane's private rows of LDS (kernels.hip.h, LdsEF) -- the LDS
// addresses depend on the lane and the entry number, never on the digit.
template <typename EFT> struct ScanSplit {
    u32 nd[8][20];
    EFT ef;
    template <typename L, typename TP> FQ_DEV void load(const TP* tbl) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            Fe2<1> v[2];
            load_nd<L>(tbl + k * L::ENTRY, v[0], v[1]);
#pragma unroll
            for (int c = 0; c < 2; c++) {
#pragma unroll
                for (int i = 0; i < 5; i++) { nd[k][c * 10 + i] = v[c].re.l[i]; nd[k][c * 10 + 5 + i] = v[c].im.l[i]; }
            }
        }
    }
    template <int P> FQ_DEV void pair(const DigitBits<3>& bits, SelectTree<10, 3>& tree, int c, u32 out[10]) const {
        if (c < 2) {
            tree.template feed<P>(bits, &nd[2 * P][c * 10], &nd[2 * P + 1][c * 10], out);
        } else {                                                   // two entries per round, as ScanMem
            const Fe2<1> v0 = ef.get((u32)(2 * P), c - 2), v1 = ef.get((u32)(2 * P + 1), c - 2);
            const u32 w0[10] = { v0.re.l[0], v0.re.l[1], v0.re.l[2], v0.re.l[3], v0.re.l[4], v0.im.l[0], v0.im.l[1], v0.im.l[2], v0.im.l[3], v0.im.l[4] };
            const u32 w1[10] = { v1.re.l[0], v1.re.l[1], v1.re.l[2], v1.re.l[3], v1.re.l[4], v1.im.l[0], v1.im.l[1], v1.im.l[2], v1.im.l[3], v1.im.l[4] };
            tree.template feed<P>(bits, w0, w1, out);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (P < 3) pair<P + 1>(bits, tree, c, out);
    }
    using Bits = DigitBits<3>;
    FQ_DEV Fe2<1> coord(const Bits& bits, int c) const {
        SelectTree<10, 3> tree;
        u32 out[10];
        pair<0>(bits, tree, c, out);
        return fe2_from_limbs(out);
    }
};
// Q + (+-T[digit]), every entry read: the constant-time form of add_table
template <int CH, typename SRC> FQ_DEV R1 add_scan(const R1& q, const SRC& src, u32 digit_value, u32 neg_mask) {
    const typename SRC::Bits digit(digit_value);
    Fe2<1> T = fe2_mulx<CH>(q.Ta, q.Tb);
    Fe2<2> N1 = fe2_add(q.X, q.Y);
    Fe2<3> D1 = fe2_subx<CH>(q.Y, q.X);
    Fe2<1> tN = src.coord(digit, 0), tD = src.coord(digit, 1);
    fe2_cswap(tN, tD, neg_mask);                                   // R2neg: (D, N, E, -F)
    Fe2<1> A = fe2_mulx<CH>(D1, tD);
    Fe2<1> B = fe2_mulx<CH>(N1, tN);
    Fe2<1> C = fe2_mulx<CH>(fe2_cnegx<CH>(src.coord(digit, 3), neg_mask), T);
    Fe2<1> D = fe2_mulx<CH>(src.coord(digit, 2), q.Z);
    Fe2<3> E = fe2_subx<CH>(B, A);
    Fe2<3> F = fe2_subx<CH>(D, C);
    Fe2<2> G = fe2_add(D, C);
    Fe2<2> H = fe2_add(B, A);
    R1 r;
    r.X = fe2_mulx<CH>(E, F);
    r.Z = fe2_mulx<CH>(G, F);
    r.Y = fe2_mulx<CH>(G, H);
    r.Ta = widen<4>(E);
    r.Tb = H;
    return r;
}
template <typename SRC> FQ_DEV Proj<1, 1, 1> start_scan(const SRC& src, u32 digit_value, u32 neg_mask) {
    const typename SRC::Bits digit(digit_value);
    Fe2<1> N = src.coord(digit, 0), D = src.coord(digit, 1);
    fe2_cswap(N, D, neg_mask);
    Proj<1, 1, 1> r;
    r.X = fe2_carry(fe2_sub(N, D));
    r.Y = fe2_carry(fe2_add(D, N));
    r.Z = src.coord(digit, 2);
    return r;
}
// the comb's mixed addition and starting point with every one of the block's entries read (3 coordinates each)
template <int CH, typename SRC> FQ_DEV R1 add_affine_scan(const R1& q, const SRC& src, u32 idx_value, u32 neg_mask) {
    const typename SRC::Bits idx(idx_value);
    Fe2<4> Ta = q.Ta;
    Fe2<2> Tb = q.Tb;
    if constexpr (CH == 2) { fe2_here(Ta); fe2_here(Tb); }      // as in add_affine_table: keeps the 90 products single instructions
    Fe2<1> T = fe2_mulx<CH>(Ta, Tb);
    Fe2<2> N1 = fe2_add(q.X, q.Y);
    Fe2<3> D1 = fe2_subx<CH>(q.Y, q.X);
    Fe2<1> tN = src.coord(idx, 0), tD = src.coord(idx, 1);
    fe2_cswap(tN, tD, neg_mask);
    Fe2<1> A = fe2_mulx<CH>(D1, tD);
    Fe2<1> B = fe2_mulx<CH>(N1, tN);
    Fe2<1> C = fe2_mulx<CH>(fe2_cnegx<CH>(src.coord(idx, 2), neg_mask), T);
    Fe2<2> D = fe2_dbl(q.Z);
    Fe2<3> E = fe2_subx<CH>(B, A);
    Fe2<3> G = fe2_add(D, C);
    Fe2<2> H = fe2_add(B, A);
    R1 r;
    if constexpr (CH == 2) {
        Fe2<3> F = fe2_sub_signed(D, C);
        r.X = fe2_mulx<CH>(E, F);
        r.Z = fe2_mulx<CH>(G, F);
    } else {
        Fe2<4> F = fe2_sub(D, C);
        r.X = fe2_mulx<CH>(E, F);
        r.Z = fe2_mulx<CH>(G, F);
    }
    r.Y = fe2_mulx<CH>(G, H);
    r.Ta = widen<4>(E);
    r.Tb = H;
    return r;
}
template <typename SRC> FQ_DEV R1 affine_scan_start(const SRC& src, u32 idx_value, u32 neg_mask) {
    const typename SRC::Bits idx(idx_value);
    Fe2<1> N = src.coord(idx, 0), D = src.coord(idx, 1);
    fe2_cswap(N, D, neg_mask);
    R1 r;
    r.X = fe2_carry(fe2_sub(N, D));
    r.Y = fe2_carry(fe2_add(N, D));
    r.Z = fe2_carry(fe2_dbl(fe2_one()));
    r.Ta = widen<4>(r.X);
    r.Tb = widen<2>(fe2_mul(r.Y, fe2_half_const()));
    return r;
}

// R2toR4(selectpt(s, T, nT)): the ladder's starting point (curve4q.py:229, :437)
template <typename L = LimbSlots, typename P> FQ_DEV Proj<1, 1, 1> start_table(const P* entry, u32 neg_mask) {
    Fe2<1> N, D;
    load_signed_nd<L>(entry, neg_mask, N, D);
    Proj<1, 1, 1> r;
    r.X = fe2_carry(fe2_sub(N, D));
    r.Y = fe2_carry(fe2_add(D, N));
    if constexpr (std::is_same<L, NDSlots>::value) r.Z = Fe2<1>{};      // E lives in LDS with this layout: the caller takes it from there
    else r.Z = L::load(entry + 2 * L::COORD);
    return r;
}

}  // namespace fq
