// Two lanes per element: the variable-base kernels (MUL_endo, MUL_windowed, DH_*; both selection modes) for SMALL batches and for
// the remainder of a batch past whole generations.
//
// One lane owns one scalar multiplication for 0.33 ms of dependent instructions however few lanes are busy (DESIGN.md section 9:
// the generation cliff), so a batch of 100 elements and the 256 elements past a full generation each cost what 65 536 cost.
// Here lane 2k holds the REAL parts and lane 2k + 1 the IMAGINARY parts of every GF(p^2) value of element k (each a GF(p)
// element: 5 signed 26-bit limbs).  Additions are limb-wise on the lane's half; a product a*b is one sum of two GF(p) products
// per lane -- even lane: a_re*b_re + (-a_im)*b_im, odd lane: a_im*b_re + a_re*b_im -- with the partner's halves fetched by
// v_mov_b32_dpp quad_perm (no LDS, no waits); a square is one GF(p) product per lane.  A lane issues 1 454 instead of 2 154
// instructions per ladder step: 0.66 of the one-lane kernels' latency at one wave per SIMD (profiles/r03_cliff.txt).  At two waves
// per SIMD, i.e. the same elements per chip, the step is within +-3 % of the one-lane step (tools/microbench/pairlane.hip,
// profiles/r03_pairlane.txt): a latency lever, not a throughput lever, which is why only small batches and remainders come here.
//
// Same formula DAG as curve.hip.h (curve4q.py:109-175, :179-185, :228-235, :258-322, :385-462), hence the same residues in the R1
// tuple and the same affine DH outputs and verdicts.  Signed flavour throughout (fp127.hip.h): every element type carries the bound B
// of its limb magnitudes and every product static_asserts its operand and column bounds, as the one-lane code does.
#pragma once
#include "curve.hip.h"
#include "recode.hip.h"

namespace fq {

template <int B> struct PF { u32 l[5]; };          // this lane's half of a GF(p^2) value, |limb| <= B * UNIT
struct PairLane { u32 even, odd_neg; };            // even = ~0 on lanes holding real parts; odd_neg = ~0 on lanes holding imaginary parts

template <int B2, int B> FQ_DEV PF<B2> pwiden(const PF<B>& a) {
    static_assert(B2 >= B, "cannot narrow a bound");
    PF<B2> r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.l[i] = a.l[i];
    return r;
}
template <int A, int B> FQ_DEV PF<A + B> padd(const PF<A>& a, const PF<B>& b) {
    static_assert((u64)(A + B) * UNIT < (1ull << 31), "limb overflow");
    PF<A + B> r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
template <int A, int B> FQ_DEV PF<A + B> psub(const PF<A>& a, const PF<B>& b) {
    static_assert((u64)(A + B) * UNIT < (1ull << 31), "limb overflow");
    PF<A + B> r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.l[i] = a.l[i] - b.l[i];
    return r;
}
template <int B> FQ_DEV PF<2 * B> pdbl(const PF<B>& a) { return padd(a, a); }
template <int B> FQ_DEV PF<B> pcneg(const PF<B>& a, u32 m) {      // -a where m == ~0
    PF<B> r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.l[i] = (a.l[i] ^ m) - m;
    return r;
}
template <int B> FQ_DEV PF<B> pneg(const PF<B>& a) { return pcneg(a, ~0u); }
template <int B> FQ_DEV PF<B> pconj(const PF<B>& a, const PairLane& pl) { return pcneg(a, pl.odd_neg); }   // fields.py:189-191

constexpr int DPP_SWAP = 0xB1, DPP_EVEN = 0xA0, DPP_ODD = 0xF5;      // quad_perm [1,0,3,2], [0,0,2,2], [1,1,3,3]
template <int CTRL, int B> FQ_DEV PF<B> pdpp(const PF<B>& a) {
    PF<B> r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.l[i] = (u32)__builtin_amdgcn_mov_dpp((int)a.l[i], CTRL, 0xF, 0xF, true);
    return r;
}
#define FQ_POPAQUE(x) asm("" : "+v"(x))
// A lane has ONE result column at a time (its partner has the other), so chained multiply-adds would follow each other back to
// back -- and behind every opaque partial sum hipcc pads an s_nop before the next instruction that reads it (467 per ladder step
// measured, a third of the step's issue slots).  Each column is therefore accumulated in TWO chains that alternate (u*v in one,
// w*z in the other; a square's products by parity) and are added once per column: +5 64-bit additions per product, -400 s_nop per step.
#ifndef FQ_PAIR_CHAINS
#define FQ_PAIR_CHAINS 2
#endif
// r = u*v + w*z (one component of a GF(p^2) product), signed limbs, carries chained: the column loop of fe2_mul_signed
template <int U, int V> FQ_DEV PF<1> pmac2(const u32 u[5], const u32 v[5], const u32 w[5], const u32 z[5]) {
    static_assert(cols_ok_signed((u64)2 * U * V), "column overflow");
    static_assert(fits8_signed<V>(), "8*v does not fit a signed 32-bit operand: swap the operands");
    u32 v8[5], z8[5];
#pragma unroll
    for (int i = 0; i < 5; i++) { v8[i] = v[i] << 3; z8[i] = z[i] << 3; }
    u32 l[5];
    i64 acc = 0;
#pragma unroll
    for (int K = 0; K < 5; K++) {
        i64 side = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int j = K - i;
            acc += smul(u[i], j >= 0 ? v[j >= 0 ? j : 0] : v8[j >= 0 ? 0 : j + 5]); FQ_POPAQUE(acc);
            if (FQ_PAIR_CHAINS == 2) { side += smul(w[i], j >= 0 ? z[j >= 0 ? j : 0] : z8[j >= 0 ? 0 : j + 5]); FQ_POPAQUE(side); }
            else { acc += smul(w[i], j >= 0 ? z[j >= 0 ? j : 0] : z8[j >= 0 ? 0 : j + 5]); FQ_POPAQUE(acc); }
        }
        if (FQ_PAIR_CHAINS == 2) acc += side;
        l[K] = (u32)acc & LIMB_MASK; acc >>= LIMB_BITS;
    }
    const Fe<1> f = fe_finish_signed(l[0], l[1], l[2], l[3], l[4], acc);
    PF<1> r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.l[i] = f.l[i];
    return r;
}
// a * b.  even lane: re = a_re*b_re - a_im*b_im ; odd lane: im = a_im*b_re + a_re*b_im: in both, u = this lane's half of a,
// v = b_re, w = the partner's half of a (negated on even lanes), z = b_im.  The operand whose 8-fold must fit 32 bits is b: when
// only a's does, the (commutative) product is taken the other way round.
template <int A, int B> FQ_DEV PF<1> pmul(const PF<A>& a, const PF<B>& b, const PairLane& pl) {
    if constexpr (!fits8_signed<B>() && fits8_signed<A>()) {
        return pmul(b, a, pl);
    } else {
        const PF<A> w = pcneg(pdpp<DPP_SWAP>(a), pl.even);
        const PF<B> v = pdpp<DPP_EVEN>(b), z = pdpp<DPP_ODD>(b);
        return pmac2<A, B>(a.l, v.l, w.l, z.l);
    }
}
// a * c for a curve constant c: both halves of c are literals, no exchange for them
template <int A> FQ_DEV PF<1> pmul_const(const PF<A>& a, const Fe2<1>& c, const PairLane& pl) {
    const PF<A> w = pcneg(pdpp<DPP_SWAP>(a), pl.even);
    u32 v[5], z[5];
#pragma unroll
    for (int i = 0; i < 5; i++) { v[i] = c.re.l[i]; z[i] = c.im.l[i]; FQ_SIGN_UNKNOWN(v[i]); FQ_SIGN_UNKNOWN(z[i]); }
    return pmac2<A, 1>(a.l, v, w.l, z);
}
// a^2.  even lane: (a_re + a_im)(a_re - a_im) ; odd lane: (2 a_re) a_im: one GF(p) product u*v per lane      fields.py:176-181
template <int A> FQ_DEV PF<1> psqr(const PF<A>& a, const PairLane& pl) {
    static_assert((u64)2 * A * UNIT < (1ull << 31), "limb overflow");
    static_assert(cols_ok_signed((u64)(2 * A) * (2 * A)), "column overflow");
    const PF<A> o = pdpp<DPP_SWAP>(a);
    u32 u[5], v[5], uw[5], vw[5];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        u[i] = o.l[i] + __builtin_amdgcn_bitop3_b32(pl.even, a.l[i], o.l[i], 0xCA);      // even: a + o ; odd: 2 o      (bound 2A)
        v[i] = a.l[i] - (o.l[i] & pl.even);                                              // even: a - o ; odd: a        (bound 2A)
    }
    constexpr bool WIDE = !fits8_signed<2 * A>();            // 8*v would not fit: the wrap-around factor 8 as (4u)(2v), as wrap_operands_signed
    static_assert(!WIDE || ((u64)4 * 2 * A * UNIT < (1ull << 31)), "no split of the factor 8 fits");
#pragma unroll
    for (int i = 0; i < 5; i++) { vw[i] = v[i] << (WIDE ? 1 : 3); uw[i] = WIDE ? u[i] << 2 : u[i]; }
    u32 l[5];
    i64 acc = 0;
#pragma unroll
    for (int K = 0; K < 5; K++) {
        i64 side = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int j = K - i;
            const i64 prod = smul(j >= 0 ? u[i] : uw[i], j >= 0 ? v[j >= 0 ? j : 0] : vw[j >= 0 ? 0 : j + 5]);
            if (FQ_PAIR_CHAINS == 2 && (i & 1)) { side += prod; FQ_POPAQUE(side); }
            else { acc += prod; FQ_POPAQUE(acc); }
        }
        if (FQ_PAIR_CHAINS == 2) acc += side;
        l[K] = (u32)acc & LIMB_MASK; acc >>= LIMB_BITS;
    }
    const Fe<1> f = fe_finish_signed(l[0], l[1], l[2], l[3], l[4], acc);
    PF<1> r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.l[i] = f.l[i];
    return r;
}
#undef FQ_POPAQUE

// ---- points (halves) -------------------------------------------------------------------------------------------------
struct PR1 { PF<1> X, Y, Z; PF<3> Ta; PF<2> Tb; };      // (X, Y, Z, Ta, Tb)
struct PR2 { PF<2> N, D, E; PF<1> F; };                 // (X+Y, Y-X, 2Z, 2dT): a table entry
struct PR3 { PF<2> N, D; PF<1> E, F; };                 // (X+Y, Y-X, Z, T)
struct PProj { PF<1> X, Y, Z; };

FQ_DEV PR2 pr1_to_r2(const PR1& p, const PairLane& pl) {                                    // curve4q.py:109-116
    PR2 r;
    r.N = padd(p.X, p.Y); r.D = psub(p.Y, p.X); r.E = pdbl(p.Z);
    r.F = pmul_const(pmul(p.Ta, p.Tb, pl), fe2_two_d(), pl);
    return r;
}
FQ_DEV PR3 pr1_to_r3(const PR1& p, const PairLane& pl) {                                    // curve4q.py:119-126
    PR3 r;
    r.N = padd(p.X, p.Y); r.D = psub(p.Y, p.X); r.E = p.Z; r.F = pmul(p.Ta, p.Tb, pl);
    return r;
}
FQ_DEV PR1 pdbl_point(const PF<1>& X, const PF<1>& Y, const PF<1>& Z, const PairLane& pl) {  // curve4q.py:138-152 (the signed DAG of curve.hip.h)
    const PF<1> A = psqr(X, pl), B = psqr(Y, pl);
    const PF<2> C = pdbl(psqr(Z, pl));
    const PF<2> D = padd(A, B);
    const PF<3> E = psub(psqr(padd(X, Y), pl), D);
    const PF<2> F = psub(B, A);
    const PF<4> G = psub(C, F);
    PR1 r;
    r.X = pmul(G, E, pl); r.Z = pmul(G, F, pl); r.Y = pmul(D, F, pl);
    r.Ta = E; r.Tb = D;
    return r;
}
FQ_DEV PR1 padd_core(const PR3& p, const PR2& q, const PairLane& pl) {                       // curve4q.py:155-171
    const PF<1> A = pmul(p.D, q.D, pl), B = pmul(p.N, q.N, pl), C = pmul(q.F, p.F, pl), D = pmul(q.E, p.E, pl);
    const PF<2> E = psub(B, A), F = psub(D, C), G = padd(D, C), H = padd(B, A);
    PR1 r;
    r.X = pmul(E, F, pl); r.Y = pmul(G, H, pl); r.Z = pmul(F, G, pl);
    r.Ta = pwiden<3>(E); r.Tb = H;
    return r;
}
FQ_DEV PProj ptau(const PF<1>& X, const PF<1>& Y, const PF<1>& Z, const PairLane& pl) {      // curve4q.py:258-267
    const PF<1> A = psqr(X, pl), B = psqr(Y, pl);
    const PF<2> C = padd(A, B), D = psub(A, B);
    PProj r;
    r.X = pmul(D, pmul(Y, pmul_const(X, c_tau(), pl), pl), pl);
    r.Y = pneg(pmul(padd(pdbl(psqr(Z, pl)), D), C, pl));
    r.Z = pmul(D, C, pl);
    return r;
}
FQ_DEV PR1 ptau_dual(const PF<1>& X, const PF<1>& Y, const PF<1>& Z, const PairLane& pl) {   // curve4q.py:269-280
    const PF<1> A = psqr(X, pl), B = psqr(Y, pl);
    const PF<2> C = padd(A, B), Ta = psub(B, A);
    const PF<4> D = psub(pdbl(psqr(Z, pl)), Ta);
    const PF<1> Tb = pmul(Y, pmul_const(X, c_taudual(), pl), pl);
    PR1 r;
    r.X = pmul(C, Tb, pl); r.Y = pmul(D, Ta, pl); r.Z = pmul(D, C, pl);
    r.Ta = pwiden<3>(Ta); r.Tb = pwiden<2>(Tb);
    return r;
}
FQ_DEV PProj pupsilon(const PProj& p, const PairLane& pl) {                                  // curve4q.py:282-302
    const PF<1> A = pmul(p.Y, pmul_const(p.X, c_phi<0>(), pl), pl), B = pmul(p.Y, p.Z, pl);
    const PF<1> C = psqr(p.Y, pl), D = psqr(p.Z, pl);
    const PF<1> F = psqr(D, pl), G = psqr(B, pl), H = psqr(C, pl);
    const PF<1> I = pmul_const(B, c_phi<1>(), pl);
    const PF<2> J = padd(C, pmul_const(D, c_phi<2>(), pl));
    const PF<3> K = padd(padd(pmul_const(G, c_phi<8>(), pl), H), pmul_const(F, c_phi<9>(), pl));
    const PF<1> x2 = pmul(psub(I, J), padd(I, J), pl);
    const PF<2> L = padd(C, pmul_const(D, c_phi<4>(), pl));
    const PF<1> M = pmul_const(B, c_phi<3>(), pl);
    const PF<1> Nn = pmul(psub(L, M), padd(L, M), pl);
    const PF<3> y2 = padd(padd(H, pmul_const(G, c_phi<6>(), pl)), pmul_const(F, c_phi<7>(), pl));
    PProj r;
    r.X = pconj(pmul(pmul(K, A, pl), x2, pl), pl);
    r.Y = pconj(pmul(y2, pmul(pmul_const(D, c_phi<5>(), pl), Nn, pl), pl), pl);
    r.Z = pconj(pmul(pmul(K, B, pl), Nn, pl), pl);
    return r;
}
FQ_DEV PProj pchi(const PProj& p, const PairLane& pl) {                                      // curve4q.py:304-316
    const PF<1> A = pconj(p.X, pl), B = pconj(p.Y, pl);
    const PF<1> C = psqr(pconj(p.Z, pl), pl), D = psqr(A, pl);
    const PF<1> G = pmul(B, padd(D, pmul_const(C, c_psi<2>(), pl)), pl);
    const PF<2> H = pneg(padd(D, pmul_const(C, c_psi<4>(), pl)));
    PProj r;
    r.X = pmul(H, pmul(pmul_const(A, c_psi<1>(), pl), C, pl), pl);
    r.Y = pmul(padd(D, pmul_const(C, c_psi<3>(), pl)), G, pl);
    r.Z = pmul(H, G, pl);
    return r;
}

// ---- four lanes per element: the element's two pairs share the products of a formula level -----------------------------
// Lanes 4k, 4k+1 are the FIRST pair of element k and lanes 4k+2, 4k+3 the SECOND; both pairs hold every value of the element
// (each lane its half), so all of the pair code above runs unchanged -- twice over.  What the second pair buys: DBL and ADD are
// levels of independent GF(p^2) products (4 squares, then 4 products; 4 products, then 3), and a level's products are taken two
// at a time, ONE PER PAIR: the operands are chosen per pair (v_cndmask on the lane's pair bit, a public value), each pair
// multiplies, and one quad_perm [2,3,0,1] exchange hands either result to the other pair.  978 instead of 1 454 instructions per
// ladder step and lane: batches of at most a QUARTER generation take 0.7 of the two-lane latency (profiles/r03_quadlane.txt).
// Same DAG, same residues (curve4q.py:138-171).
constexpr int DPP_QSWAP = 0x4E;                         // quad_perm [2,3,0,1]
struct QuadLane { bool second; };                       // this lane belongs to the element's second pair
template <int A, int B> FQ_DEV PF<(A > B ? A : B)> qsel(const QuadLane& ql, const PF<A>& first, const PF<B>& second) {
    PF<(A > B ? A : B)> r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.l[i] = ql.second ? second.l[i] : first.l[i];
    return r;
}
// r holds one value in the first pair and another in the second: both values to both pairs
template <int B> FQ_DEV void qshare(const QuadLane& ql, const PF<B>& r, PF<B>& first, PF<B>& second) {
    const PF<B> o = pdpp<DPP_QSWAP>(r);
    first = qsel(ql, r, o); second = qsel(ql, o, r);
}
// DBL (curve4q.py:138-152; the signed DAG of pdbl_point).  WITH_T: also Ta*Tb of the result (= E*D, the F of R1toR3, curve4q.py:124), which
// the addition that follows needs -- the fourth product of the second level, free on the second pair.
template <bool WITH_T> FQ_DEV PR1 qdbl_point(const PF<1>& X, const PF<1>& Y, const PF<1>& Z, const PairLane& pl, const QuadLane& ql, PF<1>& T) {
    const PF<1> s1 = psqr(qsel(ql, X, Y), pl);                               // X^2 | Y^2
    const PF<1> s2 = psqr(qsel(ql, Z, padd(X, Y)), pl);                      // Z^2 | (X+Y)^2
    PF<1> A, B, Zs, S;
    qshare(ql, s1, A, B); qshare(ql, s2, Zs, S);
    const PF<2> C = pdbl(Zs);
    const PF<2> D = padd(A, B);
    const PF<3> E = psub(S, D);
    const PF<2> F = psub(B, A);
    const PF<4> G = psub(C, F);
    PR1 r;
    const PF<1> p1 = pmul(G, qsel(ql, E, F), pl);                            // X3 = G*E | Z3 = G*F
    qshare(ql, p1, r.X, r.Z);
    if constexpr (WITH_T) {
        const PF<1> p2 = pmul(D, qsel(ql, F, E), pl);                        // Y3 = D*F | T3 = D*E
        qshare(ql, p2, r.Y, T);
    } else {
        r.Y = pmul(D, F, pl);
        T = r.Y;
    }
    r.Ta = E; r.Tb = D;
    return r;
}
// Q + (+-t) for Q = (X, Y, Z, .., ..) with T = Ta*Tb: R1toR3 (curve4q.py:119-126), the sign by masks (selectpt, curve4q.py:193-206),
// ADD_core (curve4q.py:155-171).  The first pair takes p.D*q.D and q.F*p.F, the second p.N*q.N and q.E*p.E: with the sign, the first
// wants (neg ? N : D) of the entry and the second (neg ? D : N) -- ONE masked select on neg ^ pair instead of the two of the pair code.
FQ_DEV PR1 qadd_signed_entry(const PR1& Q, const PF<1>& T, const PR2& t, u32 neg, const PairLane& pl, const QuadLane& ql) {
    asm("" : "+v"(neg) : "v"(Q.X.l[0]), "v"(Q.Y.l[0]), "v"(Q.Z.l[0]));
    const u32 pick_n = neg ^ (ql.second ? ~0u : 0u);
    PF<2> v1;
#pragma unroll
    for (int k = 0; k < 5; k++) v1.l[k] = __builtin_amdgcn_bitop3_b32(pick_n, t.N.l[k], t.D.l[k], 0xCA);
    const PF<1> r1 = pmul(qsel(ql, psub(Q.Y, Q.X), padd(Q.X, Q.Y)), v1, pl);          // A = p.D*q.D | B = p.N*q.N
    const PF<1> r2 = pmul(qsel(ql, pcneg(t.F, neg), t.E), qsel(ql, T, Q.Z), pl);      // C = q.F*p.F | D = q.E*p.E
    PF<1> A, B, C, D;
    qshare(ql, r1, A, B); qshare(ql, r2, C, D);
    const PF<2> E = psub(B, A), F = psub(D, C), G = padd(D, C), H = padd(B, A);
    PR1 r;
    const PF<1> r3 = pmul(qsel(ql, E, G), qsel(ql, F, H), pl);                         // X = E*F | Y = G*H
    qshare(ql, r3, r.X, r.Y);
    r.Z = pmul(F, G, pl);
    r.Ta = pwiden<3>(E); r.Tb = H;
    return r;
}

// Q + (+-A) for an AFFINE comb entry A = (x+y, y-x, 2dxy) (curve.hip.h, add_affine_table), four lanes per element: the products p.D*a.D, p.N*a.N
// go one per pair, a.F*T on both, then X = E*F, Y = G*H | Z = F*G, T' = E*H (the next addition's T for nothing).  T in, T' out.
FQ_DEV PR1 qadd_affine_entry(const PR1& Q, PF<1>& T, const PF<1>& aN, const PF<1>& aD, const PF<1>& aF, u32 neg, const PairLane& pl, const QuadLane& ql) {
    const u32 pick_n = neg ^ (ql.second ? ~0u : 0u);
    PF<1> v1;
#pragma unroll
    for (int k = 0; k < 5; k++) v1.l[k] = __builtin_amdgcn_bitop3_b32(pick_n, aN.l[k], aD.l[k], 0xCA);
    const PF<1> r1 = pmul(qsel(ql, psub(Q.Y, Q.X), padd(Q.X, Q.Y)), v1, pl);            // A = p.D*a.D | B = p.N*a.N
    const PF<1> C = pmul(pcneg(aF, neg), T, pl);
    PF<1> A, B;
    qshare(ql, r1, A, B);
    const PF<2> D = pdbl(Q.Z);
    const PF<2> E = psub(B, A), H = padd(B, A);
    const PF<3> F = psub(D, C), G = padd(D, C);
    PR1 r;
    const PF<1> r3 = pmul(qsel(ql, E, G), qsel(ql, F, H), pl);                           // X = E*F | Y = G*H
    const PF<1> r4 = pmul(qsel(ql, F, E), qsel(ql, G, H), pl);                           // Z = F*G | T' = E*H
    qshare(ql, r3, r.X, r.Y);
    qshare(ql, r4, r.Z, T);
    r.Ta = pwiden<3>(E); r.Tb = H;
    return r;
}

// the same addition two lanes per element (batches between a quarter and half a generation): T = Ta*Tb first, then the seven products in turn
FQ_DEV PR1 padd_affine_entry(const PR1& Q, const PF<1>& aN, const PF<1>& aD, const PF<1>& aF, u32 neg, const PairLane& pl) {
    const PF<1> T = pmul(Q.Ta, Q.Tb, pl);
    PF<1> sN, sD;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        sN.l[k] = __builtin_amdgcn_bitop3_b32(neg, aD.l[k], aN.l[k], 0xCA);
        sD.l[k] = __builtin_amdgcn_bitop3_b32(neg, aN.l[k], aD.l[k], 0xCA);
    }
    const PF<1> A = pmul(psub(Q.Y, Q.X), sD, pl), B = pmul(padd(Q.X, Q.Y), sN, pl), C = pmul(pcneg(aF, neg), T, pl);
    const PF<2> D = pdbl(Q.Z);
    const PF<2> E = psub(B, A), H = padd(B, A);
    const PF<3> F = psub(D, C), G = padd(D, C);
    PR1 r;
    r.X = pmul(E, F, pl); r.Y = pmul(G, H, pl); r.Z = pmul(F, G, pl);
    r.Ta = pwiden<3>(E); r.Tb = H;
    return r;
}

// ---- the pair's table in LDS -----------------------------------------------------------------------------------------
// 8 entries x 4 coordinates x 5 limbs per LANE (each lane keeps its half): 640 bytes per lane, 256 lanes = the CU's 160 KiB, so a
// block is 128 elements and a generation 32 768.  Limb pairs (0,1), (2,3) live in a uint2 region and limb 4 in a u32 region, both
// [value][lane]: consecutive lanes, consecutive banks, whatever the digits are.
constexpr int PAIR_VALUES = 32;
constexpr int PAIR_LDS_U32 = PAIR_VALUES * (2 * 2 + 1) * 256;                                // 163 840 bytes
struct PairTable {
    uint2* two;       // region of limb pairs + threadIdx.x
    u32* one;         // region of limb 4 + threadIdx.x
    template <int B> FQ_DEV void put(int v, const PF<B>& a) const {
        two[(size_t)(2 * v) * 256] = make_uint2(a.l[0], a.l[1]);
        two[(size_t)(2 * v + 1) * 256] = make_uint2(a.l[2], a.l[3]);
        one[(size_t)v * 256] = a.l[4];
    }
    template <int B> FQ_DEV PF<B> get(u32 v) const {
        const uint2 p = two[(size_t)(2 * v) * 256], q = two[(size_t)(2 * v + 1) * 256];
        PF<B> r;
        r.l[0] = p.x; r.l[1] = p.y; r.l[2] = q.x; r.l[3] = q.y; r.l[4] = one[(size_t)v * 256];
        return r;
    }
    FQ_DEV void put_entry(int k, const PR2& t) const { put(4 * k, t.N); put(4 * k + 1, t.D); put(4 * k + 2, t.E); put(4 * k + 3, t.F); }
    FQ_DEV PR2 get_entry(u32 k) const {
        PR2 t;
        t.N = get<2>(4 * k); t.D = get<2>(4 * k + 1); t.E = get<2>(4 * k + 2); t.F = get<1>(4 * k + 3);
        return t;
    }
    // constant-time selection (curve.hip.h, "constant-time selection"): EVERY entry is read -- the lane's LDS addresses depend on the
    // lane and the entry number, never on the digit -- and the wanted one kept by masks derived arithmetically from the digit
    template <int P> FQ_DEV void scan_pair(const DigitBits<3>& bits, SelectTree<20, 3>& tree, u32 out[20]) const {
        const PR2 t0 = get_entry((u32)(2 * P)), t1 = get_entry((u32)(2 * P + 1));
        u32 v0[20], v1[20];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            v0[i] = t0.N.l[i]; v0[5 + i] = t0.D.l[i]; v0[10 + i] = t0.E.l[i]; v0[15 + i] = t0.F.l[i];
            v1[i] = t1.N.l[i]; v1[5 + i] = t1.D.l[i]; v1[10 + i] = t1.E.l[i]; v1[15 + i] = t1.F.l[i];
        }
        tree.template feed<P>(bits, v0, v1, out);
        __builtin_amdgcn_sched_barrier(0);             // two entries per round: left alone the scheduler hoists all 96 reads ahead of the selects
        if constexpr (P < 3) scan_pair<P + 1>(bits, tree, out);
    }
    FQ_DEV PR2 scan_entry(u32 digit) const {
        const DigitBits<3> bits(digit);
        SelectTree<20, 3> tree;
        u32 out[20];
        scan_pair<0>(bits, tree, out);
        PR2 r;
#pragma unroll
        for (int i = 0; i < 5; i++) { r.N.l[i] = out[i]; r.D.l[i] = out[5 + i]; r.E.l[i] = out[10 + i]; r.F.l[i] = out[15 + i]; }
        return r;
    }
    template <bool CT> FQ_DEV PR2 select_entry(u32 digit) const { if constexpr (CT) return scan_entry(digit); else return get_entry(digit); }
};

// T[k] = P + k0*phi(P) + k1*psi(P) + k2*psi(phi(P)), built in the reference's order (curve4q.py:385-403): the structure of
// build_table_endo (kernels.hip.h), with the three endomorphism evaluations sharing one instance of tau, chi / upsilon, tau_dual.
// The working points stay in registers (a half point is 25 of them); the bases T[0..3] are read back from LDS.
template <bool QUAD = false> FQ_DEV void pair_build_table_endo(const PR1& P, const PairTable& tbl, const PairLane& pl, const QuadLane& ql) {
    tbl.put_entry(0, pr1_to_r2(P, pl));
    PF<1> X = P.X, Y = P.Y, Z = P.Z;               // step 0: P, step 1: tau(P) (shared by phi and psi), step 2: phi(P)
    PF<1> QX = P.X, QY = P.Y, QZ = P.Z;            // phi(P), produced by step 0
#pragma unroll 1
    for (int step = 0; step < 3; step++) {
        PProj t;
        if (step == 1) { t.X = X; t.Y = Y; t.Z = Z; }
        else {
            t = ptau(X, Y, Z, pl);
            if (step == 0) { X = t.X; Y = t.Y; Z = t.Z; }        // keep tau(P) for step 1
        }
        const PProj u = step == 0 ? pupsilon(t, pl) : pchi(t, pl);
        const PR1 V = ptau_dual(u.X, u.Y, u.Z, pl);
        if (step == 0) { QX = V.X; QY = V.Y; QZ = V.Z; }
        if (step == 1) { X = QX; Y = QY; Z = QZ; }               // step 2 works on phi(P)
        const PR3 V3 = pr1_to_r3(V, pl);
        const int half = 1 << step;
#pragma unroll 1
        for (int m = 0; m < half; m++) {
            if constexpr (QUAD) tbl.put_entry(half + m, pr1_to_r2(qadd_signed_entry(V, V3.F, tbl.get_entry((u32)m), 0u, pl, ql), pl));   // the addition's products shared by the two pairs
            else tbl.put_entry(half + m, pr1_to_r2(padd_core(V3, tbl.get_entry((u32)m), pl), pl));
        }
    }
}

// MUL_endo's ladder (curve4q.py:436-442) on the pair's table: the entry of a step is read from LDS a whole doubling ahead, its
// sign applied by masked selects behind the doubling (as add_entry in curve.hip.h).
FQ_DEV PR1 pair_start(const PR2& t, u32 neg);
FQ_DEV PR1 padd_signed_entry(const PR1& Q, const PR2& t, u32 neg, const PairLane& pl);
template <bool CT, bool QUAD = false> FQ_DEV PR1 pair_ladder_endo(const EndoDigits& e, const PairTable& tbl, const PairLane& pl, const QuadLane& ql) {
    PR1 Q = pair_start(tbl.select_entry<CT>(e.top & 7), 0u);  // s[64] = 1: the entry itself
#pragma unroll 1
    for (int i = 63; i >= 0; i--) {
        const PR2 t = tbl.select_entry<CT>(endo_digit(e, i)); // read a whole doubling ahead of its use
        if constexpr (QUAD) {
            PF<1> T;
            Q = qdbl_point<true>(Q.X, Q.Y, Q.Z, pl, ql, T);
            Q = qadd_signed_entry(Q, T, t, endo_neg_mask(e, i), pl, ql);
        } else {
            Q = pdbl_point(Q.X, Q.Y, Q.Z, pl);
            Q = padd_signed_entry(Q, t, endo_neg_mask(e, i), pl);
        }
    }
    return Q;
}

// T[0] = R1toR2(P); T[i] = R1toR2(ADD(DBL(P), T[i-1]))                                      curve4q.py:179-185
template <bool QUAD = false> FQ_DEV void pair_build_table_windowed(const PR1& P, const PairTable& tbl, const PairLane& pl, const QuadLane& ql) {
    const PR1 twoP1 = pdbl_point(P.X, P.Y, P.Z, pl);
    const PR3 twoP = pr1_to_r3(twoP1, pl);
    PR2 t = pr1_to_r2(P, pl);
    tbl.put_entry(0, t);
#pragma unroll 1
    for (int i = 1; i < 8; i++) {
        if constexpr (QUAD) t = pr1_to_r2(qadd_signed_entry(twoP1, twoP.F, t, 0u, pl, ql), pl);
        else t = pr1_to_r2(padd_core(twoP, t, pl), pl);
        tbl.put_entry(i, t);
    }
}
// a sum of table coordinates (bound <= 4) as a bound-1 half with non-negative tight limbs of the same residue
template <int B> FQ_DEV PF<1> ptighten(const PF<B>& a) {
    const Fe<1> f = fe_unsign(reinterpret_cast<const Fe<B>&>(a));
    PF<1> r;
#pragma unroll
    for (int i = 0; i < 5; i++) { r.l[i] = f.l[i]; FQ_SIGN_UNKNOWN(r.l[i]); }
    return r;
}
// Q + (+-T): the entry's sign by masked selects behind whatever produced Q (as add_entry in curve.hip.h)
FQ_DEV PR1 padd_signed_entry(const PR1& Q, const PR2& t, u32 neg, const PairLane& pl) {
    asm("" : "+v"(neg) : "v"(Q.X.l[0]), "v"(Q.Y.l[0]), "v"(Q.Z.l[0]));
    PR2 s;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        s.N.l[k] = __builtin_amdgcn_bitop3_b32(neg, t.D.l[k], t.N.l[k], 0xCA);
        s.D.l[k] = __builtin_amdgcn_bitop3_b32(neg, t.N.l[k], t.D.l[k], 0xCA);
    }
    s.E = t.E; s.F = pcneg(t.F, neg);
    return padd_core(pr1_to_r3(Q, pl), s, pl);
}
// R2toR4(selectpt(s, T, R2neg(T))): the ladders' starting point (curve4q.py:229, :437)
FQ_DEV PR1 pair_start(const PR2& t, u32 neg) {
    PF<2> N, D;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        N.l[k] = __builtin_amdgcn_bitop3_b32(neg, t.D.l[k], t.N.l[k], 0xCA);
        D.l[k] = __builtin_amdgcn_bitop3_b32(neg, t.N.l[k], t.D.l[k], 0xCA);
    }
    PR1 Q;
    Q.X = ptighten(psub(N, D)); Q.Y = ptighten(padd(D, N)); Q.Z = ptighten(t.E);
    Q.Ta = pwiden<3>(Q.X); Q.Tb = pwiden<2>(Q.Y);
    return Q;
}
template <bool CT, bool QUAD = false> FQ_DEV PR1 pair_ladder_windowed(const WinScalar& w, const PairTable& tbl, const PairLane& pl, const QuadLane& ql) {     // curve4q.py:228-235
    u32 code = win_top_code(w);
    PR1 Q = pair_start(tbl.select_entry<CT>(code & 7), (code >> 3) - 1u);
#pragma unroll 1
    for (int i = 61; i >= 0; i--) {
        code = win_code_from_window(win_window(w, i));
        const PR2 t = tbl.select_entry<CT>(code & 7);
        if constexpr (QUAD) {
            PF<1> T;
#pragma unroll 1
            for (int k = 0; k < 3; k++) Q = qdbl_point<false>(Q.X, Q.Y, Q.Z, pl, ql, T);
            Q = qdbl_point<true>(Q.X, Q.Y, Q.Z, pl, ql, T);
            Q = qadd_signed_entry(Q, T, t, (code >> 3) - 1u, pl, ql);
        } else {
#pragma unroll 1
            for (int k = 0; k < 4; k++) Q = pdbl_point(Q.X, Q.Y, Q.Z, pl);
            Q = padd_signed_entry(Q, t, (code >> 3) - 1u, pl);
        }
    }
    return Q;
}

// ---- the DH wrapper (curve4q.py:446-462) ---------------------------------------------------------------------------------
FQ_DEV PF<1> pair_one(const PairLane& pl) { PF<1> r; r.l[0] = pl.even & 1u; r.l[1] = r.l[2] = r.l[3] = r.l[4] = 0; return r; }   // (1, 0)
template <int B> FQ_DEV void pair_canon(const PF<B>& a, u64& lo, u64& hi) { fe_canon(fe_unsign(reinterpret_cast<const Fe<B>&>(a)), lo, hi); }
FQ_DEV u32 pair_both(u32 flag) { return flag & (u32)__builtin_amdgcn_mov_dpp((int)flag, DPP_SWAP, 0xF, 0xF, true); }    // true in both halves
FQ_DEV u32 pair_point_on_curve(const PF<1>& x, const PF<1>& y, const PairLane& pl) {                // curve4q.py:23-29
    const PF<1> x2 = psqr(x, pl), y2 = psqr(y, pl);
    const PF<2> lhs = psub(y2, x2);
    const PF<2> rhs = padd(pair_one(pl), pmul(y2, pmul_const(x2, c_d(), pl), pl));
    u64 a0, a1, b0, b1;
    pair_canon(lhs, a0, a1); pair_canon(rhs, b0, b1);
    return pair_both((a0 == b0 && a1 == b1) ? 1u : 0u);
}
template <bool QUAD = false> FQ_DEV PR1 pair_clear_cofactor_392(const PF<1>& x, const PF<1>& y, const PairLane& pl, const QuadLane& ql) {   // curve4q.py:450-455
    PR1 p0;
    p0.X = x; p0.Y = y; p0.Z = pair_one(pl); p0.Ta = pwiden<3>(x); p0.Tb = pwiden<2>(y);
    const PR2 t0 = pr1_to_r2(p0, pl);
    PR1 q;
    if constexpr (QUAD) {                                                                 // the same chain on the shared-product steps
        PF<1> T;
        q = qdbl_point<true>(p0.X, p0.Y, p0.Z, pl, ql, T);
        q = qadd_signed_entry(q, T, t0, 0u, pl, ql);                                      // 3P
#pragma unroll 1
        for (int i = 0; i < 3; i++) q = qdbl_point<false>(q.X, q.Y, q.Z, pl, ql, T);
        q = qdbl_point<true>(q.X, q.Y, q.Z, pl, ql, T);                                   // 48P
        q = qadd_signed_entry(q, T, t0, 0u, pl, ql);                                      // 49P
#pragma unroll 1
        for (int i = 0; i < 3; i++) q = qdbl_point<false>(q.X, q.Y, q.Z, pl, ql, T);      // 392P
        return q;
    }
    q = padd_core(pr1_to_r3(pdbl_point(p0.X, p0.Y, p0.Z, pl), pl), t0, pl);           // 3P
#pragma unroll 1
    for (int i = 0; i < 4; i++) q = pdbl_point(q.X, q.Y, q.Z, pl);                    // 48P
    q = padd_core(pr1_to_r3(q, pl), t0, pl);                                          // 49P
#pragma unroll 1
    for (int i = 0; i < 3; i++) q = pdbl_point(q.X, q.Y, q.Z, pl);                    // 392P
    return q;
}
// R1toAffine (curve4q.py:103-106, fields.py:193-199): 1/Z = conj(Z) / (Z_re^2 + Z_im^2).  Each lane squares its half of Z, the norm
// is completed by one exchange, BOTH lanes run the GF(p) inversion chain on it (the same instructions: no cost beyond one lane's).
FQ_DEV void pair_to_affine(const PR1& Q, const PairLane& pl, PF<1>& ax, PF<1>& ay) {
    const Fe<1> zh = fe_unsign(reinterpret_cast<const Fe<1>&>(Q.Z));
    const Fe<1> sq = fe_sqr(zh);
    Fe<1> other;
#pragma unroll
    for (int i = 0; i < 5; i++) other.l[i] = (u32)__builtin_amdgcn_mov_dpp((int)sq.l[i], DPP_SWAP, 0xF, 0xF, true);
    const Fe<1> ninv = fe_inv(fe_carry(fe_add(sq, other)));
    const Fe<1> zi_abs = fe_mul(ninv, zh);                                  // |half| of conj(Z) / norm
    const Fe<1> zi_neg = fe_carry(fe_neg(zi_abs));
    PF<1> zi;
#pragma unroll
    for (int i = 0; i < 5; i++) { zi.l[i] = __builtin_amdgcn_bitop3_b32(pl.odd_neg, zi_neg.l[i], zi_abs.l[i], 0xCA); FQ_SIGN_UNKNOWN(zi.l[i]); }
    ax = pmul(Q.X, zi, pl);
    ay = pmul(Q.Y, zi, pl);
}

}  // namespace fq
