// Scalar recoding on the device: the integer side of MUL_windowed / MUL_endo.
//
//   fixed-window digits   curve4q.py:216-226   (m mod N, forced odd, 63 signed odd base-16 digits)
//   decompose(m)          curve4q.py:339-356   (4-dimensional GLV-style decomposition, mod 2^64)
//   recode(v)             curve4q.py:358-380   (65 sign bits + 65 three-bit digits)
//
// One lane owns one scalar; everything is straight-line 64-bit integer code.
#pragma once
#include "fp127.hip.h"

namespace fq {

typedef unsigned __int128 u128;

// ((a * m) >> 256) mod 2^64 for 256-bit a, m: column 4 of the product with exact carries from
// columns 0..3 (t_i of curve4q.py:344-347; only its low 64 bits matter).
FQ_DEV u64 mul_shift256(const u64 a[4], const u64 m[4]) {
    u64 w0, w1, w2, w3, w4;
    u128 t;
    // row 0
    t = (u128)a[0] * m[0];                       w0 = (u64)t;
    t = (u128)a[0] * m[1] + (u64)(t >> 64);      w1 = (u64)t;
    t = (u128)a[0] * m[2] + (u64)(t >> 64);      w2 = (u64)t;
    t = (u128)a[0] * m[3] + (u64)(t >> 64);      w3 = (u64)t; w4 = (u64)(t >> 64);
    // row 1
    t = (u128)a[1] * m[0] + w1;                  w1 = (u64)t;
    t = (u128)a[1] * m[1] + w2 + (u64)(t >> 64); w2 = (u64)t;
    t = (u128)a[1] * m[2] + w3 + (u64)(t >> 64); w3 = (u64)t;
    w4 += a[1] * m[3] + (u64)(t >> 64);
    // row 2
    t = (u128)a[2] * m[0] + w2;                  w2 = (u64)t;
    t = (u128)a[2] * m[1] + w3 + (u64)(t >> 64); w3 = (u64)t;
    w4 += a[2] * m[2] + (u64)(t >> 64);
    // row 3
    t = (u128)a[3] * m[0] + w3;                  w3 = (u64)t;
    w4 += a[3] * m[1] + (u64)(t >> 64);
    (void)w0; (void)w1; (void)w2; (void)w3;
    return w4;
}

FQ_DEV void decompose(const u64 m[4], u64 v[4]) {                        // curve4q.py:339-356
    u64 t[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u64 ell[4] = { ELL[i][0], ELL[i][1], ELL[i][2], ELL[i][3] };
        t[i] = mul_shift256(ell, m);
    }
    u64 ac[4], acp[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u64 a = (i == 0) ? m[0] : 0;
#pragma unroll
        for (int j = 0; j < 4; j++) a -= t[j] * BASIS[j][i];
        ac[i] = a + OFFSET_C[i];
        acp[i] = a + OFFSET_CP[i];
    }
    u64 mask = (u64)0 - (ac[0] & 1);            // select(ac[0] odd, ac, acp), 64-bit mask as :340-342
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = acp[i] ^ (mask & (ac[i] ^ acp[i]));
}

// recode(): sign bit of step i is bit i of `sign` (i < 64; step 64 is always positive), digit of
// step i is bit i of the planes d[0..2]; `top` is digit 64.
struct EndoDigits {
    u64 sign;
    u64 d[3];
    u32 top;
};
FQ_DEV EndoDigits recode(const u64 vin[4]) {                             // curve4q.py:358-380
    EndoDigits r;
    r.sign = vin[0] >> 1;                       // bit(v1, i+1); bit 64 of a 64-bit value is 0
    u64 v1 = vin[1], v2 = vin[2], v3 = vin[3];
    u64 p1 = 0, p2 = 0, p3 = 0;
    u64 s = r.sign;
#pragma unroll 1
    for (int i = 0; i < 64; i++) {
        u64 nb1 = ~(s >> i) & 1;                // c = (b1 | bj) ^ b1 = ~b1 & bj
        u64 b;
        b = v1 & 1; p1 |= b << i; v1 = (v1 >> 1) + (nb1 & b);
        b = v2 & 1; p2 |= b << i; v2 = (v2 >> 1) + (nb1 & b);
        b = v3 & 1; p3 |= b << i; v3 = (v3 >> 1) + (nb1 & b);
    }
    r.d[0] = p1; r.d[1] = p2; r.d[2] = p3;
    r.top = (u32)(v1 + 2 * v2 + 4 * v3);
    return r;
}
// The same digits as ONE stream of nibbles for the fused kernels' ladder (round 6: 19 -> 8 glue instructions per ladder step).  Step i
// (0..63) is nibble i % 8 of word i / 8: bits 0..2 the digit, bit 3 SET when the step SUBTRACTS (sign bit 0, the negation of `sign`
// above -- so that an arithmetic shift of the nibble's top bit is the negation mask the addition body takes).  The ladder walks the
// steps from 63 down: it takes the top nibble of word 7 and shifts left.  Built in the same pass as the planes were: no 64-bit
// variable shifts (`b << i` three times per round above), one 32-bit shift-and-or.
struct EndoNibbles {
    u32 w[8];
    u32 top;
};
FQ_DEV EndoNibbles recode_nibbles(const u64 vin[4]) {                     // curve4q.py:358-380
    EndoNibbles r;
    u64 s = vin[0] >> 1;                        // bit(v1, i+1), consumed from the bottom
    u64 v1 = vin[1], v2 = vin[2], v3 = vin[3];
#pragma unroll
    for (int t = 0; t < 8; t++) r.w[t] = 0;
#pragma unroll 1
    for (int k = 0; k < 8; k++) {
        u32 w = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const u32 nb1 = ~(u32)s & 1;        // c = (b1 | bj) ^ b1 = ~b1 & bj
            s >>= 1;
            const u32 b1 = (u32)v1 & 1, b2 = (u32)v2 & 1, b3 = (u32)v3 & 1;
            v1 = (v1 >> 1) + (nb1 & b1);
            v2 = (v2 >> 1) + (nb1 & b2);
            v3 = (v3 >> 1) + (nb1 & b3);
            w = (w >> 4) | ((b1 | (b2 << 1) | (b3 << 2) | (nb1 << 3)) << 28);
        }
#pragma unroll
        for (int t = 0; t < 7; t++) r.w[t] = r.w[t + 1];      // word k ends up in w[k] after the eighth push
        r.w[7] = w;
    }
    r.top = (u32)(v1 + 2 * v2 + 4 * v3);
    return r;
}
FQ_DEV u32 endo_digit(const EndoDigits& e, int i) {   // i in 0..63 (wave-uniform)
    return (u32)((e.d[0] >> i) & 1) | ((u32)((e.d[1] >> i) & 1) << 1) | ((u32)((e.d[2] >> i) & 1) << 2);
}
FQ_DEV u32 endo_neg_mask(const EndoDigits& e, int i) { // ~0 when the step subtracts (sign bit 0)
    return (u32)((e.sign >> i) & 1) - 1u;
}

// ---- fixed window ---------------------------------------------------------------------------------
// r = m mod N, made odd by adding N (curve4q.py:217-219).  Digit i is then
//   d_i = ((r >> 4i) & 31 | 1) - 16     (the loop at :220-222 keeps r odd, so r_{i+1} = (r_i >> 4) | 1)
// and d_62 = (r >> 252) | 1 (:223).  Digits are therefore random-access: no sequential pass.
struct WinScalar {
    u64 r[4];
};
FQ_DEV bool ge256(const u64 a[4], const u64 b[4]) {
    bool ge = true;   // compare from the least significant word up: the last difference wins
#pragma unroll
    for (int i = 0; i < 4; i++) ge = (a[i] == b[i]) ? ge : (a[i] > b[i]);
    return ge;
}
FQ_DEV WinScalar win_reduce(const u64 m[4]) {
    u64 r[4] = { m[0], m[1], m[2], m[3] };
#pragma unroll 1
    for (int k = 10; k >= 0; k--) {             // N < 2^246: restoring division by N << k
        u64 s[4];
        s[0] = ORDER_N[0] << k;
#pragma unroll
        for (int i = 1; i < 4; i++) s[i] = (ORDER_N[i] << k) | (k ? (ORDER_N[i - 1] >> (64 - k)) : 0);
        u64 keep = ge256(r, s) ? ~(u64)0 : 0;
        u64 borrow = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            u64 si = s[i] & keep;
            u64 d1 = r[i] - si;
            u64 b1 = r[i] < si;
            u64 d2 = d1 - borrow;
            u64 b2 = d1 < borrow;
            r[i] = d2; borrow = b1 | b2;
        }
    }
    u64 addn = (r[0] & 1) ? 0 : ~(u64)0;        // even -> r += N
    u64 carry = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u64 ni = ORDER_N[i] & addn;
        u64 s1 = r[i] + ni;
        u64 c1 = s1 < ni;
        u64 s2 = s1 + carry;
        u64 c2 = s2 < carry;
        r[i] = s2; carry = c1 | c2;
    }
    WinScalar w; w.r[0] = r[0]; w.r[1] = r[1]; w.r[2] = r[2]; w.r[3] = r[3];
    return w;
}
// (sgn << 3) | ind for digit i in 0..61 from the 5-bit window at bit 4i
FQ_DEV u32 win_code_from_window(u32 w5) {
    u32 w = (w5 & 31) | 1;
    u32 pos = w >> 4;                           // d > 0  <=>  window > 16
    u32 ind = ((pos ? w : ~w) & 15) >> 1;       // (|d| - 1) / 2
    return (pos << 3) | ind;
}
FQ_DEV u32 win_window(const WinScalar& w, int i) {   // bits [4i, 4i+4] of r, i wave-uniform
    int bit = 4 * i, word = bit >> 6, off = bit & 63;
    u64 lo = (word == 0) ? w.r[0] : (word == 1) ? w.r[1] : (word == 2) ? w.r[2] : w.r[3];
    u64 hi = (word == 0) ? w.r[1] : (word == 1) ? w.r[2] : (word == 2) ? w.r[3] : 0;
    u64 v = lo >> off;
    if (off > 59) v |= hi << (64 - off);
    return (u32)v & 31;
}
FQ_DEV u32 win_top_code(const WinScalar& w) {        // digit 62 = (r >> 252) | 1, always positive
    u32 d = (u32)(w.r[3] >> 60) | 1;
    return (1u << 3) | (((d - 1) >> 1) & 7);
}

// ---- fixed-base comb: mLSB-set recoding, w = 9, v = 4, e = 7, d = 28 (and 5, 5, 10, 50) --
// Shape chosen by the multiply-add count (e - 1) * 500 + (v * e - 1) * 700 under the CU's 160 KB of LDS held by ONE block of up
// to 1 024 lanes (v * 2^(w-1) points of 144 bytes): (9, 4, 7) = 6 doublings + 27 mixed additions = 21 900 with 1 024 points
// (144 KB); (10, 2, 13) 23 500; (8, 4, 8) 25 200 with 512 points; the earlier four-blocks-per-CU shape (7, 4, 9) 28 500 with
// 256 points; round 1's (5, 5, 10) 38 800 with 80 points.
// (Faz-Hernandez, Longa, Sanchez: the method the draft points to for multiplications by the generator,
// draft-ladd-cfrg-4q.md:725-729.)  For odd k < 2^250:  k = sum_{i<250} b_i 2^i with b_i in {+-1} for i < d and
// b_i in {0, b_{i mod d}} above.  Column i (0 <= i < d) carries the sign b_i and the (w-1)-bit index
// (|b_{(w-1)d+i}| ... |b_{2d+i}| |b_{d+i}|); stored as w planes of d bits.
// Two shapes share one table object: the fast one for selection by address and a small one for the constant-time mode, whose
// additions read a whole block of 2^(w-1) entries (35 scans of 64 entries cost more than 49 scans of 16: DESIGN.md section 10).
template <int W_, int V_, int E_> struct CombShape {
    static constexpr int W = W_, V = V_, E = E_, D = V_ * E_, BLOCK_POINTS = 1 << (W_ - 1), POINTS = V_ << (W_ - 1);
    static_assert(W_ * V_ * E_ >= 250 && V_ * E_ <= 64, "comb shape");
};
typedef CombShape<9, 4, 7> CombFast;       // 1 024 points: 6 doublings + 27 mixed additions
typedef CombShape<5, 5, 10> CombScan;      // 80 points: 9 doublings + 49 mixed additions, 16-entry blocks
constexpr int COMB_POINTS_ALL = CombFast::POINTS + CombScan::POINTS;       // the table object: fast points first
template <typename S> struct CombDigits {
    typedef typename std::conditional<(S::D <= 32), u32, u64>::type plane_t;      // d bits per plane: one register when they fit
    plane_t plane[S::W];     // plane[0] bit i: b_i == +1 ; plane[r] bit i: |b_{r d + i}|
    u32 negate;          // ~0 when the scalar was replaced by N - k (even k): the result is negated
};
template <typename S> FQ_DEV CombDigits<S> comb_recode(const u64 m[4]) {
    constexpr int COMB_W = S::W, COMB_D = S::D;
    WinScalar red = win_reduce(m);                    // k mod N made odd by ADDING N when even (curve4q.py:217-219) ...
    // ... the comb wants odd k < N instead: undo the +N and use N - k, negating the result
    u64 k[4] = { red.r[0], red.r[1], red.r[2], red.r[3] };
    // win_reduce returns r = (m mod N) if that is odd, else (m mod N) + N.  Recover parity of (m mod N):
    // (m mod N) + N >= N, and an odd (m mod N) is < N, so "k >= N" identifies the even case exactly.
    u64 n[4] = { ORDER_N[0], ORDER_N[1], ORDER_N[2], ORDER_N[3] };
    const bool was_even = ge256(k, n);
    // even case: k_red = k - N (even), want N - k_red = 2N - k
    u64 twoN[4], alt[4];
    twoN[0] = n[0] << 1; twoN[1] = (n[1] << 1) | (n[0] >> 63); twoN[2] = (n[2] << 1) | (n[1] >> 63); twoN[3] = (n[3] << 1) | (n[2] >> 63);
    u64 borrow = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u64 d1 = twoN[i] - k[i], b1 = twoN[i] < k[i];
        u64 d2 = d1 - borrow, b2 = d1 < borrow;
        alt[i] = d2; borrow = b1 | b2;
    }
    CombDigits<S> c;
    c.negate = was_even ? ~0u : 0u;
#pragma unroll
    for (int i = 0; i < 4; i++) k[i] = was_even ? alt[i] : k[i];        // odd, in [1, N]
    const u64 mask_d = (1ull << COMB_D) - 1;
    const u64 sign = ((k[0] >> 1) & (mask_d >> 1)) | (1ull << (COMB_D - 1));     // b_i = +1  <=>  bit set
    typedef typename CombDigits<S>::plane_t plane_t;
    c.plane[0] = (plane_t)sign;
    // carry word c = k >> d
    u64 c0 = (k[0] >> COMB_D) | (k[1] << (64 - COMB_D)), c1 = (k[1] >> COMB_D) | (k[2] << (64 - COMB_D));
    u64 c2 = (k[2] >> COMB_D) | (k[3] << (64 - COMB_D)), c3 = k[3] >> COMB_D;
    // Plane r holds the low d digits of the running carry word c in the signed-digit form prescribed by the signs: bits T with
    //     sum_i T_i b_i 2^i == c (mod 2^d),   then   c <- (c - sum_i T_i b_i 2^i) / 2^d        (exact).
    // With M = the positions where b_i = -1 this reads T - 2 (T & M) == c (mod 2^d).  Bit i of the right-hand side of
    //     T = c + 2 (T & M)  (mod 2^d)
    // depends on bits < i of T only, so iterating that assignment d times from any start fixes one more bit per round: three
    // register-wide operations per digit instead of a 256-bit shift-and-add (the bit-serial form of the method cost 30
    // instructions per digit, 14 % of the comb kernel's instruction count).  Fixed trip counts: nothing depends on the scalar.
    const plane_t neg = (plane_t)(~sign & mask_d);
    // unrolled in full: with a run-time r the compiler keeps plane[] in scratch memory and indexes it dynamically (44-64 bytes
    // of scratch per lane in comb_kernel, the one hot kernel that had any)
#pragma unroll
    for (int r = 1; r < COMB_W; r++) {
        const plane_t low = (plane_t)(c0 & mask_d);
        plane_t T = low;
#pragma unroll 4
        for (int i = 0; i < COMB_D; i++) T = (plane_t)((low + (plane_t)((T & neg) << 1)) & (plane_t)mask_d);
        c.plane[r] = T;
        const u64 up = (u64)(T & neg) << 1, down = (u64)T;       // c <- (c + 2 (T & M) - T) >> d
        u64 s0 = c0 + up, cy = s0 < up;
        u64 s1 = c1 + cy; cy = s1 < cy;
        u64 s2 = c2 + cy; cy = s2 < cy;
        u64 s3 = c3 + cy;
        u64 t0 = s0 - down, bw = s0 < down;
        u64 t1 = s1 - bw; bw = s1 < bw;
        u64 t2 = s2 - bw; bw = s2 < bw;
        u64 t3 = s3 - bw;
        c0 = (t0 >> COMB_D) | (t1 << (64 - COMB_D)); c1 = (t1 >> COMB_D) | (t2 << (64 - COMB_D));
        c2 = (t2 >> COMB_D) | (t3 << (64 - COMB_D)); c3 = t3 >> COMB_D;
    }
    return c;
}
template <typename S> FQ_DEV u32 comb_index(const CombDigits<S>& c, int col) {    // (w-1)-bit table index of column `col` (wave-uniform col)
    u32 idx = 0;
#pragma unroll
    for (int r = 1; r < S::W; r++) idx |= (u32)((c.plane[r] >> col) & 1) << (r - 1);
    return idx;
}
template <typename S> FQ_DEV u32 comb_neg_mask(const CombDigits<S>& c, int col) { return (u32)((c.plane[0] >> col) & 1) - 1u; }
// point t of shape S's sub-table: [2^(e j) (1 + u0 2^d + u1 2^2d + ...)] B for t = (j, u); the scalar as 256 bits
template <typename S> FQ_DEV void comb_point_scalar(u32 t, u64 m[4]) {
    const u32 j = t >> (S::W - 1), u = t & ((1u << (S::W - 1)) - 1);
    m[0] = m[1] = m[2] = m[3] = 0;
    auto set_bit = [&](int bit) { m[bit >> 6] |= 1ull << (bit & 63); };
    set_bit(S::E * j);
    for (int r = 0; r < S::W - 1; r++) if ((u >> r) & 1) set_bit(S::E * j + (r + 1) * S::D);
}

}  // namespace fq
