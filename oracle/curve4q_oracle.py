"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the FourQ (Curve4Q) scalar-multiplication path.

A Python 3 restatement, on plain big integers, of the algorithms of the reference
implementation (bifurcation/fourq, `impl/fields.py` and `impl/curve4q.py`).  Every function
cites the reference lines it follows.  It exists to CHECK the HIP engine: only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it.  The product
package `fourq_amd` never imports anything from `oracle/`.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks this file against
  * every literal known-answer vector the reference's self-tests hold (curve4q.py:473-778,
    fields.py:366-403), committed in tests/golden/kat.json, and
  * seeded input/output vectors produced by running the reference itself in the build
    container (tests/golden/make_golden.py -> tests/golden/*.json).

Value conventions are the reference's: a GF(p) element is an int in [0, p); a GF(p^2) element is
a pair (re, im); points are tuples of pairs (R1: X,Y,Z,Ta,Tb; R2: X+Y,Y-X,2Z,2dT;
R3: X+Y,Y-X,Z,T; R4: X,Y,Z); scalars are non-negative ints.
"""

P127 = (1 << 127) - 1  # fields.py:5
MASK64 = (1 << 64) - 1
MASK512 = (1 << 512) - 1  # fields.py:7

# ---------------------------------------------------------------------------------------------
# GF(p), p = 2^127 - 1                                                    fields.py:9-132
# ---------------------------------------------------------------------------------------------


def fp_add(x, y):  # fields.py:30-33
    return (x + y) % P127


def fp_sub(x, y):  # fields.py:36-39
    return (x - y) % P127


def fp_mul(x, y):  # fields.py:42-45
    return (x * y) % P127


def fp_sqr(x):  # fields.py:48-51
    return (x * x) % P127


def fp_neg(x):  # fields.py:54-57
    return (P127 - x) % P127


def fp_select(c, x, y):  # fields.py:60-64  (x if c == 1 else y, via an all-ones mask)
    return y ^ ((MASK512 * c) & (x ^ y))


def _sqr_n(x, n):
    for _ in range(n):
        x = x * x % P127
    return x


def fp_inv(x):
    """x^(2^127 - 3) by the reference's fixed addition chain (fields.py:67-106)."""
    x2 = fp_mul(x, fp_sqr(x))  # 2^2 - 1
    x4 = fp_mul(x2, _sqr_n(x2, 2))  # 2^4 - 1
    x8 = fp_mul(x4, _sqr_n(x4, 4))  # 2^8 - 1
    x16 = fp_mul(x8, _sqr_n(x8, 8))  # 2^16 - 1
    x32 = fp_mul(x16, _sqr_n(x16, 16))  # 2^32 - 1
    t = fp_mul(_sqr_n(x32, 32), x32)  # 2^64 - 1
    t = fp_mul(_sqr_n(t, 32), x32)  # 2^96 - 1
    t = fp_mul(_sqr_n(t, 16), x16)  # 2^112 - 1
    t = fp_mul(_sqr_n(t, 8), x8)  # 2^120 - 1
    t = fp_mul(_sqr_n(t, 4), x4)  # 2^124 - 1
    t = fp_mul(fp_sqr(t), x)  # 2^125 - 1
    return fp_mul(_sqr_n(t, 2), x)  # 2^127 - 3


def fp_invsqrt(x):
    """x^(2^125 - 1) (fields.py:110-122); only the wire format (section 8f row 1) needs it."""
    x31 = pow(x, 31, P127)
    acc = cur = x31
    for _ in range(24):
        cur = _sqr_n(cur, 5)
        acc = fp_mul(cur, acc)
    return acc


# ---------------------------------------------------------------------------------------------
# GF(p^2) = GF(p)[i]/(i^2+1)                                              fields.py:134-238
# ---------------------------------------------------------------------------------------------

F2_ZERO, F2_ONE, F2_TWO = (0, 0), (1, 0), (2, 0)  # fields.py:140-142


def f2_add(a, b):  # fields.py:157-159
    return ((a[0] + b[0]) % P127, (a[1] + b[1]) % P127)


def f2_sub(a, b):  # fields.py:162-164
    return ((a[0] - b[0]) % P127, (a[1] - b[1]) % P127)


def f2_mul(a, b):  # fields.py:167-173 (schoolbook, one reduction per component)
    return ((a[0] * b[0] - a[1] * b[1]) % P127, (a[0] * b[1] + a[1] * b[0]) % P127)


def f2_sqr(a):  # fields.py:176-181
    return ((a[0] * a[0] - a[1] * a[1]) % P127, (2 * a[0] * a[1]) % P127)


def f2_neg(a):  # fields.py:184-186
    return (-a[0] % P127, -a[1] % P127)


def f2_conj(a):  # fields.py:189-191
    return (a[0], -a[1] % P127)


def f2_inv(a):  # fields.py:194-199   conj(a) / (a0^2 + a1^2)
    n = fp_inv(fp_add(fp_sqr(a[0]), fp_sqr(a[1])))
    return f2_mul((n, 0), f2_conj(a))


def f2_select(c, x, y):  # fields.py:237-238
    return (fp_select(c, x[0], y[0]), fp_select(c, x[1], y[1]))


def f2_invsqrt(a):  # fields.py:202-230, off the hot path and unused by the reference itself
    """Behaviour restated as written: the `== -1` tests (fields.py:217, :223) can never hold for residues in
    [0, p), so non-squares are not rejected and the second root choice is never taken."""
    if a[1] == 0:
        t = fp_invsqrt(a[0])
        return (t, 0) if fp_mul(a[0], fp_sqr(t)) == 1 else (0, t)
    n = fp_add(fp_sqr(a[0]), fp_sqr(a[1]))
    s = fp_invsqrt(n)
    c = fp_mul(n, s)
    half = 1 << 126
    delta = fp_mul(fp_add(a[0], c), half)
    g = fp_invsqrt(delta)
    h = fp_mul(delta, g)
    return (fp_mul(h, s), fp_neg(fp_mul(fp_mul(fp_mul(a[1], s), g), half)))


class GFp:
    """Reference-shaped namespace (fields.py:9) so tests read like the reference's own."""

    half = 1 << 126
    add, sub, mul, sqr, neg = map(staticmethod, (fp_add, fp_sub, fp_mul, fp_sqr, fp_neg))
    select, inv, invsqrt = map(staticmethod, (fp_select, fp_inv, fp_invsqrt))


class GFp2:
    """Reference-shaped namespace (fields.py:134)."""

    zero, one, two = F2_ZERO, F2_ONE, F2_TWO
    add, sub, mul, sqr, neg = map(staticmethod, (f2_add, f2_sub, f2_mul, f2_sqr, f2_neg))
    conj, inv, select, invsqrt = map(staticmethod, (f2_conj, f2_inv, f2_select, f2_invsqrt))


# ---------------------------------------------------------------------------------------------
# Curve constants                                  curve4q.py:9-20 (= draft-ladd-cfrg-4q.md:771-790)
# ---------------------------------------------------------------------------------------------

d = (0xE40000000000000142, 0x5E472F846657E0FCB3821488F1FC0C8D)
N = 0x29CBC14E5E0A72F05397829CBC14E5DFBD004DFE0F79992FB2540EC7768CE7
Ox, Oy = (0, 0), (1, 0)
Gx = (0x1A3472237C2FB305286592AD7B3833AA, 0x1E1F553F2878AA9C96869FB360AC77F6)
Gy = (0x0E3FEE9BA120785AB924A2462BCBB287, 0x6E1C4AF8630E024249A7C344844C8B5C)
TWO_D = f2_mul(F2_TWO, d)  # the 2d that curve4q.py:115 recomputes on every call


def PointOnCurve(Pt):  # curve4q.py:23-29     -x^2 + y^2 == 1 + d x^2 y^2
    x2, y2 = f2_sqr(Pt[0]), f2_sqr(Pt[1])
    return f2_sub(y2, x2) == f2_add(F2_ONE, f2_mul(f2_mul(d, x2), y2))


# ---------------------------------------------------------------------------------------------
# Representations and the group law                                        curve4q.py:100-175
# ---------------------------------------------------------------------------------------------


def AffineToR1(x, y):  # curve4q.py:100-101
    return (x, y, F2_ONE, x, y)


def R1toAffine(Pt):  # curve4q.py:103-106
    zi = f2_inv(Pt[2])
    return (f2_mul(Pt[0], zi), f2_mul(Pt[1], zi))


def R1toR2(Pt):  # curve4q.py:109-116
    X, Y, Z, Ta, Tb = Pt
    return (f2_add(X, Y), f2_sub(Y, X), f2_add(Z, Z), f2_mul(TWO_D, f2_mul(Ta, Tb)))


def R1toR3(Pt):  # curve4q.py:119-126
    X, Y, Z, Ta, Tb = Pt
    return (f2_add(X, Y), f2_sub(Y, X), Z, f2_mul(Ta, Tb))


def R2toR4(Pt):  # curve4q.py:129-135 -- follows the code ((N-D, D+N, E)), not its docstring
    return (f2_sub(Pt[0], Pt[1]), f2_add(Pt[1], Pt[0]), Pt[2])


def R2neg(Pt):  # local helper of MUL_* (curve4q.py:193-195, :410-412)
    return (Pt[1], Pt[0], Pt[2], f2_neg(Pt[3]))


def DBL(Pt):  # curve4q.py:138-152 ; reads only X,Y,Z
    X, Y, Z = Pt[0], Pt[1], Pt[2]
    xx, yy = f2_sqr(X), f2_sqr(Y)
    zz2 = f2_mul(F2_TWO, f2_sqr(Z))
    s = f2_add(xx, yy)
    e = f2_sub(f2_sqr(f2_add(X, Y)), s)
    f = f2_sub(yy, xx)
    g = f2_sub(zz2, f)
    return (f2_mul(e, g), f2_mul(s, f), f2_mul(f, g), e, s)


def ADD_core(Pt, Qt):  # curve4q.py:155-171 ; R3 + R2 -> R1
    a = f2_mul(Pt[1], Qt[1])
    b = f2_mul(Pt[0], Qt[0])
    c = f2_mul(Qt[3], Pt[3])
    dd = f2_mul(Qt[2], Pt[2])
    e, h = f2_sub(b, a), f2_add(b, a)
    f, g = f2_sub(dd, c), f2_add(dd, c)
    return (f2_mul(e, f), f2_mul(g, h), f2_mul(f, g), e, h)


def ADD(Pt, Qt):  # curve4q.py:174-175 ; R1 + R2 -> R1
    return ADD_core(R1toR3(Pt), Qt)


def selectpt(c, P1, P2):  # curve4q.py:198-206, :415-423
    return tuple(f2_select(c, u, v) for u, v in zip(P1, P2))


# ---------------------------------------------------------------------------------------------
# Fixed-window scalar multiplication                                       curve4q.py:179-235
# ---------------------------------------------------------------------------------------------


def table_windowed(Pt):  # curve4q.py:179-185 ; [1]P,[3]P,...,[15]P in R2
    twoP = DBL(Pt)
    T = [R1toR2(Pt)]
    while len(T) < 8:
        T.append(R1toR2(ADD(twoP, T[-1])))
    return T


def recode_windowed(m):
    """Digits of MUL_windowed (curve4q.py:216-226): returns (sgn[63], ind[63])."""
    r = m % N
    if r % 2 == 0:
        r += N
    digits = []
    for _ in range(63):
        di = (r % 32) - 16
        digits.append(di)
        r = (r - di) // 16
    digits[62] = r  # curve4q.py:223
    ind = [(abs(di) - 1) // 2 for di in digits]
    sgn = [1 if di > 0 else 0 for di in digits]
    return sgn, ind


def MUL_windowed(m, Pt, table=None):  # curve4q.py:188-235
    (X, Y, Z, Ta, Tb) = Pt  # shape check, as curve4q.py:190
    T = table if table else table_windowed(Pt)
    nT = [R2neg(t) for t in T]
    sgn, ind = recode_windowed(m)
    Q = R2toR4(selectpt(sgn[62], T[ind[62]], nT[ind[62]]))
    for i in range(61, -1, -1):
        Q = DBL(DBL(DBL(DBL(Q))))
        Q = ADD(Q, selectpt(sgn[i], T[ind[i]], nT[ind[i]]))
    return Q


# ---------------------------------------------------------------------------------------------
# Endomorphisms                      curve4q.py:240-322 (constants = draft-ladd-cfrg-4q.md:792-812)
# ---------------------------------------------------------------------------------------------

ctau = (0x1964DE2C3AFAD20C74DCD57CEBCE74C3, 0x000000000000000C0000000000000012)
ctaudual = (0x4AA740EB230586529ECAA6D9DECDF034, 0x7FFFFFFFFFFFFFF40000000000000011)
cphi = (
    (0x0000000000000005FFFFFFFFFFFFFFF7, 0x2553A0759182C3294F65536CEF66F81A),
    (0x00000000000000050000000000000007, 0x62C8CAA0C50C62CF334D90E9E28296F9),
    (0x000000000000000F0000000000000015, 0x78DF262B6C9B5C982C2CB7154F1DF391),
    (0x00000000000000020000000000000003, 0x5084C6491D76342A92440457A7962EA4),
    (0x00000000000000030000000000000003, 0x12440457A7962EA4A1098C923AEC6855),
    (0x000000000000000A000000000000000F, 0x459195418A18C59E669B21D3C5052DF3),
    (0x00000000000000120000000000000018, 0x0B232A8314318B3CCD3643A78A0A5BE7),
    (0x00000000000000180000000000000023, 0x3963BC1C99E2EA1A66C183035F48781A),
    (0x00000000000000AA00000000000000F0, 0x1F529F860316CBE544E251582B5D0EF0),
    (0x00000000000008700000000000000BEF, 0x0FD52E9CFE00375B014D3E48976E2505),
)
cpsi = (
    None,
    (0x2AF99E9A83D54A02EDF07F4767E346EF, 0x00000000000000DE000000000000013A),
    (0x00000000000000E40000000000000143, 0x21B8D07B99A81F034C7DEB770E03F372),
    (0x00000000000000060000000000000009, 0x4CB26F161D7D69063A6E6ABE75E73A61),
    (0x7FFFFFFFFFFFFFF9FFFFFFFFFFFFFFF6, 0x334D90E9E28296F9C59195418A18C59E),
)


def tau(Pt):  # curve4q.py:258-267 ; E -> E_hat (projective triple)
    X, Y, Z = Pt
    xx, yy = f2_sqr(X), f2_sqr(Y)
    s, df = f2_add(xx, yy), f2_sub(xx, yy)
    x2 = f2_mul(f2_mul(f2_mul(ctau, X), Y), df)
    y2 = f2_neg(f2_mul(f2_add(f2_mul(F2_TWO, f2_sqr(Z)), df), s))
    return (x2, y2, f2_mul(s, df))


def tau_dual(Pt):  # curve4q.py:269-280 ; E_hat -> E, R1 output
    X, Y, Z = Pt
    xx, yy = f2_sqr(X), f2_sqr(Y)
    s, ta = f2_add(xx, yy), f2_sub(yy, xx)
    w = f2_sub(f2_mul(F2_TWO, f2_sqr(Z)), ta)
    tb = f2_mul(f2_mul(ctaudual, X), Y)
    return (f2_mul(tb, s), f2_mul(ta, w), f2_mul(s, w), ta, tb)


def upsilon(Pt):  # curve4q.py:282-302
    X, Y, Z = Pt
    a = f2_mul(f2_mul(cphi[0], X), Y)
    b = f2_mul(Y, Z)
    c, dz = f2_sqr(Y), f2_sqr(Z)
    f, g, h = f2_sqr(dz), f2_sqr(b), f2_sqr(c)
    i = f2_mul(cphi[1], b)
    j = f2_add(c, f2_mul(cphi[2], dz))
    k = f2_add(f2_add(f2_mul(cphi[8], g), h), f2_mul(cphi[9], f))
    x2 = f2_mul(f2_add(i, j), f2_sub(i, j))
    x2 = f2_conj(f2_mul(f2_mul(a, k), x2))
    l = f2_add(c, f2_mul(cphi[4], dz))
    mm = f2_mul(cphi[3], b)
    n = f2_mul(f2_add(l, mm), f2_sub(l, mm))
    y2 = f2_add(f2_add(h, f2_mul(cphi[6], g)), f2_mul(cphi[7], f))
    y2 = f2_conj(f2_mul(f2_mul(f2_mul(cphi[5], dz), n), y2))
    z2 = f2_conj(f2_mul(f2_mul(b, k), n))
    return (x2, y2, z2)


def chi(Pt):  # curve4q.py:304-316
    X, Y, Z = Pt
    a, b = f2_conj(X), f2_conj(Y)
    c = f2_sqr(f2_conj(Z))
    dd, f = f2_sqr(a), f2_sqr(b)
    g = f2_mul(b, f2_add(dd, f2_mul(cpsi[2], c)))
    h = f2_neg(f2_add(dd, f2_mul(cpsi[4], c)))
    x2 = f2_mul(f2_mul(f2_mul(cpsi[1], a), c), h)
    y2 = f2_mul(g, f2_add(dd, f2_mul(cpsi[3], c)))
    return (x2, y2, f2_mul(g, h))


def phi(Pt):  # curve4q.py:318-319
    return tau_dual(upsilon(tau(Pt[:3])))


def psi(Pt):  # curve4q.py:321-322
    return tau_dual(chi(tau(Pt[:3])))


# ---------------------------------------------------------------------------------------------
# Scalar decomposition and recoding      curve4q.py:326-380 (constants = draft:816-830, :573-574)
# ---------------------------------------------------------------------------------------------

BASIS = (
    (0x0906FF27E0A0A196, -0x1363E862C22A2DA0, 0x07426031ECC8030F, -0x084F739986B9E651),
    (0x1D495BEA84FCC2D4, -0x0000000000000001, 0x0000000000000001, 0x25DBC5BC8DD167D0),
    (0x17ABAD1D231F0302, 0x02C4211AE388DA51, -0x2E4D21C98927C49F, 0x0A9E6F44C02ECD97),
    (0x136E340A9108C83F, 0x3122DF2DC3E0FF32, -0x068A49F02AA8A9B5, -0x18D5087896DE0AEA),
)
ELL = (
    0x7FC5BB5C5EA2BE5DFF75682ACE6A6BD66259686E09D1A7D4F,
    0x38FD4B04CAA6C0F8A2BD235580F468D8DD1BA1D84DD627AFB,
    0x0D038BF8D0BFFBAF6C42BD6C965DCA9029B291A33678C203C,
    0x31B073877A22D841081CBDC3714983D8212E5666B77E7FDC0,
)
OFFSET_C = tuple(5 * BASIS[1][i] - 3 * BASIS[2][i] + 2 * BASIS[3][i] for i in range(4))
OFFSET_CP = tuple(OFFSET_C[i] + BASIS[3][i] for i in range(4))


def decompose(m):  # curve4q.py:339-356
    t = [(ell * m) >> 256 for ell in ELL]
    a = [(m if i == 0 else 0) - sum(t[j] * BASIS[j][i] for j in range(4)) for i in range(4)]
    ac = [a[i] + OFFSET_C[i] for i in range(4)]
    acp = [a[i] + OFFSET_CP[i] for i in range(4)]
    s = ac[0] % 2
    # the reference's select uses a 64-bit mask (curve4q.py:340-342)
    return [acp[i] ^ ((MASK64 * s) & (ac[i] ^ acp[i])) for i in range(4)]


def recode(v):  # curve4q.py:358-380 ; returns (sign bits m[0..64], digits d[0..64])
    vv = list(v)
    signs, digits = [], []
    for i in range(64):
        b1 = (vv[0] >> (i + 1)) & 1
        dig = 0
        for j in (1, 2, 3):
            bj = vv[j] & 1
            dig += bj << (j - 1)
            vv[j] = (vv[j] >> 1) + ((b1 | bj) ^ b1)
        signs.append(b1)
        digits.append(dig)
    digits.append(vv[1] + 2 * vv[2] + 4 * vv[3])
    signs.append(1)
    return signs, digits


# ---------------------------------------------------------------------------------------------
# Endomorphism-accelerated scalar multiplication                           curve4q.py:385-442
# ---------------------------------------------------------------------------------------------


def table_endo(Pt):  # curve4q.py:385-403 ; T[k] = P + k0*phi(P) + k1*psi(P) + k2*psi(phi(P))
    Q = phi(Pt)
    R = psi(Pt)
    S = psi(Q)
    Q3, R3, S3 = R1toR3(Q), R1toR3(R), R1toR3(S)
    T = [R1toR2(Pt)]
    T.append(R1toR2(ADD_core(Q3, T[0])))
    T.append(R1toR2(ADD_core(R3, T[0])))
    T.append(R1toR2(ADD_core(R3, T[1])))
    for k in range(4):
        T.append(R1toR2(ADD_core(S3, T[k])))
    return T


def MUL_endo(m, Pt, table=None):  # curve4q.py:405-442 ; m used unreduced
    (X, Y, Z, Ta, Tb) = Pt  # shape check, as curve4q.py:407
    T = table if table else table_endo(Pt)
    nT = [R2neg(t) for t in T]
    s, dg = recode(decompose(m))
    Q = R2toR4(selectpt(s[64], T[dg[64]], nT[dg[64]]))
    for i in range(63, -1, -1):
        Q = DBL(Q)
        Q = ADD(Q, selectpt(s[i], T[dg[i]], nT[dg[i]]))
    return Q


# ---------------------------------------------------------------------------------------------
# Diffie-Hellman                                                           curve4q.py:446-468
# ---------------------------------------------------------------------------------------------


def clear_cofactor(P0):
    """[392]P by the reference's fixed chain (curve4q.py:450-455)."""
    T0 = R1toR2(P0)
    Q = ADD(DBL(P0), T0)  # 3P
    Q = DBL(DBL(DBL(DBL(Q))))  # 48P
    Q = ADD(Q, T0)  # 49P
    return DBL(DBL(DBL(Q)))  # 392P


def DH_core(m, Pt, mul, table=None):  # curve4q.py:446-462
    if not PointOnCurve(Pt):
        raise Exception("Point not on curve")
    Q = clear_cofactor(AffineToR1(Pt[0], Pt[1]))
    Q = R1toAffine(mul(m, Q, table=table))
    if Q == (Ox, Oy):
        raise Exception("DH computation resulted in neutral point")
    return Q


def DH_windowed(m, Pt, table=None):  # curve4q.py:464-465
    return DH_core(m, Pt, MUL_windowed, table=table)


def DH_endo(m, Pt, table=None):  # curve4q.py:467-468
    return DH_core(m, Pt, MUL_endo, table=table)


# BASELINE.json spellings
mul = MUL_endo
mul_windowed = MUL_windowed


def dh_exchange(a, b, table=None):
    """One exchange = DH_endo(a, DH_endo(b, G)) (the pattern of curve4q.py:731).

    `table`, when given, is table_endo([392]G) and accelerates the fixed-base half."""
    return DH_endo(a, DH_endo(b, (Gx, Gy), table=table))


# ---------------------------------------------------------------------------------------------
# Point compression                              curve4q.py:33-96
# ---------------------------------------------------------------------------------------------


def sign(x):  # curve4q.py:33-39
    return (x[0] >> 126) if x[0] != 0 else (x[1] >> 126)


def encode(x, y):  # curve4q.py:41-46 ; 32 bytes: y0 | y1, sign bit of x in the top bit
    out = bytearray(y[0].to_bytes(16, "little") + y[1].to_bytes(16, "little"))
    out[31] |= sign(x) << 7  # the reference sets y1[15], i.e. byte 31
    return out


def decode(B):
    """32 bytes -> affine (x, y), as the reference's decode (curve4q.py:49-96), exceptions included.

    Differences of form only: the input is not modified (the reference clears the sign bit in the caller's
    bytearray, curve4q.py:56) and any bytes-like object is accepted.  The branch at curve4q.py:76-77 refers to
    a GFp.two that does not exist; the reference therefore raises AttributeError whenever t == 0 (e.g. for the
    encoding of the neutral point), and so does this restatement, with the same message."""
    B = bytearray(B)
    if len(B) != 32:
        raise Exception("Malformed point: length {} != 32".format(len(B)))
    if B[15] & 0x80:
        raise Exception("Malformed point: reserved bit is not zero")
    s = B[31] >> 7
    B[31] &= 0x7F
    y0 = int.from_bytes(B[:16], "little") & P127        # fromLittleEndian masks bit 127 (fields.py:129-132)
    y1 = int.from_bytes(B[16:], "little") & P127
    if y0 >= P127 or y1 >= P127:
        raise Exception("Malformed point: reserved bit is not zero")
    y = (y0, y1)
    y2 = f2_sqr(y)
    u0, u1 = f2_sub(y2, F2_ONE)
    v0, v1 = f2_add(f2_mul(d, y2), F2_ONE)
    t0 = fp_add(fp_mul(u0, v0), fp_mul(u1, v1))
    t1 = fp_sub(fp_mul(u1, v0), fp_mul(u0, v1))
    t2 = fp_add(fp_sqr(v0), fp_sqr(v1))
    t3 = fp_add(fp_sqr(t0), fp_sqr(t1))
    t3 = fp_mul(fp_invsqrt(t3), t3)
    t = fp_mul(2, fp_add(t0, t3))
    if t == 0:
        raise AttributeError("type object 'GFp' has no attribute 'two'")   # curve4q.py:77
    a = fp_invsqrt(fp_mul(t, fp_mul(t2, fp_sqr(t2))))
    b = fp_mul(fp_mul(a, t2), t)
    x0 = fp_mul(b, GFp.half)
    x1 = fp_mul(fp_mul(a, t2), t1)
    if t != fp_mul(t2, fp_sqr(b)):
        x0, x1 = x1, x0
    x = (x0, x1)
    if sign(x) != s:
        x = f2_neg(x)
    if not PointOnCurve((x, y)):
        x = f2_conj(x)
    if not PointOnCurve((x, y)):
        raise Exception("Point not on curve")
    return (x, y)


def kat_scalars(count=1000):
    """The deterministic scalar sequence of the reference's test_mul (curve4q.py:552-559)."""
    s = [0x3AD457AB55456230, 0x3A8B3C2C6FD86E0C, 0x7E38F7C9CFBB9166, 0x0028FD6CBDA458F0]
    out = []
    for _ in range(count):
        s[1] = s[2]
        s[2] = (s[2] + s[0]) & MASK64
        out.append(s[0] | (s[1] << 64) | (s[2] << 128) | (s[3] << 192))
    return out
