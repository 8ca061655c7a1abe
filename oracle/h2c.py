"""CPU restatement of the reference's hash-to-curve path for bls12-381 G1/G2 (TEST
INFRASTRUCTURE ONLY).  Follows src/abstract/hash-to-curve.ts of paulmillr/noble-curves
(expand_message_xmd :189-228, hash_to_field :312-378, isogenyMap :381-410, createHasher :441-548,
SWUFpSqrtRatio :552-651, mapToCurveSimpleSWU :652-717) and the suite set-up of
src/bls12-381.ts (:305-313 hasher options, :560-619 clearCofactor, :668-862 isogenies and SWU
parameters).  Constants come from tools/h2c_constants.json (extracted from the reference).
Pinned by tests/test_oracle_golden.py against eip2537.json (mapToCurve) and the reference's
priv:msg:sig signature vectors (hashToCurve end to end).
"""
import hashlib
import json
import os

from .curves import BLS_P, BlsG1, BlsG2
from .field import FpInvertBatch
from .weierstrass import BLS_X, bls_g2_psi

_K = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "h2c_constants.json")))

DST_G2 = b"BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_NUL_"     # bls12-381.ts:305-313
DST_G1 = b"BLS_SIG_BLS12381G1_XMD:SHA-256_SSWU_RO_NUL_"     # bls12-381.ts:628-633


def expand_message_xmd(msg, DST, lenInBytes, H=hashlib.sha256):
    """hash-to-curve.ts:189-228 (RFC 9380 5.3.1)."""
    if len(DST) > 255:
        DST = H(b"H2C-OVERSIZE-DST-" + DST).digest()
    b_in_bytes, r_in_bytes = H().digest_size, H().block_size
    ell = -(-lenInBytes // b_in_bytes)
    if lenInBytes > 65535 or ell > 255:
        raise ValueError("expand_message_xmd: invalid lenInBytes")
    DST_prime = DST + bytes([len(DST)])
    b_0 = H(bytes(r_in_bytes) + msg + lenInBytes.to_bytes(2, "big") + b"\x00" + DST_prime).digest()
    b = [H(b_0 + b"\x01" + DST_prime).digest()]
    for i in range(1, ell):
        b.append(H(bytes(x ^ y for x, y in zip(b_0, b[i - 1])) + bytes([i + 1]) + DST_prime).digest())
    return b"".join(b)[:lenInBytes]


def hash_to_field(msg, count, p, m, k, DST):
    """hash-to-curve.ts:312-378 with expand = 'xmd', hash = sha256."""
    L = -(-(p.bit_length() + k) // 8)
    prb = expand_message_xmd(msg, DST, count * m * L)
    return [[int.from_bytes(prb[L * (j + i * m):L * (j + i * m) + L], "big") % p for j in range(m)]
            for i in range(count)]


class _F1:
    """IField view of Fp used by the SWU code (cmov/isOdd/pow on ints)."""
    def __init__(self, F):
        self.F, self.ORDER, self.ONE, self.ZERO = F, F.ORDER, 1, 0
        for n in ("add", "sub", "mul", "sqr", "neg", "inv", "eql", "is0", "sqrt"):
            setattr(self, n, getattr(F, n))

    def pow(self, a, e):
        return pow(a, e, self.ORDER)

    def isOdd(self, a):
        return a & 1 == 1

    def invertBatch(self, nums, pz):
        return FpInvertBatch(self.F, nums, pz)


class _F2:
    def __init__(self, F):
        self.F, self.ORDER, self.ONE, self.ZERO = F, F.ORDER, (1, 0), (0, 0)
        for n in ("add", "sub", "mul", "sqr", "neg", "inv", "eql", "is0", "sqrt", "pow"):
            setattr(self, n, getattr(F, n))

    def isOdd(self, x):                      # tower.ts:502-509 (sgn0_m_eq_2)
        x0, x1 = x
        return bool((x0 & 1) or (x0 == 0 and (x1 & 1)))

    def invertBatch(self, nums, pz):
        return FpInvertBatch(self.F, nums, pz)


def _cmov(a, b, c):                          # modular.ts cmov: c ? b : a
    return b if c else a


def SWUFpSqrtRatio(F, Z):
    """hash-to-curve.ts:552-651: sqrt_ratio_3mod4 for q = 3 mod 4, else the generic F.2.1.1."""
    q = F.ORDER
    if q % 4 == 3:
        c1 = (q - 3) // 4
        c2 = F.sqrt(F.neg(Z))

        def sqrt_ratio(u, v):
            tv1 = F.sqr(v)
            tv2 = F.mul(u, v)
            tv1 = F.mul(tv1, tv2)
            y1 = F.mul(F.pow(tv1, c1), tv2)
            y2 = F.mul(y1, c2)
            tv3 = F.mul(F.sqr(y1), v)
            isQR = F.eql(tv3, u)
            return (not F.is0(v)) and isQR, _cmov(y2, y1, isQR)
        return sqrt_ratio
    l, o = 0, q - 1
    while o % 2 == 0:
        o //= 2
        l += 1
    c1 = l
    p2c1 = 1 << c1
    c2 = (q - 1) // p2c1
    c3 = (c2 - 1) // 2
    c4 = p2c1 - 1
    c5 = p2c1 >> 1
    c6 = F.pow(Z, c2)
    c7 = F.pow(Z, (c2 + 1) // 2)

    def sqrt_ratio(u, v):
        tv1 = c6
        tv2 = F.pow(v, c4)
        tv3 = F.mul(F.sqr(tv2), v)
        tv5 = F.mul(F.pow(F.mul(u, tv3), c3), tv2)
        tv2 = F.mul(tv5, v)
        tv3 = F.mul(tv5, u)
        tv4 = F.mul(tv3, tv2)
        tv5 = F.pow(tv4, c5)
        isQR = F.eql(tv5, F.ONE)
        tv2 = F.mul(tv3, c7)
        tv5 = F.mul(tv4, tv1)
        tv3 = _cmov(tv2, tv3, isQR)
        tv4 = _cmov(tv5, tv4, isQR)
        for i in range(c1, 1, -1):
            e = 1 << (i - 2)
            tvv5 = F.pow(tv4, e)
            e1 = F.eql(tvv5, F.ONE)
            tv2 = F.mul(tv3, tv1)
            tv1 = F.mul(tv1, tv1)
            tvv5 = F.mul(tv4, tv1)
            tv3 = _cmov(tv2, tv3, e1)
            tv4 = _cmov(tvv5, tv4, e1)
        return (not F.is0(v)) and (isQR or F.is0(u)), tv3
    return sqrt_ratio


def mapToCurveSimpleSWU(F, A, B, Z):
    """hash-to-curve.ts:652-717 (RFC 9380 F.2)."""
    sqrtRatio = SWUFpSqrtRatio(F, Z)

    def swu(u):
        tv1 = F.mul(F.sqr(u), Z)
        tv2 = F.add(F.sqr(tv1), tv1)
        tv3 = F.mul(F.add(tv2, F.ONE), B)
        tv4 = F.mul(_cmov(Z, F.neg(tv2), not F.eql(tv2, F.ZERO)), A)
        tv2 = F.sqr(tv3)
        tv6 = F.sqr(tv4)
        tv5 = F.mul(tv6, A)
        tv2 = F.mul(F.add(tv2, tv5), tv3)
        tv6 = F.mul(tv6, tv4)
        tv5 = F.mul(tv6, B)
        tv2 = F.add(tv2, tv5)
        x = F.mul(tv1, tv3)
        isValid, value = sqrtRatio(tv2, tv6)
        y = F.mul(F.mul(tv1, u), value)
        x = _cmov(x, tv3, isValid)
        y = _cmov(y, value, isValid)
        e1 = F.isOdd(u) == F.isOdd(y)
        y = _cmov(F.neg(y), y, e1)
        tv4_inv = F.invertBatch([tv4], True)[0]
        return F.mul(x, tv4_inv), y
    return swu


def isogenyMap(F, coeffs):
    """hash-to-curve.ts:381-410: Horner over reversed coefficient rows; zero denominator -> (0, 0)."""
    rows = [list(reversed(r)) for r in coeffs]

    def iso(x, y):
        vals = []
        for row in rows:
            acc = row[0]
            for c in row[1:]:
                acc = F.add(F.mul(acc, x), c)
            vals.append(acc)
        xn, xd, yn, yd = vals
        is_zero = F.is0(xd) or F.is0(yd)
        xd_inv, yd_inv = F.invertBatch([xd, yd], True)
        if is_zero:
            return F.ZERO, F.ZERO
        return F.mul(xn, xd_inv), F.mul(y, F.mul(yn, yd_inv))
    return iso


_f1, _f2 = _F1(BlsG1.Fp), _F2(BlsG2.Fp)
_g1c, _g2c = _K["G1"], _K["G2"]
_iso1 = isogenyMap(_f1, [[int(v) for v in _g1c[k]] for k in ("xnum", "xden", "ynum", "yden")])
_iso2 = isogenyMap(_f2, [[(int(a), int(b)) for a, b in _g2c[k]] for k in ("xnum", "xden", "ynum", "yden")])
_swu1 = mapToCurveSimpleSWU(_f1, int(_g1c["A"]), int(_g1c["B"]), int(_g1c["Z"]) % BLS_P)
_swu2 = mapToCurveSimpleSWU(_f2, tuple(int(v) % BLS_P for v in _g2c["A"]), tuple(int(v) % BLS_P for v in _g2c["B"]),
                            tuple(int(v) % BLS_P for v in _g2c["Z"]))


def mapToG1(scalars):                        # bls12-381.ts:853-856
    return _iso1(*_swu1(scalars[0] % BLS_P))


def mapToG2(scalars):                        # bls12-381.ts:859-862
    return _iso2(*_swu2((scalars[0] % BLS_P, scalars[1] % BLS_P)))


def g1_clear_cofactor(P):                    # bls12-381.ts:578-581
    return P.multiplyUnsafe(BLS_X).add(P)


def _psi2(Point, P):                         # tower.ts:249-256: (x * PSI2_X, -y)
    F2 = Point.Fp
    p = F2.Fp.ORDER
    base = F2.div(F2.ONE, (1, 1))
    PSI2_X = F2.pow(base, (p * p - 1) // 3)
    if P.is0():
        return P
    x, y = P.toAffine()
    return Point.fromAffine((F2.mul(x, PSI2_X), F2.neg(y)))


def g2_clear_cofactor(P):                    # bls12-381.ts:604-618
    Pt = type(P)
    t1 = P.multiplyUnsafe(BLS_X).negate()
    t2 = bls_g2_psi(Pt, P) if not P.is0() else P
    t3 = _psi2(Pt, P.double())
    t3 = t3.subtract(t2)
    t2 = t1.add(t2)
    t2 = t2.multiplyUnsafe(BLS_X).negate()
    t3 = t3.add(t2)
    t3 = t3.subtract(t1)
    return t3.subtract(P)


class Hasher:
    """createHasher (hash-to-curve.ts:441-548) for one of the two bls12-381 groups."""

    def __init__(self, Point, map_fn, clear_fn, m, DST):
        self.Point, self._map, self._clear, self.m, self.DST = Point, map_fn, clear_fn, m, DST

    def map(self, num):
        return self.Point.fromAffine(self._map(num))

    def clear(self, P):
        Q = self._clear(P)
        return self.Point.ZERO if Q.is0() else Q

    def hashToCurve(self, msg, DST=None):
        u = hash_to_field(msg, 2, BLS_P, self.m, 128, self.DST if DST is None else DST)
        return self.clear(self.map(u[0]).add(self.map(u[1])))

    def encodeToCurve(self, msg, DST=None):
        u = hash_to_field(msg, 1, BLS_P, self.m, 128, self.DST if DST is None else DST)
        return self.clear(self.map(u[0]))

    def mapToCurve(self, scalars):
        if self.m == 1:
            if not isinstance(scalars, int):
                raise ValueError("expected bigint (m=1)")
            return self.clear(self.map([scalars]))
        if not isinstance(scalars, (list, tuple)) or len(scalars) != self.m:
            raise ValueError("expected array of %d bigints" % self.m)
        return self.clear(self.map(list(scalars)))


G1_hasher = Hasher(BlsG1, mapToG1, g1_clear_cofactor, 1, DST_G1)
G2_hasher = Hasher(BlsG2, mapToG2, g2_clear_cofactor, 2, DST_G2)
