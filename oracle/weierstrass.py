"""Short-Weierstrass group law (projective XYZ, Renes-Costello-Batina complete
formulas) restating reference `src/abstract/weierstrass.ts`.
Test infrastructure - see oracle/__init__.py.
"""
from . import curve as _curve


def _tdiv(a, b):
    """BigInt `/` truncates toward zero; Python `//` floors."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def divNearest(num, den):
    """weierstrass.ts:106: (num + (num >= 0 ? den : -den) / 2n) / den with truncating `/`."""
    half = _tdiv(den if num >= 0 else -den, 2)
    return _tdiv(num + half, den)


def _splitEndoScalar(k, basis, n):
    """weierstrass.ts:121-148 - GLV split k -> (k1neg, k1, k2neg, k2), |ki| < 2^ceil(bits/2)."""
    if not (0 <= k < n):
        raise ValueError("expected valid scalar: 0 <= n < %d" % n)
    (a1, b1), (a2, b2) = basis
    c1 = divNearest(b2 * k, n)
    c2 = divNearest(-b1 * k, n)
    k1 = k - c1 * a1 - c2 * a2
    k2 = -c1 * b1 - c2 * b2
    k1neg, k2neg = k1 < 0, k2 < 0
    if k1neg:
        k1 = -k1
    if k2neg:
        k2 = -k2
    MAX_NUM = _curve.bitMask(-(-_curve.bitLen(n) // 2)) + 1
    if k1 < 0 or k1 >= MAX_NUM or k2 < 0 or k2 >= MAX_NUM:
        raise ValueError("splitScalar (endomorphism): failed for k")
    return k1neg, k1, k2neg, k2


def weierstrass(CURVE, Fp, Fn, endo=None, name="W"):
    """Builds a Point class for y^2 = x^3 + a x + b (weierstrass.ts:501-1020).
    CURVE: dict with a, b, Gx, Gy, h.  `endo`: dict(beta=..., basises=...)."""
    a, b = CURVE["a"], CURVE["b"]
    b3 = Fp.mul(b, 3)                                   # weierstrass.ts:612
    a_is0 = Fp.is0(a)

    def mulA(x):                                        # weierstrass.ts:613
        return Fp.ZERO if a_is0 else Fp.mul(a, x)

    class Point:
        __slots__ = ("X", "Y", "Z")

        def __init__(self, X, Y, Z):                    # weierstrass.ts:696-704
            for t, v, ban0 in (("x", X, False), ("y", Y, True), ("z", Z, False)):
                if not Fp.isValid(v) or (ban0 and Fp.is0(v)):
                    raise ValueError("bad point coordinate %s" % t)
            self.X, self.Y, self.Z = X, Y, Z

        @classmethod
        def fromAffine(cls, p):                         # weierstrass.ts:710-718
            x, y = p
            if not Fp.isValid(x) or not Fp.isValid(y):
                raise ValueError("invalid affine point")
            if Fp.is0(x) and Fp.is0(y):
                return cls.ZERO
            return cls(x, y, Fp.ONE)

        def equals(self, o):                            # weierstrass.ts:775-783
            U1 = Fp.eql(Fp.mul(self.X, o.Z), Fp.mul(o.X, self.Z))
            U2 = Fp.eql(Fp.mul(self.Y, o.Z), Fp.mul(o.Y, self.Z))
            return U1 and U2

        def negate(self):                               # weierstrass.ts:785-787
            return Point(self.X, Fp.neg(self.Y), self.Z)

        def double(self):                               # weierstrass.ts:793-828 (RCB alg. 3)
            X1, Y1, Z1 = self.X, self.Y, self.Z
            mul, add, sub = Fp.mul, Fp.add, Fp.sub
            t0 = mul(X1, X1)
            t1 = mul(Y1, Y1)
            t2 = mul(Z1, Z1)
            t3 = mul(X1, Y1)
            t3 = add(t3, t3)
            Z3 = mul(X1, Z1)
            Z3 = add(Z3, Z3)
            X3 = mulA(Z3)
            Y3 = mul(b3, t2)
            Y3 = add(X3, Y3)
            X3 = sub(t1, Y3)
            Y3 = add(t1, Y3)
            Y3 = mul(X3, Y3)
            X3 = mul(t3, X3)
            Z3 = mul(b3, Z3)
            t2 = mulA(t2)
            t3 = sub(t0, t2)
            t3 = mulA(t3)
            t3 = add(t3, Z3)
            Z3 = add(t0, t0)
            t0 = add(Z3, t0)
            t0 = add(t0, t2)
            t0 = mul(t0, t3)
            Y3 = add(Y3, t0)
            t2 = mul(Y1, Z1)
            t2 = add(t2, t2)
            t0 = mul(t2, t3)
            X3 = sub(X3, t0)
            Z3 = mul(t2, t1)
            Z3 = add(Z3, Z3)
            Z3 = add(Z3, Z3)
            return Point(X3, Y3, Z3)

        def add(self, o):                               # weierstrass.ts:834-880 (RCB alg. 1)
            if not isinstance(o, Point):
                raise TypeError("Weierstrass Point expected")
            X1, Y1, Z1 = self.X, self.Y, self.Z
            X2, Y2, Z2 = o.X, o.Y, o.Z
            mul, add, sub = Fp.mul, Fp.add, Fp.sub
            t0 = mul(X1, X2)
            t1 = mul(Y1, Y2)
            t2 = mul(Z1, Z2)
            t3 = add(X1, Y1)
            t4 = add(X2, Y2)
            t3 = mul(t3, t4)
            t4 = add(t0, t1)
            t3 = sub(t3, t4)
            t4 = add(X1, Z1)
            t5 = add(X2, Z2)
            t4 = mul(t4, t5)
            t5 = add(t0, t2)
            t4 = sub(t4, t5)
            t5 = add(Y1, Z1)
            X3 = add(Y2, Z2)
            t5 = mul(t5, X3)
            X3 = add(t1, t2)
            t5 = sub(t5, X3)
            Z3 = mulA(t4)
            X3 = mul(b3, t2)
            Z3 = add(X3, Z3)
            X3 = sub(t1, Z3)
            Z3 = add(t1, Z3)
            Y3 = mul(X3, Z3)
            t1 = add(t0, t0)
            t1 = add(t1, t0)
            t2 = mulA(t2)
            t4 = mul(b3, t4)
            t1 = add(t1, t2)
            t2 = sub(t0, t2)
            t2 = mulA(t2)
            t4 = add(t4, t2)
            t0 = mul(t1, t4)
            Y3 = add(Y3, t0)
            t0 = mul(t5, t4)
            X3 = mul(t3, X3)
            X3 = sub(X3, t0)
            t0 = mul(t3, t1)
            Z3 = mul(t5, Z3)
            Z3 = add(Z3, t0)
            return Point(X3, Y3, Z3)

        def subtract(self, o):                          # weierstrass.ts:882-887
            return self.add(o.negate())

        def is0(self):                                  # weierstrass.ts:889-891
            return self.equals(Point.ZERO)

        def toAffine(self, iz=None):                    # weierstrass.ts:951-969
            X, Y, Z = self.X, self.Y, self.Z
            if Fp.eql(Z, Fp.ONE):
                return (X, Y)
            is0 = self.is0()
            if iz is None:
                iz = Fp.ONE if is0 else Fp.inv(Z)
            x = Fp.mul(X, iz)
            y = Fp.mul(Y, iz)
            zz = Fp.mul(Z, iz)
            if is0:
                return (Fp.ZERO, Fp.ZERO)
            if not Fp.eql(zz, Fp.ONE):
                raise ValueError("invZ was invalid")
            return (x, y)

        def multiplyUnsafe(self, sc):                   # weierstrass.ts:915-928
            if not (isinstance(sc, int) and Fn.isValid(sc)):
                raise ValueError("invalid scalar: out of range")
            if sc == 0 or self.is0():
                return Point.ZERO
            if sc == 1:
                return self
            points, scalars = [], []
            _pushWnafPair(points, scalars, self, sc)
            return _curve.mulAddUnsafe(Point, points, scalars)

        def mulAddUnsafe(self, a_, other, b_):          # weierstrass.ts:937-944
            points, scalars = [], []
            _pushWnafPair(points, scalars, self, a_)
            _pushWnafPair(points, scalars, other, b_)
            return _curve.mulAddUnsafe(Point, points, scalars)

        def multiply(self, sc):
            """weierstrass.ts:900-907.  The reference runs a blinded constant-time ladder
            (curve.ts:647-729) and normalises; the *value* is sc*P, Z=1-normalised and
            independent of the blind (test/point.test.ts:647-651).  The oracle computes
            the same value with the fixed-window ladder curve.ts:707-729 unblinded."""
            if not (isinstance(sc, int) and Fn.isValidNot0(sc)):
                raise ValueError("invalid scalar: out of range")
            p = fixedWindow(self, sc, Fn.BITS)
            return _curve.normalizeZ(Point, [p])[0]

        def __repr__(self):
            return "%s.Point(%x, %x, %x)" % (name, self.X, self.Y, self.Z) \
                if isinstance(self.X, int) else "%s.Point(%r)" % (name, (self.X, self.Y, self.Z))

    def fixedWindow(p, n, bits, W=5):
        """curve.ts:707-729 fixedWindowCT value: flat 2^W table, W doublings + 1 add per window."""
        size = 1 << W
        table = [Point.ZERO, p]
        for i in range(2, size):
            table.append(table[i - 1].add(p))
        windows = -(-bits // W)
        acc = Point.ZERO
        for w in range(windows - 1, -1, -1):
            if w != windows - 1:
                for _ in range(W):
                    acc = acc.double()
            acc = acc.add(table[(n >> (w * W)) & (size - 1)])
        return acc

    def _pushWnafPair(points, scalars, p, k):           # weierstrass.ts:660-671
        if not (isinstance(k, int) and Fn.isValid(k)):
            raise ValueError("invalid scalar: out of range")
        if endo:
            k1neg, k1, k2neg, k2 = _splitEndoScalar(k, endo["basises"], Fn.ORDER)
            psi = Point(Fp.mul(p.X, endo["beta"]), p.Y, p.Z)
            points.append(p.negate() if k1neg else p)
            points.append(psi.negate() if k2neg else psi)
            scalars.extend([k1, k2])
        else:
            points.append(p)
            scalars.append(k)

    Point.Fp = Fp
    Point.Fn = Fn
    Point.CURVE = CURVE
    Point.endo = endo
    Point.ZERO = Point.__new__(Point)                   # weierstrass.ts:687 (0, 1, 0)
    Point.ZERO.X, Point.ZERO.Y, Point.ZERO.Z = Fp.ZERO, Fp.ONE, Fp.ZERO
    Point.BASE = Point(CURVE["Gx"], CURVE["Gy"], Fp.ONE)
    Point.isValidXY = staticmethod(
        lambda x, y: Fp.eql(Fp.sqr(y), Fp.add(Fp.add(Fp.mul(Fp.sqr(x), x), Fp.mul(x, a)), b)))
    return Point


# ------------------------------------------------------------------ codecs (test-side helpers)
def sec1_decode(Point, data):
    """SEC1 point decoding, weierstrass.ts:566-605 (compressed 02/03 and uncompressed 04)."""
    Fp = Point.Fp
    L = Fp.BYTES
    data = bytes(data)
    head = data[0] if data else None
    if len(data) == L + 1 and head in (2, 3):
        x = int.from_bytes(data[1:], "big")
        if not Fp.isValid(x):
            raise ValueError("bad point: is not on curve, wrong x")
        C = Point.CURVE
        y2 = Fp.add(Fp.add(Fp.mul(Fp.sqr(x), x), Fp.mul(x, C["a"])), C["b"])
        try:
            y = Fp.sqrt(y2)
        except ValueError:
            raise ValueError("bad point: is not on curve, sqrt error")
        if (y & 1) != (head & 1):
            y = Fp.neg(y)
        p = Point.fromAffine((x, y))
    elif len(data) == 2 * L + 1 and head == 4:
        x = int.from_bytes(data[1:1 + L], "big")
        y = int.from_bytes(data[1 + L:], "big")
        if not (Fp.isValid(x) and Fp.isValid(y)) or not Point.isValidXY(x, y):
            raise ValueError("bad point: is not on curve")
        p = Point.fromAffine((x, y))
    else:
        raise ValueError("bad point: got length %d" % len(data))
    if p.is0():
        raise ValueError("bad point: ZERO")
    return p


def sec1_encode(p, compressed=True):
    """weierstrass.ts:541-564."""
    Fp = type(p).Fp
    if p.is0():
        raise ValueError("bad point: ZERO")
    x, y = p.toAffine()
    if compressed:
        return bytes([3 if (y & 1) else 2]) + x.to_bytes(Fp.BYTES, "big")
    return b"\x04" + x.to_bytes(Fp.BYTES, "big") + y.to_bytes(Fp.BYTES, "big")


def bls_g1_decode_uncompressed(Point, data):
    """bls12-381.ts:377-433 G1 codec, 96-byte form: flag bits (compressed 0x80,
    infinity 0x40, sort 0x20) live in the top three bits of byte 0 (:446-468)."""
    data = bytearray(data)
    flags = data[0] & 0xE0
    data[0] &= 0x1F
    if flags & 0x40:
        return Point.ZERO
    x = int.from_bytes(data[:48], "big")
    y = int.from_bytes(data[48:], "big")
    return Point.fromAffine((x, y))


def bls_g2_decode_uncompressed(Point, data):
    """bls12-381.ts:354-368,435-504 G2 codec, 192-byte form, wire order c1 then c0."""
    data = bytearray(data)
    flags = data[0] & 0xE0
    data[0] &= 0x1F
    if flags & 0x40:
        return Point.ZERO
    x1 = int.from_bytes(data[0:48], "big")
    x0 = int.from_bytes(data[48:96], "big")
    y1 = int.from_bytes(data[96:144], "big")
    y0 = int.from_bytes(data[144:192], "big")
    return Point.fromAffine(((x0, x1), (y0, y1)))


BLS_X = 0xD201000000010000                               # bls12-381.ts:117
G1_BETA = 0x5F19672FDF76CE51BA69C6076A0F77EADDB3A93BE6F89688DE17D813620A00022E01FFFFFFFEFFFE


def bls_g1_is_torsion_free(Point, p):
    """bls12-381.ts:567-577: [x^2]P == phi(P) with phi(X, Y, Z) = (beta X, Y, Z)."""
    Fp = Point.Fp
    phi = Point(Fp.mul(p.X, G1_BETA), p.Y, p.Z)
    xP = p.multiplyUnsafe(BLS_X).negate()
    u2P = xP.multiplyUnsafe(BLS_X)
    return u2P.equals(phi)


def bls_g1_decode_compressed(Point, data):
    """G1 point decoding from the 48-byte compressed form (bls12-381.ts:377-433 coder.decode with
    parseMask/validateMask :436-459), followed by what Point.fromBytes adds: fromAffine +
    assertValidity, i.e. on-curve and prime-order-subgroup checks (weierstrass.ts:720-766)."""
    Fp = Point.Fp
    data = bytes(data)
    if len(data) != 48:
        raise ValueError("invalid G1 point: expected 48 bytes")
    mask = data[0] & 0xE0
    compressed, infinity, sort = bool(mask & 0x80), bool(mask & 0x40), bool(mask & 0x20)
    if (not compressed and sort) or (compressed and infinity and sort):
        raise ValueError("invalid encoding flag")
    if not compressed:
        raise ValueError("invalid G1 point: expected 96 bytes")
    value = bytes([data[0] & 0x1F]) + data[1:]
    if infinity:
        if any(value):
            raise ValueError("invalid G1 point: non-canonical zero")
        return Point.ZERO
    x = int.from_bytes(value, "big")
    if not Fp.isValid(x):
        raise ValueError("invalid field element: outside of range 0..ORDER")
    y = Fp.sqrt(Fp.add(Fp.pow(x, 3), Point.CURVE["b"]))     # raises if there is no root
    if bool((y * 2) // Fp.ORDER) != sort:                    # sortBit :346-351 (y != 0 on this curve)
        y = Fp.neg(y)
    P = Point.fromAffine((x, y))
    if P.is0():
        raise ValueError("bad point: ZERO")
    if not bls_g1_is_torsion_free(Point, P):
        raise ValueError("bad point: not in prime-order subgroup")
    return P


def bls_g1_encode_compressed(p):
    """bls12-381.ts:400-410 coder.encode, compressed form."""
    Fp = type(p).Fp
    if p.is0():
        return bytes([0xC0]) + bytes(47)
    x, y = p.toAffine()
    b = bytearray(x.to_bytes(48, "big"))
    b[0] |= 0x80
    if (y * 2) // Fp.ORDER:
        b[0] |= 0x20
    return bytes(b)


def bls_g2_psi(Point, P):
    """tower.ts:240-247,257-270 psiFrobenius.G2psi with base 1/(u+1) (bls12-381.ts:283):
    psi(x, y) = (conj(x) * PSI_X, conj(y) * PSI_Y) on affine coordinates."""
    F2 = Point.Fp
    p = F2.Fp.ORDER
    base = F2.div(F2.ONE, (1, 1))
    PSI_X = F2.pow(base, (p - 1) // 3)
    PSI_Y = F2.pow(base, (p - 1) // 2)
    x, y = P.toAffine()
    return Point.fromAffine((F2.mul(F2.frobeniusMap(x, 1), PSI_X), F2.mul(F2.frobeniusMap(y, 1), PSI_Y)))


def bls_g2_is_torsion_free(Point, P):
    """bls12-381.ts:599-601: [-x]P == psi(P)."""
    return P.multiplyUnsafe(BLS_X).negate().equals(bls_g2_psi(Point, P))


def _bls_sort_bit(parts, p):
    """bls12-381.ts:346-351."""
    for part in parts:
        if part != 0:
            return bool((part * 2) // p)
    return False


def bls_g2_decode_compressed(Point, data):
    """G2 point decoding from the 96-byte compressed form: coder.decode (bls12-381.ts:377-433)
    with fp2.decode (wire order c1 || c0, :354-368) and yparts [c1, c0] (:476-479), then
    fromAffine + assertValidity (weierstrass.ts:720-766) incl. the psi subgroup check."""
    F2 = Point.Fp
    Fp = F2.Fp
    data = bytes(data)
    if len(data) != 96:
        raise ValueError("invalid G2 point: expected 96 bytes")
    mask = data[0] & 0xE0
    compressed, infinity, sort = bool(mask & 0x80), bool(mask & 0x40), bool(mask & 0x20)
    if (not compressed and sort) or (compressed and infinity and sort):
        raise ValueError("invalid encoding flag")
    if not compressed:
        raise ValueError("invalid G2 point: expected 192 bytes")
    value = bytes([data[0] & 0x1F]) + data[1:]
    if infinity:
        if any(value):
            raise ValueError("invalid G2 point: non-canonical zero")
        return Point.ZERO
    x = (Fp.fromBytes(value[48:]), Fp.fromBytes(value[:48]))
    y = F2.sqrt(F2.add(F2.mul(F2.sqr(x), x), Point.CURVE["b"]))   # raises if there is no root
    if _bls_sort_bit([y[1], y[0]], Fp.ORDER) != sort:
        y = F2.neg(y)
    P = Point.fromAffine((x, y))
    if P.is0():
        raise ValueError("bad point: ZERO")
    if not bls_g2_is_torsion_free(Point, P):
        raise ValueError("bad point: not in prime-order subgroup")
    return P


def bls_g2_encode_compressed(p):
    """bls12-381.ts:400-410 coder.encode with fp2.encode, compressed form."""
    Fp = type(p).Fp.Fp
    if p.is0():
        return bytes([0xC0]) + bytes(95)
    x, y = p.toAffine()
    b = bytearray(x[1].to_bytes(48, "big") + x[0].to_bytes(48, "big"))
    b[0] |= 0x80
    if _bls_sort_bit([y[1], y[0]], Fp.ORDER):
        b[0] |= 0x20
    return bytes(b)
