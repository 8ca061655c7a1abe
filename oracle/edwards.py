"""Twisted-Edwards group law (extended XYZT, hwcd-2008) and EdDSA verification,
restating reference `src/abstract/edwards.ts` and `src/ed25519.ts`.
Test infrastructure - see oracle/__init__.py.
"""
import hashlib

from . import curve as _curve
from .field import pow2


def edwards(CURVE, Fp, Fn, uvRatio, name="E"):
    """edwards.ts:297-660.  CURVE: dict(a, d, Gx, Gy, h)."""
    a, d = CURVE["a"], CURVE["d"]
    cofactor = CURVE["h"]
    MASK = 1 << (8 * Fp.BYTES)                          # edwards.ts:320
    P = Fp.ORDER
    if Fp.eql(a, Fp.neg(Fp.ONE)):                       # edwards.ts:347-350
        mulA = Fp.neg
    elif Fp.eql(a, Fp.ONE):
        mulA = lambda x: x                              # noqa: E731
    else:
        mulA = lambda x: Fp.mul(a, x)                   # noqa: E731

    def acoord(title, n, ban_zero=False):               # edwards.ts:356-360
        lo = 1 if ban_zero else 0
        if not (isinstance(n, int) and lo <= n < MASK):
            raise ValueError("expected valid coordinate %s" % title)
        return n

    class Point:
        __slots__ = ("X", "Y", "Z", "T")

        def __init__(self, X, Y, Z, T):                 # edwards.ts:379-385
            self.X = acoord("x", X)
            self.Y = acoord("y", Y)
            self.Z = acoord("z", Z, True)
            self.T = acoord("t", T)

        @classmethod
        def fromAffine(cls, p):                         # edwards.ts:396-402
            x, y = p
            acoord("x", x)
            acoord("y", y)
            return cls(x, y, Fp.ONE, Fp.mul(x, y))

        @classmethod
        def fromBytes(cls, data, zip215=False):         # edwards.ts:405-436
            ln = Fp.BYTES
            if len(data) != ln:
                raise ValueError("point: expected %d bytes" % ln)
            normed = bytearray(data)
            lastByte = data[ln - 1]
            normed[ln - 1] = lastByte & 0x7F
            y = int.from_bytes(normed, "little")
            mx = MASK if zip215 else P
            if not (0 <= y < mx):
                raise ValueError("point.y out of range")
            y2 = Fp.sqr(y)
            u = Fp.sub(y2, Fp.ONE)
            v = Fp.sub(d * y2, a)
            isValid, x = uvRatio(u, v)
            if not isValid:
                raise ValueError("bad point: invalid y coordinate")
            isXOdd = (x & 1) == 1
            isLastByteOdd = (lastByte & 0x80) != 0
            if not zip215 and Fp.is0(x) and isLastByteOdd:
                raise ValueError("bad point: x=0 and x_0=1")
            if isLastByteOdd != isXOdd:
                x = Fp.neg(x)
            return cls.fromAffine((x, y))

        def equals(self, o):                            # edwards.ts:482-491
            X1Z2 = Fp.mul(self.X, o.Z)
            X2Z1 = Fp.mul(o.X, self.Z)
            Y1Z2 = Fp.mul(self.Y, o.Z)
            Y2Z1 = Fp.mul(o.Y, self.Z)
            return X1Z2 == X2Z1 and Y1Z2 == Y2Z1

        def is0(self):                                  # edwards.ts:493-495
            return self.equals(Point.ZERO)

        def negate(self):                               # edwards.ts:497-500
            return Point(Fp.neg(self.X), self.Y, self.Z, Fp.neg(self.T))

        def double(self):                               # edwards.ts:505-521 (dbl-2008-hwcd)
            X1, Y1, Z1 = self.X, self.Y, self.Z
            A = Fp.sqr(X1)
            B = Fp.sqr(Y1)
            C = Fp.mul(Fp.sqr(Z1), 2)
            D = mulA(A)
            x1y1 = X1 + Y1
            E = Fp.sub(Fp.sqr(x1y1) - A, B)
            G = D + B
            F = G - C
            H = D - B
            return Point(Fp.mul(E, F), Fp.mul(G, H), Fp.mul(F, G), Fp.mul(E, H))

        def add(self, o):                               # edwards.ts:526-545 (add-2008-hwcd)
            if not isinstance(o, Point):
                raise TypeError("EdwardsPoint expected")
            X1, Y1, Z1, T1 = self.X, self.Y, self.Z, self.T
            X2, Y2, Z2, T2 = o.X, o.Y, o.Z, o.T
            A = Fp.mul(X1, X2)
            B = Fp.mul(Y1, Y2)
            C = Fp.mul(T1 * d, T2)
            D = Fp.mul(Z1, Z2)
            E = Fp.sub((X1 + Y1) * (X2 + Y2) - A, B)
            F = D - C
            G = D + C
            H = Fp.sub(B, mulA(A))
            return Point(Fp.mul(E, F), Fp.mul(G, H), Fp.mul(F, G), Fp.mul(E, H))

        def subtract(self, o):                          # edwards.ts:547-552
            return self.add(o.negate())

        def multiply(self, sc):                         # edwards.ts:555-564 (value only)
            if not (isinstance(sc, int) and Fn.isValidNot0(sc)):
                raise ValueError("invalid scalar: expected 1 <= sc < curve.n")
            p = _curve.naiveMul(Point, self, sc)
            return _curve.normalizeZ(Point, [p])[0]

        def multiplyUnsafe(self, sc):                   # edwards.ts:571-577
            if not (isinstance(sc, int) and Fn.isValid(sc)):
                raise ValueError("invalid scalar: expected 0 <= sc < curve.n")
            if sc == 0:
                return Point.ZERO
            if self.is0() or sc == 1:
                return self
            # wnaf.mulUnsafe -> mulAddUnsafe(Point, [p], [sc], true)  curve.ts:752-764
            return _curve.mulAddUnsafe(Point, [self], [sc], True)

        def isSmallOrder(self):                         # edwards.ts:583-585
            return self.clearCofactor().is0()

        def isTorsionFree(self):                        # edwards.ts:589-591
            return _curve.mulAddUnsafe(Point, [self], [Fn.ORDER], True).is0()

        def toAffine(self, iz=None):                    # edwards.ts:595-609
            X, Y, Z = self.X, self.Y, self.Z
            is0 = self.is0()
            if iz is None:
                iz = Fp.create(8) if is0 else Fp.inv(Z)
            x = Fp.mul(X, iz)
            y = Fp.mul(Y, iz)
            zz = Fp.mul(Z, iz)
            if is0:
                return (Fp.ZERO, Fp.ONE)
            if zz != Fp.ONE:
                raise ValueError("invZ was invalid")
            return (x, y)

        def clearCofactor(self):                        # edwards.ts:611-618
            if cofactor == 1:
                return self
            if cofactor == 2:
                return self.double()
            if cofactor == 4:
                return self.double().double()
            if cofactor == 8:
                return self.double().double().double()
            return self.multiplyUnsafe(cofactor)

        def toBytes(self):                              # edwards.ts:620-628
            x, y = self.toAffine()
            b = bytearray(y.to_bytes(Fp.BYTES, "little"))
            b[-1] |= 0x80 if (x & 1) else 0
            return bytes(b)

    Point.Fp = Fp
    Point.Fn = Fn
    Point.CURVE = CURVE
    Point.ZERO = Point(Fp.ZERO, Fp.ONE, Fp.ONE, Fp.ZERO)   # edwards.ts:370
    Point.BASE = Point(CURVE["Gx"], CURVE["Gy"], Fp.ONE, Fp.mul(CURVE["Gx"], CURVE["Gy"]))
    return Point


# ---------------------------------------------------------------- ed25519 specifics
ED25519_P = (1 << 255) - 19                                # ed25519.ts:49-51
ED25519_SQRT_M1 = 19681161376707505956807079304988542015446066515923890162744021073123829784752


def ed25519_pow_2_252_3(x):
    """ed25519.ts:67-86 - returns (x^((p-5)/8), x^3)."""
    P = ED25519_P
    x2 = x * x % P
    b2 = x2 * x % P
    b4 = pow2(b2, 2, P) * b2 % P
    b5 = pow2(b4, 1, P) * x % P
    b10 = pow2(b5, 5, P) * b5 % P
    b20 = pow2(b10, 10, P) * b10 % P
    b40 = pow2(b20, 20, P) * b20 % P
    b80 = pow2(b40, 40, P) * b40 % P
    b160 = pow2(b80, 80, P) * b80 % P
    b240 = pow2(b160, 80, P) * b80 % P
    b250 = pow2(b240, 10, P) * b10 % P
    pow_p_5_8 = pow2(b250, 2, P) * x % P
    return pow_p_5_8, b2


def ed25519_uvRatio(u, v):
    """ed25519.ts:107-125 - sqrt(u/v) with the three-candidate check."""
    P = ED25519_P
    v3 = v * v * v % P
    v7 = v3 * v3 * v % P
    pw = ed25519_pow_2_252_3(u * v7)[0]
    x = u * v3 * pw % P
    vx2 = v * x * x % P
    root1 = x
    root2 = x * ED25519_SQRT_M1 % P
    useRoot1 = vx2 == u
    useRoot2 = vx2 == (-u) % P
    noRoot = vx2 == (-u * ED25519_SQRT_M1) % P
    if useRoot1:
        x = root1
    if useRoot2 or noRoot:
        x = root2
    if (x % P) & 1:                                      # isNegativeLE modular.ts:422
        x = (-x) % P
    return (useRoot1 or useRoot2), x


def eddsa_hash_k(Fn, r_bytes, pk_bytes, msg):
    """edwards.ts:900-906 hashDomainToScalar + :866-868 modN_LE, SHA-512 (ed25519.ts:166)."""
    h = hashlib.sha512(bytes(r_bytes) + bytes(pk_bytes) + bytes(msg)).digest()
    return int.from_bytes(h, "little") % Fn.ORDER


def eddsa_verify(Point, sig, msg, publicKey, zip215=True, k=None):
    """edwards.ts:942-989.  `k` may be supplied (already-hashed challenge) to
    restate only the curve part."""
    Fn = Point.Fn
    ln = 2 * Point.Fp.BYTES
    if len(sig) != ln:
        raise ValueError("signature: expected %d bytes" % ln)
    if len(publicKey) != Point.Fp.BYTES:
        raise ValueError("publicKey: expected %d bytes" % Point.Fp.BYTES)
    mid = ln // 2
    r = sig[:mid]
    s = int.from_bytes(sig[mid:], "little")
    try:
        A = Point.fromBytes(publicKey, zip215)
        R = Point.fromBytes(r, zip215)
        SB = Point.BASE.multiplyUnsafe(s)
    except ValueError:
        return False
    if not zip215 and A.isSmallOrder():
        return False
    if k is None:
        k = eddsa_hash_k(Fn, r, publicKey, msg)
    RkA = R.add(A.multiplyUnsafe(k))
    return RkA.subtract(SB).clearCofactor().is0()
