"""Host-side mirror of the reference's prime-field object (IField<bigint>, src/abstract/modular.ts:429-607,
class _Field :888-1040) for callers that build on the shim: same members, argument rules and error
messages, Python ints for bigints.  This is the reference's HOST logic (it is host BigInt code there too);
the device field arithmetic lives in csrc/fe9.hpp / fe29.hpp / fp.hpp and is reached through the point-level
entry points, never through this class.

Square roots follow the reference's dispatch (modular.ts:398-408) so the SAME root comes back:
p = 3 (mod 4) -> n^((p+1)/4); p = 5 (mod 8) -> Atkin; p = 9 (mod 16) -> Kong (RFC 9380 I.3);
otherwise Tonelli-Shanks with the first non-residue Z >= 2.
"""


def mod(a, b):
    """modular.ts `mod`: result in [0, b); b must be positive."""
    if b <= 0:
        raise ValueError("mod: expected positive modulus, got %d" % b)
    return a % b


def invert(number, modulo):
    """Extended Euclid (modular.ts:159-182), same failure messages."""
    if number == 0:
        raise ValueError("invert: expected non-zero number")
    if modulo <= 1:
        raise ValueError("invert: expected modulus > 1, got %d" % modulo)
    a, b, x, u = number % modulo, modulo, 0, 1
    while a != 0:
        q = b // a
        b, a, x, u = a, b - a * q, u, x - u * q
    if b != 1:
        raise ValueError("invert: does not exist")
    return x % modulo


def _is_int(n):
    return isinstance(n, int) and not isinstance(n, bool)


def FpPow(F, num, power):
    """modular.ts:666-706 (value only: the windowing is an implementation detail there)."""
    if not _is_int(power):
        raise TypeError("invalid exponent: expected bigint, got " + type(power).__name__)
    if power < 0:
        raise ValueError("invalid exponent, negatives unsupported")
    if power == 0:
        return F.ONE
    if power == 1:
        return num
    p, d = F.ONE, num
    while power > 0:
        if power & 1:
            p = F.mul(p, d)
        d = F.sqr(d)
        power >>= 1
    return p


def FpInvertBatch(F, nums, passZero=False):
    """Montgomery's trick (modular.ts:722-747): zeros are skipped; they come back as 0 with passZero,
    else as None (the reference leaves `undefined`)."""
    inverted = [F.ZERO if passZero else None] * len(nums)
    acc = F.ONE
    for i, n in enumerate(nums):
        if F.is0(n):
            continue
        inverted[i] = acc
        acc = F.mul(acc, n)
    inv = F.inv(acc)
    for i in range(len(nums) - 1, -1, -1):
        if F.is0(nums[i]):
            continue
        inverted[i] = F.mul(inv, inverted[i])
        inv = F.mul(inv, nums[i])
    return inverted


def _odd_modulus(order, name):
    if order & 1 == 0:
        raise ValueError("%s: expected odd modulus, got %d" % (name, order))


def FpLegendre(F, n):
    """1 / 0 / -1 (modular.ts:804-817)."""
    _odd_modulus(F.ORDER, "FpLegendre")
    powered = F.pow(n, (F.ORDER - 1) // 2)
    yes, zero, no = F.eql(powered, F.ONE), F.eql(powered, F.ZERO), F.eql(powered, F.neg(F.ONE))
    if not (yes or zero or no):
        raise ValueError("invalid Legendre symbol result")
    return 1 if yes else (0 if zero else -1)


def FpIsSquare(F, n):
    return FpLegendre(F, n) != -1


def _assert_square(F, root, n):
    if not F.eql(F.sqr(root), n):
        raise ValueError("Cannot find square root")


def _sqrt3mod4(F, n):
    root = F.pow(n, (F.ORDER + 1) // 4)
    _assert_square(F, root, n)
    return root


def _sqrt5mod8(F, n):
    v = F.pow(F.mul(n, 2), (F.ORDER - 5) // 8)
    nv = F.mul(n, v)
    i = F.mul(F.mul(nv, 2), v)
    root = F.mul(nv, F.sub(i, F.ONE))
    _assert_square(F, root, n)
    return root


def tonelliShanks(P):
    """modular.ts:306-371: returns sqrt(F, n) for the field of order P."""
    if P < 3:
        raise ValueError("sqrt is not defined for small field")
    _odd_modulus(P, "tonelliShanks")
    Q, S = P - 1, 0
    while Q % 2 == 0:
        Q //= 2
        S += 1
    base = Field(P)
    Z = 2
    while FpLegendre(base, Z) == 1:
        if Z > 1000:
            raise ValueError("Cannot find square root: probably non-prime P")
        Z += 1
    if S == 1:
        return _sqrt3mod4
    cc = base.pow(Z, Q)
    q1div2 = (Q + 1) // 2

    def slow(F, n):
        if F.is0(n):
            return n
        if FpLegendre(F, n) != 1:
            raise ValueError("Cannot find square root")
        M, c, t, R = S, F.mul(F.ONE, cc), F.pow(n, Q), F.pow(n, q1div2)
        while not F.eql(t, F.ONE):
            if F.is0(t):
                raise ValueError("Cannot find square root: probably non-prime P")
            i, t_tmp = 1, F.sqr(t)
            while not F.eql(t_tmp, F.ONE):
                i += 1
                t_tmp = F.sqr(t_tmp)
                if i == M:
                    raise ValueError("Cannot find square root")
            b = F.pow(c, 1 << (M - i - 1))
            M, c = i, F.sqr(b)
            t, R = F.mul(t, c), F.mul(R, b)
        return R

    return slow


def _sqrt9mod16(P):
    base = Field(P)
    tn = tonelliShanks(P)
    c1 = tn(base, base.neg(base.ONE))
    c2 = tn(base, c1)
    c3 = tn(base, base.neg(c1))
    c4 = (P + 7) // 16

    def kong(F, n):
        tv1 = F.pow(n, c4)
        tv2, tv3, tv4 = F.mul(tv1, c1), F.mul(tv1, c2), F.mul(tv1, c3)
        e1, e2 = F.eql(F.sqr(tv2), n), F.eql(F.sqr(tv3), n)
        tv1 = F.cmov(tv1, tv2, e1)
        tv2 = F.cmov(tv4, tv3, e2)
        root = F.cmov(tv1, tv2, F.eql(F.sqr(tv2), n))
        _assert_square(F, root, n)
        return root

    return kong


def FpSqrt(P):
    """Dispatcher (modular.ts:396-408)."""
    _odd_modulus(P, "Fp.sqrt")
    if P % 4 == 3:
        return _sqrt3mod4
    if P % 8 == 5:
        return _sqrt5mod8
    if P % 16 == 9:
        return _sqrt9mod16(P)
    return tonelliShanks(P)


class Field:
    """Field(ORDER, opts) of modular.ts:888-1040.  opts: BITS, sqrt (custom root function taking n),
    isLE, allowedLengths, modFromBytes."""

    ZERO = 0
    ONE = 1

    def __init__(self, ORDER, BITS=None, sqrt=None, isLE=False, allowedLengths=None, modFromBytes=False):
        if not _is_int(ORDER) or ORDER <= 1:
            raise ValueError("invalid field: expected ORDER > 1, got %s" % (ORDER,))
        self.ORDER = ORDER
        self.BITS = BITS if BITS is not None else ORDER.bit_length()
        self.BYTES = (self.BITS + 7) // 8
        if self.BYTES > 2048:
            raise ValueError("invalid field: expected ORDER of <= 2048 bytes")
        self.isLE = bool(isLE)
        self._lengths = tuple(allowedLengths) if allowedLengths else None
        self._mod = bool(modFromBytes)
        self._custom_sqrt = sqrt
        self._sqrt_fn = None

    # 1-argument
    def create(self, n):
        return n % self.ORDER

    def isValid(self, n):
        if not _is_int(n):
            raise TypeError("invalid field element: expected bigint, got " + type(n).__name__)
        return 0 <= n < self.ORDER

    def is0(self, n):
        return n == 0

    def isValidNot0(self, n):
        return not self.is0(n) and self.isValid(n)

    def isOdd(self, n):
        return n & 1 == 1

    def neg(self, n):
        return -n % self.ORDER

    def inv(self, n):
        return invert(n, self.ORDER)

    def sqrt(self, n):
        if self._custom_sqrt is not None:
            return self._custom_sqrt(n)
        if self._sqrt_fn is None:
            self._sqrt_fn = FpSqrt(self.ORDER)
        return self._sqrt_fn(self, n)

    def sqr(self, n):
        return n * n % self.ORDER

    # 2-argument
    def eql(self, a, b):
        return a == b

    def add(self, a, b):
        return (a + b) % self.ORDER

    def sub(self, a, b):
        return (a - b) % self.ORDER

    def mul(self, a, b):
        return a * b % self.ORDER

    def pow(self, n, power):
        if not _is_int(power):
            raise TypeError("invalid exponent: expected bigint, got " + type(power).__name__)
        if power < 0:
            raise ValueError("invalid exponent, negatives unsupported")
        return pow(n, power, self.ORDER)

    def div(self, a, b):
        return a * invert(b, self.ORDER) % self.ORDER

    # non-normalising forms
    def sqrN(self, n):
        return n * n

    def addN(self, a, b):
        return a + b

    def subN(self, a, b):
        return a - b

    def mulN(self, a, b):
        return a * b

    def toBytes(self, n):
        return int(n).to_bytes(self.BYTES, "little" if self.isLE else "big")

    def fromBytes(self, b, skipValidation=False):
        if not isinstance(b, (bytes, bytearray, memoryview)):
            raise TypeError("Uint8Array expected")
        b = bytes(b)
        if self._lengths:
            if len(b) < 1 or len(b) not in self._lengths or len(b) > self.BYTES:
                raise ValueError("Field.fromBytes: expected %s bytes, got %d"
                                 % (",".join(str(x) for x in self._lengths), len(b)))
            pad = bytes(self.BYTES - len(b))
            b = b + pad if self.isLE else pad + b
        if len(b) != self.BYTES:
            raise ValueError("Field.fromBytes: expected %d bytes, got %d" % (self.BYTES, len(b)))
        n = int.from_bytes(b, "little" if self.isLE else "big")
        if self._mod:
            n %= self.ORDER
        if not skipValidation and not self.isValid(n):
            raise ValueError("invalid field element: outside of range 0..ORDER")
        return n

    def invertBatch(self, lst):
        return FpInvertBatch(self, lst, True)

    def cmov(self, a, b, condition):
        if not isinstance(condition, bool):
            raise TypeError("cmov: expected boolean condition")
        return b if condition else a
