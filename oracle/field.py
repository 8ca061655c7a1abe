"""Prime-field and Fp2 arithmetic, restating reference `src/abstract/modular.ts`
and `src/abstract/tower.ts` (Fp2 only).  Test infrastructure - see oracle/__init__.py.
"""


def mod(a, b):
    """modular.ts:50-54 - a % b lifted into [0, b).  (Python's % already is.)"""
    return a % b


def invert(number, modulo):
    """modular.ts:159-182 - extended Euclid, throws on 0 / non-coprime."""
    if number == 0:
        raise ValueError("invert: expected non-zero number")
    if modulo <= 1:
        raise ValueError("invert: expected modulus > 1, got %d" % modulo)
    a, b = number % modulo, modulo
    x, u = 0, 1
    while a != 0:
        q = b // a
        r = b - a * q
        m = x - u * q
        b, a, x, u = a, r, u, m
    if b != 1:
        raise ValueError("invert: does not exist")
    return x % modulo


def pow2(x, power, modulo):
    """modular.ts:134-143 - x^(2^power) mod modulo by repeated squaring."""
    res = x
    for _ in range(power):
        res = res * res % modulo
    return res


class Field:
    """`_Field` class, modular.ts:888-1038: canonical residues in [0, ORDER)."""

    def __init__(self, order, is_le=False, bits=None):
        self.ORDER = order
        self.BITS = order.bit_length() if bits is None else bits
        self.BYTES = (self.BITS + 7) // 8
        self.isLE = is_le
        self.ZERO = 0
        self.ONE = 1

    def create(self, n):            # modular.ts:922
        return n % self.ORDER

    def isValid(self, n):           # modular.ts:925-931
        if not isinstance(n, int) or isinstance(n, bool):
            raise TypeError("invalid field element: expected bigint")
        return 0 <= n < self.ORDER

    def isValidNot0(self, n):
        return self.isValid(n) and n != 0

    def is0(self, n):
        return n == 0

    def isOdd(self, n):
        return (n & 1) == 1

    def neg(self, n):               # modular.ts:940
        return (-n) % self.ORDER

    def eql(self, a, b):
        return a == b

    def sqr(self, n):               # modular.ts:947
        return n * n % self.ORDER

    def add(self, a, b):            # modular.ts:950
        return (a + b) % self.ORDER

    def sub(self, a, b):            # modular.ts:953
        return (a - b) % self.ORDER

    def mul(self, a, b):            # modular.ts:956
        return a * b % self.ORDER

    def pow(self, n, p):            # modular.ts:959 -> FpPow :666-706
        if p < 0:
            raise ValueError("invalid exponent, negatives unsupported")
        return pow(n, p, self.ORDER)

    def div(self, a, b):            # modular.ts:962
        return a * invert(b, self.ORDER) % self.ORDER

    def inv(self, n):               # modular.ts:982
        return invert(n, self.ORDER)

    def invertBatch(self, nums, pass_zero=False):
        return FpInvertBatch(self, nums, pass_zero)

    def sqrt(self, n):
        """modular.ts:238-258 (p = 3 mod 4) / :398-408 dispatcher; enough for the
        fields on the path (secp256k1 p, bls12-381 p are both 3 mod 4)."""
        p = self.ORDER
        if p % 4 != 3:
            raise NotImplementedError("sqrt only for p = 3 mod 4 here")
        root = pow(n, (p + 1) // 4, p)
        if root * root % p != n % p:
            raise ValueError("Cannot find square root")
        return root

    def toBytes(self, n):
        return n.to_bytes(self.BYTES, "little" if self.isLE else "big")

    def fromBytes(self, b):
        if len(b) != self.BYTES:
            raise ValueError("Field.fromBytes: expected %d bytes" % self.BYTES)
        n = int.from_bytes(b, "little" if self.isLE else "big")
        if not self.isValid(n):
            raise ValueError("invalid field element: outside of range 0..ORDER")
        return n


def FpInvertBatch(F, nums, pass_zero=False):
    """modular.ts:728-760 - Montgomery trick; zeros -> None (or ZERO with pass_zero)."""
    inverted = [F.ZERO if pass_zero else None] * len(nums)
    acc = F.ONE
    for i, num in enumerate(nums):
        if F.is0(num):
            continue
        inverted[i] = acc
        acc = F.mul(acc, num)
    inv_acc = F.inv(acc)
    for i in range(len(nums) - 1, -1, -1):
        num = nums[i]
        if F.is0(num):
            continue
        inverted[i] = F.mul(inv_acc, inverted[i])
        inv_acc = F.mul(inv_acc, num)
    return inverted


class Field2:
    """`_Field2`, tower.ts:305-561: Fp2 = Fp[u]/(u^2+1).  Elements are (c0, c1) tuples."""

    def __init__(self, Fp):
        self.Fp = Fp
        self.ORDER = Fp.ORDER * Fp.ORDER
        self.BITS = self.ORDER.bit_length()
        self.BYTES = 2 * Fp.BYTES
        self.ZERO = (0, 0)
        self.ONE = (1, 0)

    def create(self, n):
        return (n[0] % self.Fp.ORDER, n[1] % self.Fp.ORDER)

    def isValid(self, n):
        return (isinstance(n, tuple) and len(n) == 2
                and self.Fp.isValid(n[0]) and self.Fp.isValid(n[1]))

    def is0(self, n):
        return n[0] == 0 and n[1] == 0

    def eql(self, a, b):            # tower.ts:388
        return a[0] == b[0] and a[1] == b[1]

    def neg(self, n):               # tower.ts:393
        Fp = self.Fp
        return (Fp.neg(n[0]), Fp.neg(n[1]))

    def add(self, a, b):            # tower.ts:404
        Fp = self.Fp
        return (Fp.add(a[0], b[0]), Fp.add(a[1], b[1]))

    def sub(self, a, b):            # tower.ts:413
        Fp = self.Fp
        return (Fp.sub(a[0], b[0]), Fp.sub(a[1], b[1]))

    def mul(self, a, rhs):          # tower.ts:420-431 (Karatsuba, 3 Fp.mul)
        Fp = self.Fp
        if isinstance(rhs, int):
            return (Fp.mul(a[0], rhs), Fp.mul(a[1], rhs))
        c0, c1 = a
        r0, r1 = rhs
        t1 = Fp.mul(c0, r0)
        t2 = Fp.mul(c1, r1)
        o0 = Fp.sub(t1, t2)
        o1 = Fp.sub(Fp.mul(Fp.add(c0, c1), Fp.add(r0, r1)), Fp.add(t1, t2))
        return (o0, o1)

    def sqr(self, n):               # tower.ts:432-438
        Fp = self.Fp
        c0, c1 = n
        a = Fp.add(c0, c1)
        b = Fp.sub(c0, c1)
        c = Fp.add(c0, c0)
        return (Fp.mul(a, b), Fp.mul(c, c1))

    def inv(self, n):               # tower.ts:458-475
        Fp = self.Fp
        a, b = n
        factor = Fp.inv(Fp.create(a * a + b * b))
        return (Fp.mul(factor, Fp.create(a)), Fp.mul(factor, Fp.create(-b)))

    def invertBatch(self, nums, pass_zero=True):
        return FpInvertBatch(self, nums, pass_zero)

    def div(self, a, b):
        return self.mul(a, self.inv(b))

    def pow(self, n, e):            # tower.ts -> mod.FpPow (modular.ts:666-706): value only
        r, b = self.ONE, n
        while e > 0:
            if e & 1:
                r = self.mul(r, b)
            b = self.sqr(b)
            e >>= 1
        return r

    def frobeniusMap(self, n, power):   # tower.ts:516-521: conjugation for odd powers
        return (n[0], self.Fp.neg(n[1])) if power & 1 else n

    def sqrt(self, num):
        """tower.ts:476-500: complex method over the non-residue -1, root normalised so that
        (im, re) is the lexicographically larger of the two candidates."""
        Fp = self.Fp
        p = Fp.ORDER
        c0, c1 = num

        def legendre(x):            # modular.ts FpLegendre: 1, 0 or -1
            r = pow(x, (p - 1) // 2, p)
            return -1 if r == p - 1 else r

        nonres = p - 1                                   # Fp_NONRESIDUE = -1
        div2 = Fp.div(1, 2)
        if c1 == 0:
            if legendre(c0) == 1:
                return (Fp.sqrt(c0), 0)
            return (0, Fp.sqrt(Fp.div(c0, nonres)))
        a = Fp.sqrt(Fp.sub(Fp.sqr(c0), Fp.mul(Fp.sqr(c1), nonres)))
        d = Fp.mul(Fp.add(a, c0), div2)
        if legendre(d) == -1:
            d = Fp.sub(d, a)
        a0 = Fp.sqrt(d)
        cand = (a0, Fp.div(Fp.mul(c1, div2), a0))
        if not self.eql(self.sqr(cand), num):
            raise ValueError("Cannot find square root")
        x1, x2 = cand, self.neg(cand)
        if x1[1] > x2[1] or (x1[1] == x2[1] and x1[0] > x2[0]):
            return x1
        return x2
