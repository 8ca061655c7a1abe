"""Scalar-multiplication ladders and Pippenger MSM, restating reference
`src/abstract/curve.ts`.  Test infrastructure - see oracle/__init__.py.

Point classes are duck-typed: they provide add/double/negate/equals/toAffine and
their constructor class provides ZERO/BASE/Fp/Fn/fromAffine (curve.ts:56-195).
"""
from .field import FpInvertBatch


def bitLen(n):
    """utils.ts:659 - number of bits of a non-negative bigint."""
    return n.bit_length()


def bitMask(n):
    """utils.ts:725 - (1 << n) - 1."""
    return (1 << n) - 1


def validateMSMPoints(points, c):
    """curve.ts:390-395."""
    if not isinstance(points, (list, tuple)):
        raise TypeError("array expected")
    for i, p in enumerate(points):
        if not isinstance(p, c):
            raise ValueError("invalid point at index %d" % i)


def validateMSMScalars(scalars, field, max_=None):
    """curve.ts:398-404."""
    if not isinstance(scalars, (list, tuple)):
        raise TypeError("array of scalars expected")
    for i, s in enumerate(scalars):
        ok = isinstance(s, int) and not isinstance(s, bool) and (
            field.isValid(s) if max_ is None else 0 <= s < max_)
        if not ok:
            raise ValueError("invalid scalar at index %d" % i)


def normalizeZ(c, points):
    """curve.ts:311-326 - batch to Z=1 through one shared inversion."""
    validateMSMPoints(points, c)
    inv = FpInvertBatch(c.Fp, [p.Z for p in points])
    return [c.fromAffine(p.toAffine(inv[i])) for i, p in enumerate(points)]


def oddMultiples(p, size):
    """curve.ts:420-425 - [1P, 3P, ..., (2*size-1)P]."""
    dbl = p.double()
    t = [p]
    for j in range(1, size):
        t.append(t[j - 1].add(dbl))
    return t


def wnafDigits(n, W):
    """curve.ts:431-447 - width-W wNAF, LSB first, digits 0 or odd, |d| < 2^(W-1)."""
    size = 1 << W
    half = size >> 1
    mask = size - 1
    d = []
    while n > 0:
        w = 0
        if n & 1:
            w = n & mask
            if w >= half:
                w -= size
            n -= w
        d.append(w)
        n >>= 1
    return d


def signedWindowDigits(n, W, windows):
    """curve.ts:454-472 - fixed-position signed windows, digits in [-2^(W-1)+1, 2^(W-1)]."""
    size = 1 << W
    half = size >> 1
    mask = size - 1
    d = []
    for _ in range(windows):
        v = n & mask
        n >>= W
        if v > half:
            v -= size
            n += 1
        d.append(v)
    if n != 0:
        raise ValueError("invalid wnaf")
    return d


def wnafWalk(zero, tables, digits):
    """curve.ts:479-498 - shared-doubling Straus walk, MSB -> LSB."""
    mx = max([len(d) for d in digits] + [0])
    acc = zero
    for bit in range(mx - 1, -1, -1):
        if bit != mx - 1:
            acc = acc.double()
        for i, dg in enumerate(digits):
            w = dg[bit] if bit < len(dg) else 0
            if w:
                item = tables[i][(abs(w) - 1) >> 1]
                acc = acc.add(item.negate() if w < 0 else item)
    return acc


def mulAddUnsafe(c, points, scalars, allowOversized=False):
    """curve.ts:820-836 - sum s_i*P_i by interleaved width-4 wNAF."""
    validateMSMPoints(points, c)
    validateMSMScalars(scalars, c.Fn, c.Fn.ORDER ** 4 if allowOversized else None)
    if len(points) != len(scalars):
        raise ValueError("arrays of points and scalars must have equal length")
    tables = [oddMultiples(p, 4) for p in points]
    digits = [wnafDigits(n, 4) for n in scalars]
    return wnafWalk(c.ZERO, tables, digits)


def pippenger_window(plength):
    """curve.ts:879-883 - window width chosen from the number of points."""
    wbits = bitLen(plength)
    if wbits > 12:
        return wbits - 3
    if wbits > 4:
        return wbits - 2
    if wbits > 0:
        return 2
    return 1


def pippenger(c, points, scalars):
    """curve.ts:863-905 - bucket MSM with unsigned windows; digit 0 is not skipped."""
    fieldN = c.Fn
    validateMSMPoints(points, c)
    validateMSMScalars(scalars, fieldN)
    plength, slength = len(points), len(scalars)
    if plength != slength:
        raise ValueError("arrays of points and scalars must have equal length")
    zero = c.ZERO
    if plength == 0:
        return zero
    windowSize = pippenger_window(plength)
    MASK = bitMask(windowSize)
    nb = MASK + 1
    lastBits = ((fieldN.BITS - 1) // windowSize) * windowSize
    total = zero
    for i in range(lastBits, -1, -windowSize):
        buckets = [zero] * nb
        for j in range(slength):
            wb = (scalars[j] >> i) & MASK
            buckets[wb] = buckets[wb].add(points[j])
        resI = zero
        sumI = zero
        for j in range(nb - 1, 0, -1):
            sumI = sumI.add(buckets[j])
            resI = resI.add(sumI)
        total = total.add(resI)
        if i != 0:
            for _ in range(windowSize):
                total = total.double()
    return total


def pippenger_op_count(plength, fn_bits):
    """(adds, doublings) executed by curve.ts:863-905 for `plength` points - the
    reference-equivalent work figure used for roofline accounting."""
    c = pippenger_window(plength)
    nwin = (fn_bits - 1) // c + 1
    adds = nwin * (plength + 2 * ((1 << c) - 1) + 1)
    dbls = (nwin - 1) * c
    return adds, dbls


def wnafCachedMul(c, base, n, W, bits):
    """Value computed by ScalarMultiplier.buildWnafTable + wnafCachedCT
    (curve.ts:560-606) for a precomputed base: sum over windows of digit*2^(w*W)*base.
    The fake accumulator `f` only exists for constant time and is dropped."""
    windows = -(-bits // W) + 1      # curve.ts:564 ceil(bits/W)+1
    half = 1 << (W - 1)
    comp = []
    b = base
    for _ in range(windows):         # curve.ts:567-575
        acc = b
        for _i in range(half):
            comp.append(acc)
            acc = acc.add(b)
        b = comp[-1].double()
    digits = signedWindowDigits(n, W, windows)
    acc = c.ZERO
    for w in range(windows):
        d = digits[w]
        if d == 0:
            continue
        sel = comp[w * half + abs(d) - 1]
        acc = acc.add(sel.negate() if d < 0 else sel)
    return acc


def naiveMul(c, p, n):
    """Independent double-and-add used by the reference's own tests
    (test/point.test.ts:561-570)."""
    acc = c.ZERO
    d = p
    while n > 0:
        if n & 1:
            acc = acc.add(d)
        d = d.double()
        n >>= 1
    return acc
