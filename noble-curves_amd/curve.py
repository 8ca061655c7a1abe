"""Host-side mirror of the reference's Point / pippenger interface for the hot path.

Names, argument meaning and error messages follow the reference so the parity tests read like
its own tests:
  * `Point` classes per curve with `BASE`, `ZERO`, `Fp`, `Fn`, `fromAffine`, `toAffine`, `equals`,
    `negate`, `is0`, `multiply`, `multiplyUnsafe`              (src/abstract/curve.ts:56-195,
    src/abstract/weierstrass.ts:685-969)
  * `pippenger(c, points, scalars)`                            (src/abstract/curve.ts:863-905)
  * array entry points `multiplyUnsafeBatch` / `multiplyBatch` (a single scalar-mult
    is far below launch + PCIe cost, so the shim adds batch forms)

All group arithmetic happens in the HIP kernels behind `libncg.so`; this file only validates
inputs (throwing the reference's messages BEFORE crossing the boundary), marshals to the wire
format of include/ncg.h and wraps results.  Points are held in affine form (x, y) or ZERO - the
reference's projective (X, Y, Z) triples are representation details that are not part of the
contract.  There is no CPU fallback: without the library / a GPU every operation
raises NativeError.
"""
import numpy as np

from . import _native
from .field import Field
from ._native import BLS12_381_G1, BLS12_381_G2, ED25519, FIELD_BYTES, POINT_BYTES, SECP256K1, get_engine

_BLS_P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
_BLS_R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


class _Field(Field):
    """Fp / Fn of a point class: the full prime-field contract of field.py (IField, src/abstract/modular.ts:429-607);
    `degree` 2 marks bls12-381's Fp2 base field, whose elements are (c0, c1) pairs and for which only the
    membership tests the validators need are defined here."""

    def __init__(self, order, degree=1, isLE=False):
        super().__init__(order, isLE=isLE)
        self.degree = degree

    def isValid(self, n):
        if self.degree == 2:
            return (isinstance(n, tuple) and len(n) == 2 and all(isinstance(c, int) and 0 <= c < self.ORDER for c in n))
        return super().isValid(n)

    def isValidNot0(self, n):
        return self.isValid(n) and not self.is0(n)

    def is0(self, n):
        return n == (0, 0) if self.degree == 2 else n == 0


def _make_point_class(name, curve_id, Fp, Fn, gx, gy):
    fb = FIELD_BYTES[curve_id]
    pb = POINT_BYTES[curve_id]
    zero_xy = ((0, 0), (0, 0)) if Fp.degree == 2 else (0, 0)

    class Point:
        """Affine point or ZERO of %s (reference: weierstrass.ts:685-1020)."""
        __slots__ = ("x", "y", "_inf")
        CURVE_ID = curve_id
        POINT_BYTES = pb

        def __init__(self, x, y, _inf=False):
            self.x, self.y, self._inf = x, y, _inf

        # -- construction -------------------------------------------------------------------
        @classmethod
        def fromAffine(cls, p):
            """weierstrass.ts:710-718: validates coordinate ranges only; (0,0) is ZERO."""
            x, y = (p["x"], p["y"]) if isinstance(p, dict) else p
            if not Fp.isValid(x) or not Fp.isValid(y):
                raise ValueError("invalid affine point")
            if Fp.is0(x) and Fp.is0(y):
                return cls.ZERO
            return cls(x, y)

        def toAffine(self):
            """weierstrass.ts:951-969; ZERO -> (0, 0)."""
            return zero_xy if self._inf else (self.x, self.y)

        # -- cheap host-side predicates (no field arithmetic) ---------------------------------
        def is0(self):
            return self._inf

        def equals(self, other):
            _apoint(other)
            return self.toAffine() == other.toAffine()

        def negate(self):
            if self._inf:
                return self
            if Fp.degree == 2:
                return Point(self.x, tuple((-c) % Fp.ORDER for c in self.y))
            return Point(self.x, (-self.y) % Fp.ORDER)

        # -- scalar multiplication (GPU, batch of one) ----------------------------------------
        def multiply(self, scalar):
            """weierstrass.ts:900-907: 1 <= scalar < n, result normalised."""
            return multiplyBatch(Point, [self], [scalar])[0]

        def multiplyUnsafe(self, scalar):
            """weierstrass.ts:915-928: 0 <= scalar < n."""
            return multiplyUnsafeBatch(Point, [self], [scalar])[0]

        def add(self, other):
            """Group addition through the MSM path with unit scalars."""
            _apoint(other)
            return pippenger(Point, [self, other], [1, 1])

        def double(self):
            return self.multiplyUnsafe(2) if not self._inf else self

        def subtract(self, other):
            _apoint(other)
            return self.add(other.negate())

        # -- wire format ----------------------------------------------------------------------
        def _wire(self):
            if self._inf:
                return b"\x00" * pb
            cs = (self.x + self.y) if Fp.degree == 2 else (self.x, self.y)
            return b"".join(int(c).to_bytes(fb, "little") for c in cs)

        @classmethod
        def _from_wire(cls, row, inf):
            if inf:
                return cls.ZERO
            vals = [int.from_bytes(bytes(row[i * fb:(i + 1) * fb]), "little") for i in range(pb // fb)]
            if Fp.degree == 2:
                return cls((vals[0], vals[1]), (vals[2], vals[3]))
            return cls(vals[0], vals[1])

        def __repr__(self):
            return "%s.Point.ZERO" % name if self._inf else "%s.Point(%r, %r)" % (name, self.x, self.y)

    def _apoint(o):
        if not isinstance(o, Point):
            raise TypeError("Weierstrass Point expected")

    Point.__name__ = name + "Point"
    Point.Fp = Fp
    Point.Fn = Fn
    Point.ZERO = Point(None, None, True)
    Point.BASE = Point(gx, gy)
    return Point


def _make_edwards_point_class(name, curve_id, Fp, Fn, gx, gy):
    """Twisted-Edwards point class (reference: src/abstract/edwards.ts:368-660): affine (x, y),
    ZERO = (0, 1) (:370, :606); coordinates are accepted below 2^(8*Fp.BYTES) like the reference
    (ZIP-215 keeps unreduced y, :356-360) and reduced mod p when marshalled (SURVEY 8a gotcha 4)."""
    MASK = 1 << (8 * Fp.BYTES)
    P = Fp.ORDER

    class Point:
        __slots__ = ("x", "y")
        CURVE_ID = curve_id
        POINT_BYTES = 64

        def __init__(self, x, y):
            self.x, self.y = x, y

        @classmethod
        def fromAffine(cls, p):
            x, y = (p["x"], p["y"]) if isinstance(p, dict) else p
            for t, v in (("x", x), ("y", y)):
                if not (isinstance(v, int) and not isinstance(v, bool) and 0 <= v < MASK):
                    raise ValueError("expected valid coordinate %s" % t)
            return cls(x, y)

        def toAffine(self):
            return (self.x % P, self.y % P)

        def is0(self):
            return self.toAffine() == (0, 1)

        def equals(self, other):
            _apoint(other)
            return self.toAffine() == other.toAffine()

        def negate(self):
            return Point((-self.x) % P, self.y % P)

        def multiply(self, scalar):
            """edwards.ts:555-564: 1 <= scalar < n."""
            if not (isinstance(scalar, int) and not isinstance(scalar, bool) and 1 <= scalar < Fn.ORDER):
                raise ValueError("invalid scalar: expected 1 <= sc < curve.n")
            return multiplyUnsafeBatch(Point, [self], [scalar], _err="invalid scalar: expected 0 <= sc < curve.n")[0]

        def multiplyUnsafe(self, scalar):
            """edwards.ts:571-577: 0 <= scalar < n."""
            return multiplyUnsafeBatch(Point, [self], [scalar], _err="invalid scalar: expected 0 <= sc < curve.n")[0]

        def add(self, other):
            _apoint(other)
            return pippenger(Point, [self, other], [1, 1])

        def double(self):
            return self.multiplyUnsafe(2)

        def subtract(self, other):
            _apoint(other)
            return self.add(other.negate())

        def _wire(self):
            x, y = self.toAffine()
            return x.to_bytes(32, "little") + y.to_bytes(32, "little")

        @classmethod
        def _from_wire(cls, row, inf):
            if inf:
                return cls.ZERO
            return cls(int.from_bytes(bytes(row[:32]), "little"), int.from_bytes(bytes(row[32:64]), "little"))

        def __repr__(self):
            return "%s.Point(%r, %r)" % (name, self.x, self.y)

    def _apoint(o):
        if not isinstance(o, Point):
            raise TypeError("EdwardsPoint expected")

    Point.__name__ = name + "Point"
    Point.Fp = Fp
    Point.Fn = Fn
    Point.ZERO = Point(0, 1)
    Point.BASE = Point(gx, gy)
    return Point


# ---------------------------------------------------------------------------------- validation
def validateMSMPoints(points, c):
    """curve.ts:390-395."""
    if not isinstance(points, (list, tuple)):
        raise TypeError("array expected")
    for i, p in enumerate(points):
        if not isinstance(p, c):
            raise ValueError("invalid point at index %d" % i)


def validateMSMScalars(scalars, field):
    """curve.ts:398-404."""
    if not isinstance(scalars, (list, tuple)):
        raise TypeError("array of scalars expected")
    for i, s in enumerate(scalars):
        if not (isinstance(s, int) and not isinstance(s, bool) and 0 <= s < field.ORDER):
            raise ValueError("invalid scalar at index %d" % i)


def _points_wire(points, pb):
    arr = np.frombuffer(b"".join(p._wire() for p in points), dtype=np.uint8)
    return arr.reshape(-1, pb) if len(points) else np.zeros((0, pb), np.uint8)


def _scalars_wire(scalars):
    return _native.ints_to_le(scalars, 32)


# ---------------------------------------------------------------------------------- entry points
def pippenger(c, points, scalars, engine=None):
    """MSM sum_i scalars[i]*points[i] (curve.ts:863-905): same validation order and messages;
    empty input returns ZERO (:878); zero scalars and ZERO points are allowed."""
    if isinstance(points, PointSet):               # resident set: validated at upload, only scalars cross
        if points.c is not c:
            raise ValueError("invalid point at index 0")
        validateMSMScalars(scalars, c.Fn)
        if len(points) != len(scalars):
            raise ValueError("arrays of points and scalars must have equal length")
        if len(points) == 0:
            return c.ZERO
        out, inf = points.resident.msm(_scalars_wire(scalars))
        return c._from_wire(out, inf)
    validateMSMPoints(points, c)
    validateMSMScalars(scalars, c.Fn)
    if len(points) != len(scalars):
        raise ValueError("arrays of points and scalars must have equal length")
    if len(points) == 0:
        return c.ZERO
    eng = engine or get_engine()
    out, inf = eng.msm(c.CURVE_ID, _points_wire(points, c.POINT_BYTES), _scalars_wire(scalars))
    return c._from_wire(out, inf)


class PointSet:
    """A validated point set resident on the GPU: `uploadPoints` / `uploadEncoded` make one, `pippenger`
    and `multiplyUnsafeBatch` accept it in place of the point list - only the scalars cross per call.
    The reference's counterpart is the closure of interleavedMSMUnsafe (curve.ts:907-959): precompute
    for a fixed point set once, then call with scalars."""

    def __init__(self, c, resident):
        self.c, self.resident = c, resident

    def __len__(self):
        return len(self.resident)

    @property
    def inSubgroup(self):
        """bls12-381: every point is known to be torsion-free (decoded by fromBytes, or verified at upload),
        so `pippenger` on this set splits the scalars along the curve endomorphism - same result, half (G1) or
        a quarter (G2) of the windows."""
        return self.resident.in_subgroup

    def free(self):
        self.resident.free()


def uploadPoints(c, points, engine=None, checkSubgroup=False):
    """Validate like pippenger (curve.ts:390-395) and keep the points on the device.  checkSubgroup (bls12-381
    G1 / G2): also run p.isTorsionFree() (bls12-381.ts:567-577, :599-601) on every point, once, on the device;
    if all pass the set takes the faster endomorphism MSM - pippenger accepts points outside the subgroup, so
    this is never assumed, and a set with such a point simply keeps the generic path."""
    validateMSMPoints(points, c)
    eng = engine or get_engine()
    pset = PointSet(c, eng.upload_points(c.CURVE_ID, _points_wire(points, c.POINT_BYTES)))
    if checkSubgroup and c.CURVE_ID in (BLS12_381_G1, BLS12_381_G2) and len(points):
        pset.resident.verify_subgroup()
    return pset


def uploadEncoded(c, encodings, zip215=False, engine=None):
    """Point set from compressed encodings (c.fromBytes(b) for each b), decoded and validated on the
    device; an entry the reference would reject raises ValueError naming its index."""
    size = _native.ENCODED_BYTES.get(c.CURVE_ID)
    if size is None:
        raise ValueError("noble-gpu: no batch decoder for this curve")
    if isinstance(encodings, np.ndarray):
        rows = np.ascontiguousarray(encodings, dtype=np.uint8).reshape(-1, size)
    else:
        for i, b in enumerate(encodings):
            if len(bytes(b)) != size:
                raise ValueError("invalid point encoding at index %d: expected %d bytes" % (i, size))
        rows = np.frombuffer(b"".join(bytes(b) for b in encodings), dtype=np.uint8).reshape(-1, size)
    eng = engine or get_engine()
    res, bad = eng.upload_encoded(c.CURVE_ID, rows, zip215)
    if res is None:
        raise ValueError("invalid point encoding at index %d" % bad)
    return PointSet(c, res)


def interleavedMSMUnsafe(c, points, windowSize, engine=None):
    """curve.ts:938-959: MSM over a FIXED point set; returns the closure `scalars -> Point`.  Same
    argument checks and messages (window in [2..Fn.BITS], validateMSMPoints); the closure accepts at most
    len(points) scalars and treats omitted trailing ones as zero.  `windowSize` only sizes the reference's
    per-point wNAF tables and does not change the result: here the set is uploaded once, its precomputation is the
    device's own (window-shifted copies of the points, ncg_points_precompute, for sets of >= 4096 points), and
    every call runs the bucket MSM on the resident points with only the scalars crossing."""
    bits = c.Fn.BITS
    if not (isinstance(windowSize, int) and not isinstance(windowSize, bool) and 2 <= windowSize <= bits):
        raise ValueError("invalid window size, expected [2..%d], got W=%s" % (bits, windowSize))
    validateMSMPoints(points, c)
    n = len(points)
    pset = uploadPoints(c, points, engine) if n else None
    if pset is not None:
        pset.resident.precompute()          # the reference builds its per-point tables here too (curve.ts:948)

    def msm(scalars):
        validateMSMScalars(scalars, c.Fn)
        if len(scalars) > n:
            raise ValueError("array of scalars must not be larger than array of points")
        if n == 0:
            return c.ZERO
        out, inf = pset.resident.msm(_scalars_wire(list(scalars) + [0] * (n - len(scalars))))
        return c._from_wire(out, inf)

    return msm


def multiplyUnsafeBatch(c, points, scalars, engine=None, _err="invalid scalar: out of range"):
    """[p.multiplyUnsafe(k) for p, k in zip(points, scalars)] in one launch
    (weierstrass.ts:915-928: 0 <= k < n else RangeError('invalid scalar: out of range'))."""
    resident = points if isinstance(points, PointSet) else None
    if resident is None:
        validateMSMPoints(points, c)
    elif resident.c is not c:
        raise ValueError("invalid point at index 0")
    if len(points) != len(scalars):
        raise ValueError("arrays of points and scalars must have equal length")
    for k in scalars:
        if not (isinstance(k, int) and not isinstance(k, bool) and 0 <= k < c.Fn.ORDER):
            raise ValueError(_err)
    if len(points) == 0:
        return []
    if resident is not None:
        out, inf = resident.resident.mul_var_batch(_scalars_wire(scalars))
    else:
        eng = engine or get_engine()
        out, inf = eng.mul_var_batch(c.CURVE_ID, _points_wire(points, c.POINT_BYTES), _scalars_wire(scalars))
    return [c._from_wire(out[i], bool(inf[i])) for i in range(len(points))]


def fromBytesBatch(c, encodings, zip215=False, engine=None):
    """[c.fromBytes(b) for b in encodings] in one launch; entries the reference would reject
    (bad prefix / flags, x or y out of range, no square root, point outside the prime-order
    subgroup on bls12-381) come back as None instead of raising.  Encodings: secp256k1 33-byte
    SEC1 compressed (weierstrass.ts:566-605), bls12-381 G1 48-byte / G2 96-byte compressed
    (bls12-381.ts:377-459, :567-577, :599-601), ed25519 32 bytes (edwards.ts:405-436, `zip215`)."""
    size = {SECP256K1: 33, ED25519: 32, BLS12_381_G1: 48, BLS12_381_G2: 96}.get(c.CURVE_ID)
    if size is None:
        raise ValueError("noble-gpu: no batch decoder for this curve")
    if not encodings:
        return []
    rows = []
    for i, b in enumerate(encodings):
        b = bytes(b)
        if len(b) != size:
            raise ValueError("invalid point encoding at index %d: expected %d bytes" % (i, size))
        rows.append(np.frombuffer(b, dtype=np.uint8))
    eng = engine or get_engine()
    out, ok, inf = eng.decode_points_batch(c.CURVE_ID, np.array(rows), zip215)
    return [c._from_wire(out[i], bool(inf[i])) if ok[i] else None for i in range(len(rows))]


def toBytesBatch(c, points, engine=None):
    """[p.toBytes() for p in points] (compressed form) in one launch: secp256k1 SEC1 33 bytes
    (weierstrass.ts:541-564), bls12-381 G1 48 / G2 96 bytes (bls12-381.ts:400-410), ed25519 32 bytes
    (edwards.ts:620-628).  Non-normalised inputs are batch-normalised first (curve.ts:311-326).
    secp256k1 ZERO raises like the reference ('bad point: ZERO')."""
    if c.CURVE_ID not in (SECP256K1, ED25519, BLS12_381_G1, BLS12_381_G2):
        raise ValueError("noble-gpu: no batch encoder for this curve")
    validateMSMPoints(points, c)
    if not points:
        return []
    eng = engine or get_engine()
    enc, ok = eng.encode_points_batch(c.CURVE_ID, _points_wire(points, c.POINT_BYTES))
    for i in range(len(points)):
        if not ok[i]:
            raise ValueError("bad point: ZERO")
    return [enc[i].tobytes() for i in range(len(points))]


def addBatch(c, ps, qs, engine=None, _subtract=False):
    """[p.add(q) for p, q in zip(ps, qs)] (weierstrass.ts:834-880, edwards.ts:526-545) in one launch."""
    validateMSMPoints(ps, c)
    validateMSMPoints(qs, c)
    if len(ps) != len(qs):
        raise ValueError("arrays of points must have equal length")
    if not ps:
        return []
    eng = engine or get_engine()
    out, inf = eng.add_pairs_batch(c.CURVE_ID, _points_wire(ps, c.POINT_BYTES), _points_wire(qs, c.POINT_BYTES), _subtract)
    return [c._from_wire(out[i], bool(inf[i])) for i in range(len(ps))]


def subtractBatch(c, ps, qs, engine=None):
    """[p.subtract(q) ...] (weierstrass.ts:882-885)."""
    return addBatch(c, ps, qs, engine, _subtract=True)


def mulAddUnsafeBatch(c, ps, a_s, qs, b_s, engine=None):
    """[p.mulAddUnsafe(a, q, b) ...] = a*p + b*q per item (weierstrass.ts:937-944; the u1*G + u2*P of ECDSA
    verification and public-key recovery, :1403, :1609): two batch multiplies and one pairwise addition,
    scalars 0 <= k < n like multiplyUnsafe (:915-928)."""
    if not (len(ps) == len(a_s) == len(qs) == len(b_s)):
        raise ValueError("arrays of points and scalars must have equal length")
    if ps and all(p is c.BASE or (not p.is0() and p.toAffine() == c.BASE.toAffine()) for p in ps):
        for k in a_s:                              # same range rule as multiplyUnsafe
            if not (isinstance(k, int) and not isinstance(k, bool) and 0 <= k < c.Fn.ORDER):
                raise ValueError("invalid scalar: out of range")
        A = multiplyBaseBatch(c, list(a_s), engine, unsafe=True)   # u1*G through the fixed-base table
    else:
        A = multiplyUnsafeBatch(c, ps, a_s, engine)
    B = multiplyUnsafeBatch(c, qs, b_s, engine)
    return addBatch(c, A, B, engine)


def aggregateFromBytes(c, encodings, zip215=False, engine=None):
    """sum(c.fromBytes(b) for b in encodings) - the group part of bls.aggregatePublicKeys /
    aggregateSignatures on encoded inputs (src/abstract/bls.ts:857-873).  Decoding, validity and
    subgroup checks and the sum all run on the device; an entry the reference's fromBytes would
    reject raises ValueError naming its index.  Empty input raises like the reference's aNonEmpty guard
    (bls.ts:426-431 'expected non-empty array')."""
    size = _native.ENCODED_BYTES.get(c.CURVE_ID)
    if size is None:
        raise ValueError("noble-gpu: no batch decoder for this curve")
    rows = []
    for i, b in enumerate(encodings):
        b = bytes(b)
        if len(b) != size:
            raise ValueError("invalid point encoding at index %d: expected %d bytes" % (i, size))
        rows.append(np.frombuffer(b, dtype=np.uint8))
    if not rows:
        raise ValueError("expected non-empty array")
    eng = engine or get_engine()
    out, inf, bad = eng.aggregate_encoded(c.CURVE_ID, np.array(rows), zip215)
    if bad >= 0:
        raise ValueError("invalid point encoding at index %d" % bad)
    return c._from_wire(out, inf)


def isTorsionFreeBatch(c, points, engine=None):
    """[p.isTorsionFree() for p in points]: membership in the prime-order subgroup, decided as the
    reference's generic test does - [n]P == ZERO with n = Fn.ORDER (weierstrass.ts:976-981,
    edwards.ts:589-591) - one batch multiply by the group order.  (The bls12-381 endomorphism tests of
    bls12-381.ts:567-577 / :599-601 give the same booleans; the decoders use those.)  secp256k1 has
    cofactor 1: every curve point qualifies."""
    validateMSMPoints(points, c)
    if not points:
        return []
    if c.CURVE_ID == SECP256K1:
        return [True] * len(points)
    eng = engine or get_engine()
    n = len(points)
    _, inf = eng.mul_var_batch(c.CURVE_ID, _points_wire(points, c.POINT_BYTES), _scalars_wire([c.Fn.ORDER] * n))
    return [bool(f) for f in inf]


def clearCofactorBatch(c, points, engine=None):
    """[p.clearCofactor() for p in points] where that is an integer multiple that fits the batch
    multiply: ed25519 [8]P (edwards.ts:611-618), bls12-381 G1 [|x| + 1]P = [x]P + P with the
    reference's positive BLS_X (bls12-381.ts:578-581), secp256k1 the identity map.  (G2's psi-based
    clearing is part of `h2c.bls12_381_G2_hasher`.)"""
    validateMSMPoints(points, c)
    if not points:
        return []
    if c.CURVE_ID == SECP256K1:
        return list(points)
    if c.CURVE_ID == ED25519:
        k = 8
    elif c.CURVE_ID == BLS12_381_G1:
        k = 0xD201000000010000 + 1
    else:
        raise ValueError("noble-gpu: clearCofactorBatch: no integer-multiple form for this curve")
    eng = engine or get_engine()
    out, inf = eng.mul_var_batch(c.CURVE_ID, _points_wire(points, c.POINT_BYTES), _scalars_wire([k] * len(points)))
    return [c._from_wire(out[i], bool(inf[i])) for i in range(len(points))]


def sumPoints(c, points, engine=None):
    """Sum of a batch of points - the group operation behind bls.aggregatePublicKeys /
    aggregateSignatures (src/abstract/bls.ts:857-873).  Runs as an MSM with unit
    scalars: every point lands in one bucket, which the balanced accumulation and the tree
    fix-up turn into a parallel reduction."""
    return pippenger(c, list(points), [1] * len(points), engine)


def multiplyBaseBatch(c, scalars, engine=None, unsafe=False):
    """[c.BASE.multiply(k) for k in scalars] through the fixed-base window table
    (curve.ts:588-606 wnafCachedCT; e.g. getPublicKey).  `unsafe` allows k = 0 like multiplyUnsafe."""
    lo = 0 if unsafe else 1
    for k in scalars:
        if not (isinstance(k, int) and not isinstance(k, bool) and lo <= k < c.Fn.ORDER):
            raise ValueError("invalid scalar: out of range")
    if not scalars:
        return []
    eng = engine or get_engine()
    out, inf = eng.mul_base_batch(c.CURVE_ID, _scalars_wire(scalars))
    return [c._from_wire(out[i], bool(inf[i])) for i in range(len(scalars))]


def multiplyBatch(c, points, scalars, engine=None):
    """[p.multiply(k) ...] (weierstrass.ts:900-907: 1 <= k < n).  Same group element as
    multiplyUnsafe, returned normalised (the kernels always return affine points).  The GPU path
    is not constant-time: use it for public scalars only."""
    for k in scalars:
        if not (isinstance(k, int) and not isinstance(k, bool) and 1 <= k < c.Fn.ORDER):
            raise ValueError("invalid scalar: out of range")
    return multiplyUnsafeBatch(c, points, scalars, engine)


def normalizeZ(c, points):
    """curve.ts:311-326.  Points of this shim are always affine (Z = 1), so this validates and
    returns equal points; use `normalizeProjective` for the reference's (X, Y, Z) triples."""
    validateMSMPoints(points, c)
    return list(points)


def normalizeProjective(c, triples, engine=None):
    """Batch (X, Y, Z) -> Point with x = X/Z, y = Y/Z: what normalizeZ / toAffine(invZ) compute for
    the reference's projective points (curve.ts:311-326, weierstrass.ts:951-969, edwards.ts:595-609),
    one shared inversion per 8 points on the GPU (FpInvertBatch, modular.ts:728-760)."""
    fb, deg = c.Fp.BYTES, c.Fp.degree
    if not triples:
        return []
    rows = []
    for i, t in enumerate(triples):
        if len(t) != 3 or not all(c.Fp.isValid(v) for v in t):
            raise ValueError("invalid point at index %d" % i)
        cs = [v for co in t for v in (co if deg == 2 else (co,))]
        rows.append(b"".join(int(v).to_bytes(fb, "little") for v in cs))
    arr = np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(triples), -1)
    eng = engine or get_engine()
    out, inf = eng.normalize_batch(c.CURVE_ID, arr)
    return [c._from_wire(out[i], bool(inf[i])) for i in range(len(triples))]


# ---------------------------------------------------------------------------------- curve instances
secp256k1_Point = _make_point_class(
    "secp256k1", SECP256K1,
    _Field(0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2F),
    _Field(0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141),
    0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
    0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)                # secp256k1.ts:48-56
bls12_381_G1_Point = _make_point_class(
    "bls12_381_G1", BLS12_381_G1, _Field(_BLS_P), _Field(_BLS_R),
    0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
    0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1)  # bls12-381.ts:134-148
bls12_381_G2_Point = _make_point_class(
    "bls12_381_G2", BLS12_381_G2, _Field(_BLS_P, degree=2), _Field(_BLS_R),
    (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
     0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
    (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
     0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE))  # bls12-381.ts:321-345
ed25519_Point = _make_edwards_point_class(
    "ed25519", ED25519, _Field((1 << 255) - 19, isLE=True),         # curve.ts:1036: Edwards fields are little-endian
    _Field(0x1000000000000000000000000000000014DEF9DEA2F79CD65812631A5CF5D3ED, isLE=True),
    0x216936D3CD6E53FEC0A4E231FDD6DC5C692CC7609525A7B2C9562D608F25D51A,
    0x6666666666666666666666666666666666666666666666666666666666666658)                # ed25519.ts:57-65
