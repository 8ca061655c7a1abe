"""Host mirror of the reference's hash-to-curve interface for bls12-381 G1 / G2
(`bls12_381.G1.hashToCurve`, `.encodeToCurve`, `.mapToCurve` - createHasher,
src/abstract/hash-to-curve.ts:441-548; suite options src/bls12-381.ts:305-313, :628-634) plus the
batch forms the GPU needs.  The shim does the byte hashing exactly as the reference does
(expand_message_xmd over SHA-256, hash_to_field: :189-228, :312-378, here with hashlib); all
field and curve arithmetic runs in `libncg.so` (`ncg_map_to_curve_batch`).
"""
import hashlib

import numpy as np

from . import curve as _curve
from ._native import BLS12_381_G1, BLS12_381_G2, get_engine

_P = _curve.bls12_381_G1_Point.Fp.ORDER


def _bytes(x, title):
    if isinstance(x, str):
        return x.encode("ascii")
    if isinstance(x, (bytes, bytearray, memoryview)):
        return bytes(x)
    raise TypeError('"%s" expected Uint8Array, got type=%s' % (title, type(x).__name__))


def expand_message_xmd(msg, DST, lenInBytes, H=hashlib.sha256):
    """hash-to-curve.ts:189-228 (RFC 9380 5.3.1)."""
    msg, DST = _bytes(msg, "msg"), _bytes(DST, "DST")
    if not isinstance(lenInBytes, int) or isinstance(lenInBytes, bool) or lenInBytes < 0:
        raise ValueError("invalid lenInBytes")
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


def hash_to_field(msg, count, options):
    """hash-to-curve.ts:312-378 for expand = 'xmd' / hash = sha256: count x m integers mod p."""
    p, m, k, DST = options["p"], options["m"], options["k"], options["DST"]
    if p <= 1:
        raise ValueError("hash_to_field: expected valid field characteristic")
    if count < 1:
        raise ValueError("hash_to_field: expected count >= 1")
    if m < 1:
        raise ValueError("hash_to_field: expected m >= 1")
    if k < 0:
        raise ValueError("hash_to_field: invalid k")
    L = -(-(p.bit_length() + k) // 8)
    prb = expand_message_xmd(msg, DST, count * m * L)
    return [[int.from_bytes(prb[L * (j + i * m):L * (j + i * m) + L], "big") % p for j in range(m)]
            for i in range(count)]


class H2CHasher:
    """createHasher(Point, mapToCurve, defaults) for one bls12-381 group; `*Batch` methods take
    lists and make one device launch."""

    def __init__(self, Point, curve_id, m, DST, engine=None):
        self.Point, self._cid, self._engine = Point, curve_id, engine
        self.defaults = {"DST": DST, "encodeDST": DST, "p": _P, "m": m, "k": 128, "expand": "xmd", "hash": "sha256"}

    def _opts(self, options, key="DST"):
        o = dict(self.defaults)
        o["DST"] = self.defaults[key]
        if options and options.get("DST") is not None:
            o["DST"] = options["DST"]
        return o

    def _launch(self, us, count):
        """us: per output point, `count` elements of m integers each."""
        m = self.defaults["m"]
        raw = bytearray()
        for elems in us:
            for e in elems:
                for c in e:
                    raw += int(c).to_bytes(48, "little")
        eng = self._engine or get_engine()
        u = np.frombuffer(bytes(raw), dtype=np.uint8).reshape(len(us), count * m * 48)
        out, inf = eng.map_to_curve_batch(self._cid, u, count)
        return [self.Point._from_wire(out[i], bool(inf[i])) for i in range(len(us))]

    # ---- batch forms -------------------------------------------------------------------------
    def hashToCurveBatch(self, msgs, options=None):
        o = self._opts(options)
        return self._launch([hash_to_field(_bytes(m, "msg"), 2, o) for m in msgs], 2) if msgs else []

    def encodeToCurveBatch(self, msgs, options=None):
        o = self._opts(options, "encodeDST")
        return self._launch([hash_to_field(_bytes(m, "msg"), 1, o) for m in msgs], 1) if msgs else []

    def mapToCurveBatch(self, scalars_list):
        m = self.defaults["m"]
        us = []
        for scalars in scalars_list:
            if m == 1:
                if not isinstance(scalars, int) or isinstance(scalars, bool):
                    raise ValueError("expected bigint (m=1)")
                scalars = [scalars]
            else:
                if not isinstance(scalars, (list, tuple)):
                    raise ValueError("expected array of bigints")
                if len(scalars) != m:
                    raise ValueError("expected array of %d bigints" % m)
                for i in scalars:
                    if not isinstance(i, int) or isinstance(i, bool):
                        raise ValueError("expected array of bigints")
            us.append([[s % _P for s in scalars]])          # Fp.create (bls12-381.ts:854, :860)
        return self._launch(us, 1) if us else []

    # ---- the reference's single-call forms -----------------------------------------------------
    def hashToCurve(self, msg, options=None):
        return self.hashToCurveBatch([msg], options)[0]

    def encodeToCurve(self, msg, options=None):
        return self.encodeToCurveBatch([msg], options)[0]

    def mapToCurve(self, scalars):
        return self.mapToCurveBatch([scalars])[0]


bls12_381_G1_hasher = H2CHasher(_curve.bls12_381_G1_Point, BLS12_381_G1, 1, "BLS_SIG_BLS12381G1_XMD:SHA-256_SSWU_RO_NUL_")
bls12_381_G2_hasher = H2CHasher(_curve.bls12_381_G2_Point, BLS12_381_G2, 2, "BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_NUL_")
