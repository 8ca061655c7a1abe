"""Host mirror of the reference's FFT interface for the bls12-381 scalar field
(src/abstract/fft.ts): `rootsOfUnity(Fr, 7)` / `FFT(roots, Fr).direct|inverse(values, brpInput,
brpOutput)` with the same names, argument meaning and error messages; the transform itself runs in
`libncg.so` (`ncg_ntt`).  The host side only does what the reference does once per field: the
2-adic chain of primitive roots (:238-241) - a handful of modular exponentiations.
"""
import numpy as np

from . import _native
from ._native import get_engine

BLS12_381_FR_ORDER = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


class _Fr:
    """The slice of IField (src/abstract/modular.ts:429-607) the FFT front end touches."""
    ORDER = BLS12_381_FR_ORDER
    BITS = 255
    BYTES = 32
    ONE = 1
    ZERO = 0

    @staticmethod
    def pow(a, e):
        return pow(a, e, BLS12_381_FR_ORDER)


bls12_381_Fr = _Fr()


def isPowerOfTwo(x):                       # fft.ts:56-59
    return isinstance(x, int) and x > 0 and (x & (x - 1)) == 0


def nextPowerOfTwo(n):                     # fft.ts:72-76
    return 1 if n <= 1 else 1 << (n - 1).bit_length()


def log2(n):                               # fft.ts:116-119
    return n.bit_length() - 1


def reverseBits(n, bits):                  # fft.ts:93-99
    r = 0
    for _ in range(bits):
        r = (r << 1) | (n & 1)
        n >>= 1
    return r


def bitReversalPermutation(values):        # fft.ts:136-171 (copying form)
    n = len(values)
    if n < 2 or not isPowerOfTwo(n):
        raise ValueError("n must be a power of 2 and greater than 1. Got " + str(n))
    bits = log2(n)
    return [values[reverseBits(i, bits)] for i in range(n)]


class RootsOfUnity:
    """fft.ts:230-312 for a field whose ORDER matches the device field (bls12-381 Fr)."""

    def __init__(self, field, generator=None):
        if getattr(field, "ORDER", None) != BLS12_381_FR_ORDER:
            raise ValueError("noble-gpu: the device NTT is built for the bls12-381 scalar field only")
        if generator is not None and (not isinstance(generator, int) or isinstance(generator, bool)):
            raise TypeError('"generator" expected bigint, got type=' + type(generator).__name__)
        odd, p2 = field.ORDER - 1, 0
        while odd & 1 == 0:
            odd >>= 1
            p2 += 1
        if generator is None:              # findGenerator :175-180
            generator = 2
            while pow(generator, field.ORDER >> 1, field.ORDER) == 1:
                generator += 1
        self.field = field
        self.info = {"G": generator, "oddFactor": odd, "powerOfTwo": p2}
        self._omegas = [0] * (p2 + 1)
        self._omegas[p2] = pow(generator, odd, field.ORDER)
        for i in range(p2, 0, -1):
            self._omegas[i - 1] = self._omegas[i] * self._omegas[i] % field.ORDER
        self._cache = {}

    def _check(self, bits):
        if not isinstance(bits, int) or isinstance(bits, bool) or bits < 0:
            raise ValueError("wrong u32 integer: bits")
        if bits > 31 or bits > self.info["powerOfTwo"]:
            raise ValueError("rootsOfUnity: wrong bits %d powerOfTwo=%d" % (bits, self.info["powerOfTwo"]))
        return bits

    def omega(self, bits):
        return self._omegas[self._check(bits)]

    def roots(self, bits):
        """Natural-order table; served by the device as the forward transform of the delta at 1."""
        self._check(bits)
        if bits not in self._cache:
            n = 1 << bits
            delta = [0] * n
            if n > 1:
                delta[1] = 1
                self._cache[bits] = FFT(self, self.field).direct(delta)
            else:
                self._cache[bits] = [1]
        return self._cache[bits]

    def brp(self, bits):
        r = self.roots(bits)
        return bitReversalPermutation(r) if bits else list(r)

    def inverse(self, bits):               # fft.ts:296-304
        r = self.roots(bits)
        return [r[0]] + r[1:][::-1]

    def clear(self):
        self._cache = {}


def rootsOfUnity(field, generator=None):
    return RootsOfUnity(field, generator)


class FFT:
    """fft.ts:518-577.  `values`: list of ints in [0, r) (or a uint8 array [N, 32], little-endian
    canonical residues, returned in kind)."""

    def __init__(self, roots, opts=None, engine=None):
        if not isinstance(roots, RootsOfUnity):
            raise TypeError("noble-gpu: FFT expects the RootsOfUnity returned by rootsOfUnity()")
        self.roots = roots
        self._engine = engine

    def _run(self, values, inverse, brpInput, brpOutput):
        raw = isinstance(values, np.ndarray)
        N = values.shape[0] if raw else len(values)
        if not isPowerOfTwo(N):
            raise ValueError("FFT: Polynomial size should be power of two")
        bits = log2(N)
        order = self.roots.field.ORDER
        if raw:
            data = np.ascontiguousarray(values, dtype=np.uint8).reshape(N, 32)
        else:
            for v in values:
                if not isinstance(v, int) or isinstance(v, bool) or not (0 <= v < order):
                    raise ValueError("invalid field element: outside of range 0..ORDER")
            data = _native.ints_to_le(values, 32)
        eng = self._engine or get_engine()
        out = eng.ntt(bits, data, self.roots.omega(bits), inverse=inverse, brp_input=brpInput, brp_output=brpOutput)
        return out if raw else _native.le_to_ints(out, 32)

    def direct(self, values, brpInput=False, brpOutput=False):
        return self._run(values, False, bool(brpInput), bool(brpOutput))

    def inverse(self, values, brpInput=False, brpOutput=False):
        return self._run(values, True, bool(brpInput), bool(brpOutput))
