"""The shim's prime-field object (noble-curves_amd/field.py) against the reference's own checks for it
(test/modular.test.ts) and against the oracle's restatement on the fields of the path.  Host logic only."""
import random

import pytest

from noble_curves_amd.field import Field, FpInvertBatch, FpIsSquare, FpLegendre, FpPow, invert, tonelliShanks
from oracle import field as ofield
from oracle.curves import BLS_P, BLS_R, ED25519_L, ED25519_P, SECP256K1_N, SECP256K1_P

PATH_FIELDS = [SECP256K1_P, SECP256K1_N, ED25519_P, ED25519_L, BLS_P, BLS_R]


def test_small_primes_exhaustive_sqrt_legendre():
    """test/modular.test.ts:509-531: all four sqrt dispatch classes against brute force."""
    smalls = [3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 73, 89, 97, 113, 193, 233, 241, 257, 337, 353, 1039, 7681]
    for P in smalls:
        F = Field(P)
        qrs = {x * x % P for x in range(P)}
        for n in range(P):
            exp = 0 if n == 0 else (1 if n in qrs else -1)
            assert FpLegendre(F, n) == exp and FpIsSquare(F, n) == (exp >= 0)
            if exp >= 0:
                r = F.sqrt(n)
                assert r * r % P == n
            else:
                with pytest.raises(ValueError, match="Cannot find square root"):
                    F.sqrt(n)


def test_sqrt_known_answer_and_high_two_adicity():
    """test/modular.test.ts:490-497 (Tonelli-Shanks root of -1 mod the P-224 prime, 'verified against sage'),
    :534-548 (S = 192), :499-507 (non-prime modulus refused)."""
    F = Field(2**224 - 2**96 + 1)
    r = F.sqrt(-1)
    assert r == 23621584063597419797792593680131996961517196803742576047493035507225
    assert F.neg(r) == 3338362603553219996874421406887633712040719456283732096017030791656
    assert F.sqr(r) == F.neg(F.ONE)
    stark = (1 << 251) + 17 * (1 << 192) + 1
    Fs = Field(stark)
    rng = random.Random(0x57A4C)
    for _ in range(6):
        x = rng.randrange(stark)
        n = x * x % stark
        assert Fs.sqrt(n) ** 2 % stark == n
    z = 2
    while FpLegendre(Fs, z) != -1:
        z += 1
    with pytest.raises(ValueError):
        Fs.sqrt(z)
    with pytest.raises(ValueError):
        tonelliShanks(21888242871839275222246405745257275088614511777268538073601725287587578984328)


def test_path_fields_match_oracle():
    """Every op on the six fields of the hot path equals the oracle's Field (pinned to the reference through
    the point-level goldens): random and edge operands."""
    rng = random.Random(5)
    for P in PATH_FIELDS:
        F, O = Field(P), ofield.Field(P)
        assert (F.ORDER, F.BITS, F.BYTES, F.isLE, F.ZERO, F.ONE) == (P, P.bit_length(), (P.bit_length() + 7) // 8, False, 0, 1)
        vals = [0, 1, 2, P - 1, P - 2] + [rng.randrange(P) for _ in range(20)]
        for a in vals:
            assert F.create(a + P) == O.create(a + P) == a and F.neg(a) == O.neg(a) and F.sqr(a) == O.sqr(a)
            assert F.isValid(a) and not F.isValid(P) and not F.isValid(-1) and F.is0(a) == (a == 0)
            assert F.isValidNot0(a) == (a != 0) and F.isOdd(a) == bool(a & 1)
            assert F.toBytes(a) == O.toBytes(a) and F.fromBytes(F.toBytes(a)) == a
            for b in vals[:8]:
                assert F.add(a, b) == O.add(a, b) and F.sub(a, b) == O.sub(a, b) and F.mul(a, b) == O.mul(a, b)
                if b:
                    assert F.div(a, b) == O.div(a, b)
            if a:
                assert F.inv(a) == O.inv(a) == invert(a, P) and F.mul(a, F.inv(a)) == 1
            e = rng.randrange(P)
            assert F.pow(a, e) == O.pow(a, e) == FpPow(F, a, e)
            if P % 4 == 3:
                sq = F.sqr(a)
                assert F.sqrt(sq) == O.sqrt(sq)
            else:
                assert F.sqr(F.sqrt(F.sqr(a))) == F.sqr(a)
        assert F.invertBatch(vals) == O.invertBatch(vals, True)
        assert FpInvertBatch(F, [0, 3, 0, 5]) == [None, F.inv(3), None, F.inv(5)]
        assert F.addN(P, P) == 2 * P and F.subN(1, 2) == -1 and F.mulN(P, 2) == 2 * P and F.sqrN(P) == P * P
        assert F.cmov(3, 4, True) == 4 and F.cmov(3, 4, False) == 3 and F.eql(a, a)


def test_errors_and_byte_options():
    """Messages of modular.ts:159-182, :666-675, :897-931, :996-1022, :1032-1037."""
    with pytest.raises(ValueError, match="invalid field: expected ORDER > 1"):
        Field(1)
    F = Field(17)
    with pytest.raises(TypeError, match="invalid field element: expected bigint, got str"):
        F.isValid("3")
    with pytest.raises(ValueError, match="invert: expected non-zero number"):
        F.inv(0)
    with pytest.raises(ValueError, match="invert: does not exist"):
        invert(6, 9)
    with pytest.raises(ValueError, match="invert: expected modulus > 1"):
        invert(3, 1)
    with pytest.raises(ValueError, match="invalid exponent, negatives unsupported"):
        F.pow(3, -1)
    with pytest.raises(TypeError, match="invalid exponent: expected bigint"):
        FpPow(F, 3, 2.0)
    with pytest.raises(TypeError, match="cmov"):
        F.cmov(1, 2, 1)
    with pytest.raises(ValueError, match="Fp.sqrt: expected odd modulus"):
        Field(16).sqrt(4)
    with pytest.raises(ValueError, match=r"Field.fromBytes: expected 1 bytes, got 2"):
        F.fromBytes(b"\x00\x01")
    with pytest.raises(ValueError, match="outside of range 0..ORDER"):
        F.fromBytes(b"\x11")
    assert F.fromBytes(b"\x11", True) == 17 and Field(17, modFromBytes=True).fromBytes(b"\x13") == 2
    L = Field(2**255 - 19, isLE=True)
    assert L.toBytes(1) == b"\x01" + bytes(31) and L.fromBytes(b"\x02" + bytes(31)) == 2
    A = Field(SECP256K1_N, allowedLengths=[16, 32])
    assert A.fromBytes(b"\x01" * 16) == int.from_bytes(b"\x01" * 16, "big")
    with pytest.raises(ValueError, match=r"Field.fromBytes: expected 16,32 bytes, got 17"):
        A.fromBytes(b"\x01" * 17)
    assert Field(17, sqrt=lambda n: 99).sqrt(4) == 99 and Field(17, BITS=8).BYTES == 1
    assert FpPow(F, 3, 0) == 1 and FpPow(F, 3, 1) == 3 and F.pow(3, 16) == 1


def test_point_class_fields_are_full_fields():
    from noble_curves_amd import curve as G
    K1 = G.secp256k1_Point
    assert isinstance(K1.Fp, Field) and K1.Fp.sqrt(4) in (2, SECP256K1_P - 2) and K1.Fn.inv(2) * 2 % SECP256K1_N == 1
    assert G.bls12_381_G2_Point.Fp.isValid((1, 2)) and not G.bls12_381_G2_Point.Fp.isValid(3)
