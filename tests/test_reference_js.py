"""The Python oracle against the REFERENCE ITSELF, executed here: the reference's TypeScript sources, type-stripped by
oracle/ref_js/downlevel.py into oracle/_ref/refjs.bundle (git-ignored; built by __graft_entry__.build() when /root/reference is
present, shipped to the GPU box with the snapshot), run on this machine's Node 12 with node:crypto standing in for
@noble/hashes.  This pins the oracle a second time - beside the reference's golden vectors (test_oracle_golden.py) - on inputs
chosen here: random and edge scalars through Point.multiplyUnsafe / Point.multiply (weierstrass.ts:900-928), the scalar
patterns of benchmark/msm_timings.ts through pippenger (curve.ts:863-905) on G1 and G2, Edwards multiples of points with a
torsion component, and ed25519.verify (edwards.ts:942-989) on the zip215.json cases in both modes."""
import numpy as np
import pytest

from helpers import ORACLE_CURVE, load_golden, points_to_wire, scalars_to_wire, wire_to_affine
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2, ED25519, SECP256K1
from oracle import curve as OC
from oracle import refjs
from oracle.curves import makeRng
from oracle.edwards import eddsa_verify

pytestmark = pytest.mark.skipif(not refjs.available(), reason="oracle/_ref/refjs.bundle not built (needs /root/reference once, and node)")


def _affine_rows(curve, out):
    return [wire_to_affine(curve, out[i]) for i in range(out.shape[0])]


@pytest.mark.parametrize("curve", [SECP256K1, BLS12_381_G1, ED25519])
def test_oracle_multiply_equals_the_reference_run_here(curve):
    Pt = ORACLE_CURVE[curve]
    order = Pt.Fn.ORDER
    rng = makeRng(0x7E57 + curve)
    ks = [1, 2, order - 1, order - 2, (1 << 128) - 1, 1 << 128, (1 << 128) + 1, (1 << 255) % order, 2 ** 180 - 15820]
    ks += [rng.rndBelow(order - 1) + 1 for _ in range(16)]
    pts = [Pt.BASE.multiplyUnsafe(rng.rndBelow(order - 1) + 1) for _ in ks]
    pw, sw = points_to_wire(curve, pts), scalars_to_wire(ks)
    out_u, _ = refjs.multiply(curve, pw, sw, unsafe=True)
    out_m, _ = refjs.multiply(curve, pw, sw, unsafe=False)
    for i, (p, k) in enumerate(zip(pts, ks)):
        want = p.multiplyUnsafe(k).toAffine()
        assert wire_to_affine(curve, out_u[i]) == want, ("multiplyUnsafe", i)
        assert wire_to_affine(curve, out_m[i]) == p.multiply(k).toAffine() == want, ("multiply", i)
    # k = 0 and the identity through multiplyUnsafe (multiply rejects 0: weierstrass.ts:904)
    z = [0, 5, 0]
    zp = [pts[0], Pt.ZERO, Pt.ZERO]
    out_z, _ = refjs.multiply(curve, points_to_wire(curve, zp), scalars_to_wire(z), unsafe=True)
    assert all(wire_to_affine(curve, out_z[i]) == Pt.ZERO.toAffine() for i in range(3))


@pytest.mark.parametrize("curve", [BLS12_381_G1, BLS12_381_G2])
def test_oracle_pippenger_equals_the_reference_run_here(curve):
    Pt = ORACLE_CURVE[curve]
    N = Pt.Fn.ORDER
    bits = N.bit_length() - 1
    ones, onezero, one8zero = int("1" * bits, 2), int("10" * (bits // 2), 2), int("10000000" * (bits // 8), 2)
    G = Pt.BASE
    base5 = [G.multiply(i) for i in (3, 5, 7, 11, 13)]
    cases = [([G], [0]), ([G], [N - 1]), ([Pt.ZERO], [1]), (base5, [N - 1, N - 100, N - 200, N - 300, N - 400]),
             (base5, [ones] * 5), (base5, [onezero] * 5), (base5, [one8zero] * 5), (base5, [1] * 5), ([Pt.ZERO] * 5, [0] * 5)]
    rng = makeRng(0xB15 + curve)
    n = 40 if curve == BLS12_381_G1 else 20
    rp = [G.multiplyUnsafe(rng.rndBelow(N - 1) + 1) for _ in range(n)]
    rp[3] = Pt.ZERO
    rp[5] = rp[4].negate()
    rs = [0 if i % 7 == 2 else rng.rndBelow(N) for i in range(n)]
    rs[5] = rs[4]
    cases.append((rp, rs))
    for pts, sc in cases:
        got, _ = refjs.pippenger(curve, points_to_wire(curve, pts), scalars_to_wire(sc))
        assert wire_to_affine(curve, got) == OC.pippenger(Pt, pts, sc).toAffine(), (len(pts), sc[:2])


def test_oracle_ed25519_verify_equals_the_reference_run_here():
    Ed = ORACLE_CURVE[ED25519]
    z = load_golden("ed25519_zip215.json")
    sigs = [bytes.fromhex(c["sig_bytes"]) for c in z]
    pks = [bytes.fromhex(c["vk_bytes"]) for c in z]
    msgs = [b"Zcash"] * len(z)
    for zip215 in (True, False):
        got, _ = refjs.ed25519_verify(sigs, msgs, pks, zip215=zip215)
        want = [eddsa_verify(Ed, s, m, k, zip215=zip215) for s, m, k in zip(sigs, msgs, pks)]
        assert list(got) == want, "zip215=%s" % zip215
        if zip215:
            assert all(got)       # every case of the file is valid under ZIP-215 (test/ed25519.test.ts:397-410)


@pytest.mark.parametrize("curve", [BLS12_381_G1, BLS12_381_G2])
def test_oracle_hash_to_curve_equals_the_reference_run_here(curve):
    """bls12_381.G1 / G2 .hashToCurve and .encodeToCurve (hash-to-curve.ts:487-500: expand_message_xmd, hash_to_field, SSWU + isogeny,
    clear cofactor) on messages and domain-separation tags chosen here."""
    from oracle import h2c as OH
    H = OH.G1_hasher if curve == BLS12_381_G1 else OH.G2_hasher
    msgs = [b"", b"abc", b"abcdef0123456789", b"\x00" * 32, bytes(range(64)), b"q128_" + b"q" * 59]
    for dst in (H.DST, b"QUUX-V01-CS02-with-BLS12381G%d_XMD:SHA-256_SSWU_RO_" % (1 if curve == BLS12_381_G1 else 2), b"x"):
        for encode in (False, True):
            got, _ = refjs.hash_to_curve(curve, msgs, dst, encode=encode)
            for i, m in enumerate(msgs):
                want = (H.encodeToCurve(m, dst) if encode else H.hashToCurve(m, dst)).toAffine()
                assert wire_to_affine(curve, got[i]) == want, (dst, encode, i)


def test_oracle_fft_equals_the_reference_run_here():
    """FFT(rootsOfUnity(Fr), Fr) of the bls12-381 scalar field (fft.ts:518-577): direct and inverse, every combination of
    bit-reversed input / output, sizes 1 .. 256."""
    from oracle.curves import BLS_R
    from oracle.fft import FFT, RootsOfUnity
    from oracle.field import Field
    Fr = Field(BLS_R)
    f = FFT(RootsOfUnity(Fr), Fr)
    rng = makeRng(0xFF7)
    for bits in (0, 1, 3, 8):
        vals = [rng.rndBelow(BLS_R) for _ in range(1 << bits)]
        if bits:
            vals[0], vals[-1] = 0, BLS_R - 1
        for inverse in (False, True):
            for brp_in in (False, True):
                for brp_out in (False, True):
                    got, _ = refjs.fft(vals, inverse, brp_in, brp_out)
                    want = (f.inverse if inverse else f.direct)(vals, brp_in, brp_out)
                    assert got == want, (bits, inverse, brp_in, brp_out)


@pytest.mark.parametrize("curve", [SECP256K1, ED25519, BLS12_381_G1, BLS12_381_G2])
def test_oracle_point_encodings_equal_the_reference_run_here(curve):
    """Point.toBytes (compressed SEC1 / RFC 8032 / zkcrypto flags, weierstrass.ts:551-566, edwards.ts:620-628, bls12-381.ts G1 / G2
    encoders) of random points, the base point and points whose y decides the sign / sort bit both ways; the reference also decodes
    each encoding back (Point.fromBytes) inside the run."""
    from oracle import weierstrass as OW
    Pt = ORACLE_CURVE[curve]
    order = Pt.Fn.ORDER
    rng = makeRng(0xC0DEC + curve)
    pts = [Pt.BASE, Pt.BASE.negate()] + [Pt.BASE.multiplyUnsafe(rng.rndBelow(order - 1) + 1) for _ in range(12)]
    pts += [p.negate() for p in pts[2:6]]
    got, info = refjs.codec(curve, points_to_wire(curve, pts))
    for i, p in enumerate(pts):
        if curve == SECP256K1:
            want = OW.sec1_encode(p, True)
        elif curve == ED25519:
            want = p.toBytes()
        elif curve == BLS12_381_G1:
            want = OW.bls_g1_encode_compressed(p)
        else:
            want = OW.bls_g2_encode_compressed(p)
        assert bytes(got[i]) == bytes(want), i
