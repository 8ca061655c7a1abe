"""Pins the oracle's C restatement (oracle/c) to the Python oracle, which is itself pinned by
the reference's fixtures (tests/test_oracle_golden.py)."""
import time

import pytest

from helpers import load_golden, points_to_wire, scalars_to_wire, wire_to_affine
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2, SECP256K1
from oracle import cport
from oracle import curve as C
from oracle.curves import BLS_R, BlsG1, BlsG2, SECP256K1_N, Secp256k1, makeRng


def test_c_secp256k1_multiply_unsafe_vs_python_and_golden():
    n = SECP256K1_N
    rng = makeRng(0xC0DE)
    lam = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
    ks = [0, 1, 2, 3, n - 1, n - 2, lam, n - lam, 1 << 128, (1 << 128) - 1] + [rng.rndBelow(n) for _ in range(120)]
    pts = [Secp256k1.BASE.multiplyUnsafe(rng.rndBelow(n - 1) + 1) for _ in ks]
    pts[5] = Secp256k1.ZERO
    for t in load_golden("secp256k1_endomorphism.json"):
        pts.append(Secp256k1.fromAffine((int(t["ax"]), int(t["ay"]))))
        ks.append(int(t["scalar"]))
    out, inf = cport.multiply_unsafe("secp256k1", points_to_wire(SECP256K1, pts), scalars_to_wire(ks))
    for i, (p, k) in enumerate(zip(pts, ks)):
        exp = p.multiplyUnsafe(k).toAffine()
        assert wire_to_affine(SECP256K1, out[i]) == exp
        assert bool(inf[i]) == (exp == (0, 0))
    for t, i in zip(load_golden("secp256k1_endomorphism.json"), range(len(ks) - 3, len(ks))):
        assert wire_to_affine(SECP256K1, out[i]) == (int(t["cx"]), int(t["cy"]))


def test_c_g1_multiply_unsafe_vs_python():
    rng = makeRng(0xC1)
    ks = [0, 1, 2, BLS_R - 1] + [rng.rndBelow(BLS_R) for _ in range(30)]
    pts = [BlsG1.BASE.multiplyUnsafe(rng.rndBelow(BLS_R - 1) + 1) for _ in ks]
    out, inf = cport.multiply_unsafe("bls12_381_g1", points_to_wire(BLS12_381_G1, pts), scalars_to_wire(ks))
    for i, (p, k) in enumerate(zip(pts, ks)):
        assert wire_to_affine(BLS12_381_G1, out[i]) == p.multiplyUnsafe(k).toAffine()


def test_c_secp256k1_multiply_ct_shape_vs_python():
    """Point.multiply through the fixed-window constant-time path (curve.ts:707-729), unblinded and with the 128-bit
    blind n = k + r * ORDER (:663-690): the value never depends on the blind (test/point.test.ts:647-651)."""
    import numpy as np
    n = SECP256K1_N
    rng = makeRng(0xC7)
    ks = [1, 2, 31, 32, n - 1, n - 2, (1 << 180) - 15820, 1 << 255] + [rng.rndBelow(n - 1) + 1 for _ in range(40)]
    pts = [Secp256k1.BASE.multiplyUnsafe(rng.rndBelow(n - 1) + 1) for _ in ks]
    exp = [p.multiply(k).toAffine() for p, k in zip(pts, ks)]
    pw, sw = points_to_wire(SECP256K1, pts), scalars_to_wire(ks)
    out, inf = cport.multiply(pw, sw)
    blinds = np.array([[(rng.rnd64() >> 8) & 0xFF for _ in range(16)] for _ in ks], dtype=np.uint8)
    blinds[0] = 0          # rngMin-like (test/point.test.ts:586): the forced top bits make it 2^127
    blinds[1] = 0xFF       # rngMax-like
    outb, infb = cport.multiply(pw, sw, blinds)
    for i in range(len(ks)):
        assert wire_to_affine(SECP256K1, out[i]) == exp[i] and not inf[i]
        assert wire_to_affine(SECP256K1, outb[i]) == exp[i] and not infb[i]


@pytest.mark.parametrize("name,curve,Pt,n", [("bls12_381_g1", BLS12_381_G1, BlsG1, 200),
                                             ("secp256k1", SECP256K1, Secp256k1, 150),
                                             ("bls12_381_g2", BLS12_381_G2, BlsG2, 40)])
def test_c_pippenger_vs_python(name, curve, Pt, n):
    order = Pt.Fn.ORDER
    rng = makeRng(0xC2 + curve)
    pts = [Pt.BASE.multiplyUnsafe(rng.rndBelow(order - 1) + 1) for _ in range(n)]
    sc = [0 if i % 17 == 0 else rng.rndBelow(order) for i in range(n)]
    pts[3] = Pt.ZERO
    pts[4] = pts[5].negate()
    sc[4] = sc[5]
    out, inf = cport.pippenger(name, points_to_wire(curve, pts), scalars_to_wire(sc))
    exp = C.pippenger(Pt, pts, sc).toAffine()
    assert wire_to_affine(curve, out) == exp and inf == (exp == (0, 0))
    out, inf = cport.pippenger(name, points_to_wire(curve, []), scalars_to_wire([]))
    assert inf


def test_c_ed25519_verify_vs_python_both_modes():
    import numpy as np
    from oracle.curves import Ed25519
    from oracle.edwards import eddsa_hash_k, eddsa_verify
    from test_host_logic import _ed_cases
    cases = _ed_cases()
    sig = np.array([np.frombuffer(c[0], np.uint8) for c in cases])
    pk = np.array([np.frombuffer(c[2], np.uint8) for c in cases])
    ks = np.array([np.frombuffer(eddsa_hash_k(Ed25519.Fn, c[0][:32], c[2], c[1]).to_bytes(32, "little"), np.uint8)
                   for c in cases])
    for zip215 in (True, False):
        got = cport.ed25519_verify_batch(sig, pk, ks, zip215)
        exp = [eddsa_verify(Ed25519, c[0], c[1], c[2], zip215=zip215) for c in cases]
        assert list(got) == exp


def test_c_fft_matches_python_oracle():
    """oracle/c FFT over Fr == the Python restatement (itself pinned to test/fft.test.ts) for every
    (inverse, brpInput, brpOutput) combination."""
    import numpy as np
    from oracle import cport
    from oracle.curves import Fr_bls, makeRng
    from oracle.fft import FFT, RootsOfUnity
    roots = RootsOfUnity(Fr_bls, 7)
    f = FFT(roots, Fr_bls)
    rng = makeRng(0xCFF7)
    for bits in (0, 1, 3, 6):
        x = [rng.rndBelow(Fr_bls.ORDER) for _ in range(1 << bits)]
        data = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in x), dtype=np.uint8).reshape(-1, 32)
        for flags in range(8):
            inv, bi, bo = bool(flags & 1), bool(flags & 2), bool(flags & 4)
            got = cport.fft_fr(bits, data, roots.omega(bits), inv, bi, bo)
            exp = (f.inverse if inv else f.direct)(x, bi, bo)
            assert [int.from_bytes(r.tobytes(), "little") for r in got] == exp, (bits, flags)
