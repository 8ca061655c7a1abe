"""GPU parity: batch variable-base scalar multiplication (ncg_mul_var_batch) against the CPU
oracle and the reference's golden vectors.  Bit-exact on canonical affine coordinates."""
import numpy as np
import pytest

from noble_curves_amd import get_engine
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2, SECP256K1
from oracle import curve as C
from oracle.curves import BLS_R, BlsG1, BlsG2, SECP256K1_N, Secp256k1, makeRng
from oracle.weierstrass import bls_g1_decode_uncompressed, bls_g2_decode_uncompressed, sec1_decode

from helpers import (ORACLE_CURVE, load_golden, points_to_wire, scalars_to_wire, wire_to_affine)

pytestmark = pytest.mark.gpu


def run_and_check(curve, pts, scalars, expected_affine):
    eng = get_engine()
    out, inf = eng.mul_var_batch(curve, points_to_wire(curve, pts), scalars_to_wire(scalars))
    Pt = ORACLE_CURVE[curve]
    zero_aff = Pt.ZERO.toAffine()
    for i, exp in enumerate(expected_affine):
        got = wire_to_affine(curve, out[i])
        assert got == exp, "item %d: k=%x" % (i, scalars[i])
        assert bool(inf[i]) == (exp == zero_aff)


def test_secp256k1_golden_vectors():
    """test/secp256k1.test.ts:59-71,79-131 and test/nist.test.ts:550-559 through the GPU path."""
    pts, ks, exp = [], [], []
    for k, x, y in load_golden("secp256k1_privates2.json"):
        pts.append(Secp256k1.BASE)
        ks.append(int(k))
        exp.append((int(x, 16), int(y, 16)))
    for t in load_golden("secp256k1_endomorphism.json"):
        pts.append(Secp256k1.fromAffine((int(t["ax"]), int(t["ay"]))))
        ks.append(int(t["scalar"]))
        exp.append((int(t["cx"]), int(t["cy"])))
    v = load_golden("secp256k1_points.json")
    for t in v["valid"]["pointMultiply"]:
        if t["expected"]:
            pts.append(sec1_decode(Secp256k1, bytes.fromhex(t["P"])))
            ks.append(int(t["d"], 16))
            exp.append(sec1_decode(Secp256k1, bytes.fromhex(t["expected"])).toAffine())
    for t in v["valid"]["pointFromScalar"]:
        pts.append(Secp256k1.BASE)
        ks.append(int(t["d"], 16))
        exp.append(sec1_decode(Secp256k1, bytes.fromhex(t["expected"])).toAffine())
    run_and_check(SECP256K1, pts, ks, exp)


def test_secp256k1_random_and_edges_vs_oracle():
    n = SECP256K1_N
    rng = makeRng(0x6E6F626C6502)
    a, b = rng.rndBelow(n - 1) + 1, rng.rndBelow(n - 1) + 1
    lam = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
    edge = [0, 1, 2, 3, n - 1, n - 2, n - 3, 1 << 128, (1 << 128) - 1, (1 << 128) + 1, (1 << 255), lam, lam + 1,
            lam - 1, n - lam, (n + 1) // 2, n // 2, (1 << 64), 0xFFFFFFFF, 1 << 32, 15, 16, 17, 255, 256]
    pts, ks = [], []
    for i, k in enumerate(edge):
        pts.append(Secp256k1.BASE.multiplyUnsafe((a + i * b) % n))
        ks.append(k)
    pts.append(Secp256k1.ZERO)      # infinity input
    ks.append(12345)
    pts.append(Secp256k1.ZERO)
    ks.append(0)
    for i in range(700):
        pts.append(Secp256k1.BASE.multiplyUnsafe((a + (100 + i) * b) % n))
        ks.append(rng.rndBelow(n))
    exp = [p.multiplyUnsafe(k).toAffine() for p, k in zip(pts, ks)]
    run_and_check(SECP256K1, pts, ks, exp)


def test_secp256k1_non_multiple_of_wave_and_empty():
    eng = get_engine()
    out, inf = eng.mul_var_batch(SECP256K1, np.zeros((0, 64), np.uint8), np.zeros((0, 32), np.uint8))
    assert out.shape == (0, 64) and inf.shape == (0,)
    for n in (1, 63, 65):
        pts = [Secp256k1.BASE.multiplyUnsafe(i + 1) for i in range(n)]
        ks = [(i * 0x9E3779B97F4A7C15 + 7) % SECP256K1_N for i in range(n)]
        run_and_check(SECP256K1, pts, ks, [p.multiplyUnsafe(k).toAffine() for p, k in zip(pts, ks)])


@pytest.mark.parametrize("curve,decode,key", [(BLS12_381_G1, bls_g1_decode_uncompressed, "G1_Uncompressed"),
                                              (BLS12_381_G2, bls_g2_decode_uncompressed, "G2_Uncompressed")])
def test_bls_multiples_golden(curve, decode, key):
    """test/bls12-381.test.ts:1463-1535: BASE.multiplyUnsafe(i) == zkcrypto i*G, i = 0..255."""
    Pt = ORACLE_CURVE[curve]
    rows = load_golden("bls12_381_multiples.json")[key]
    pts = [Pt.BASE] * len(rows)
    ks = list(range(len(rows)))
    exp = [decode(Pt, bytes.fromhex(t)).toAffine() for t in rows]
    run_and_check(curve, pts, ks, exp)


@pytest.mark.parametrize("curve,count", [(BLS12_381_G1, 150), (BLS12_381_G2, 70)])
def test_bls_random_vs_oracle(curve, count):
    Pt = ORACLE_CURVE[curve]
    rng = makeRng(0xB15 + curve)
    pts, ks = [], []
    for k in (0, 1, 2, BLS_R - 1, BLS_R - 2, 1 << 254, (1 << 255) - 19 if (1 << 255) - 19 < BLS_R else 5):
        pts.append(Pt.BASE.multiplyUnsafe(rng.rndBelow(BLS_R - 1) + 1))
        ks.append(k)
    pts.append(Pt.ZERO)
    ks.append(99)
    for _ in range(count):
        pts.append(Pt.BASE.multiplyUnsafe(rng.rndBelow(BLS_R - 1) + 1))
        ks.append(rng.rndBelow(BLS_R))
    exp = [p.multiplyUnsafe(k).toAffine() for p, k in zip(pts, ks)]
    run_and_check(curve, pts, ks, exp)


@pytest.mark.parametrize("curve", [SECP256K1, BLS12_381_G1, BLS12_381_G2])
def test_fixed_base_batch(curve):
    """BASE.multiply(k) through the fixed-base table (curve.ts:588-606): golden k*G vectors,
    window-boundary and carry-chain scalars, random scalars."""
    Pt = ORACLE_CURVE[curve]
    n = Pt.Fn.ORDER
    rng = makeRng(0xBA5E + curve)
    ks = [0, 1, 2, 3, 255, 256, 257, (1 << 248) - 1, 1 << 248, (1 << 255) % n, n - 1, n - 2, n // 2, n // 2 + 1,
          int("ff" * 31, 16), int("80" * 32, 16) % n, int("7f" * 32, 16) % n, int("01" * 32, 16) % n]
    ks += [rng.rndBelow(n) for _ in range(60 if curve != BLS12_381_G2 else 25)]
    out, inf = get_engine().mul_base_batch(curve, scalars_to_wire(ks))
    for i, k in enumerate(ks):
        exp = Pt.BASE.multiplyUnsafe(k).toAffine()
        assert wire_to_affine(curve, out[i]) == exp, hex(k)
        assert bool(inf[i]) == (k == 0)
    if curve == SECP256K1:
        rows = load_golden("secp256k1_privates2.json")
        out, _ = get_engine().mul_base_batch(curve, scalars_to_wire([int(r[0]) for r in rows]))
        for i, (_, x, y) in enumerate(rows):
            assert wire_to_affine(curve, out[i]) == (int(x, 16), int(y, 16))


def test_ed25519_mul_var_and_msm_gpu():
    """ed25519 multiplyUnsafe batch (incl. torsioned points) and pippenger on the GPU."""
    from noble_curves_amd._native import ED25519
    from oracle.curves import ED25519_L, Ed25519
    from test_host_logic import _ed_points_and_scalars
    pts, ks = _ed_points_and_scalars()
    eng = get_engine()
    out, inf = eng.mul_var_batch(ED25519, points_to_wire(ED25519, pts), scalars_to_wire(ks))
    for i, (p, k) in enumerate(zip(pts, ks)):
        exp = C.naiveMul(Ed25519, p, k).toAffine()
        assert wire_to_affine(ED25519, out[i]) == exp and bool(inf[i]) == (exp == (0, 1))
    # MSM: progression identity + degenerate inputs
    rng = makeRng(0xED5)
    a, b = rng.rndBelow(ED25519_L - 1) + 1, rng.rndBelow(ED25519_L - 1) + 1
    n = 700
    kk = [(a + i * b) % ED25519_L for i in range(n)]
    P = [Ed25519.BASE.multiplyUnsafe(k) for k in kk]
    sc = [0 if i % 17 == 0 else rng.rndBelow(ED25519_L) for i in range(n)]
    exp = Ed25519.BASE.multiplyUnsafe(sum(k * s for k, s in zip(kk, sc)) % ED25519_L).toAffine()
    got, ginf = eng.msm(ED25519, points_to_wire(ED25519, P), scalars_to_wire(sc))
    assert wire_to_affine(ED25519, got) == exp and not ginf
    got, ginf = eng.msm(ED25519, points_to_wire(ED25519, [P[1], P[1].negate(), Ed25519.ZERO]), scalars_to_wire([5, 5, 9]))
    assert wire_to_affine(ED25519, got) == (0, 1) and ginf
    got, ginf = eng.msm(ED25519, points_to_wire(ED25519, pts[:12]), scalars_to_wire(ks[:12]))   # torsion inside an MSM
    acc = Ed25519.ZERO
    for p, k in zip(pts[:12], ks[:12]):
        acc = acc.add(C.naiveMul(Ed25519, p, k))
    assert wire_to_affine(ED25519, got) == acc.toAffine()
    got, ginf = eng.msm(ED25519, points_to_wire(ED25519, []), scalars_to_wire([]))
    assert wire_to_affine(ED25519, got) == (0, 1) and ginf


def test_normalize_projective_batch():
    """normalizeZ / FpInvertBatch (curve.ts:311-326, modular.ts:728-760): non-normalised
    projective points (results of add/double, as in test/slow-curves.test.ts:220), ZERO inside
    the batch (zero Z must not poison the shared inversion), odd batch sizes."""
    from noble_curves_amd import curve as G
    from oracle.curves import Ed25519
    for Pt, M in ((Secp256k1, G.secp256k1_Point), (BlsG1, G.bls12_381_G1_Point), (BlsG2, G.bls12_381_G2_Point),
                  (Ed25519, G.ed25519_Point)):
        pts = []
        acc = Pt.BASE
        for i in range(21):
            acc = acc.double().add(Pt.BASE) if i % 2 else acc.add(acc.double())
            pts.append(acc)
        if Pt is not Ed25519:
            pts[3] = Pt.ZERO
            pts[8] = Pt.BASE.add(Pt.BASE.negate())      # (X, Y, 0) with non-trivial X, Y
        triples = [(p.X, p.Y, p.Z) for p in pts]
        got = G.normalizeProjective(M, triples)
        for p, g in zip(pts, got):
            assert g.toAffine() == p.toAffine()


def test_from_bytes_batch_gpu():
    """SURVEY 8(f) row 1 on the GPU: SEC1 / G1-compressed / ed25519 decoding with the
    reference's accept/reject behaviour (isPoint vectors, zkcrypto vectors, flag and range
    violations, points outside the G1 subgroup)."""
    from noble_curves_amd import curve as G
    from oracle.curves import Ed25519
    from test_host_logic import decode_cases_g1, decode_cases_g2, decode_cases_secp
    cs = decode_cases_secp()
    got = G.fromBytesBatch(G.secp256k1_Point, [c[0] for c in cs])
    for (e, exp), g in zip(cs, got):
        assert (g.toAffine() if g is not None else None) == exp, e.hex()
    cg = decode_cases_g1()
    got = G.fromBytesBatch(G.bls12_381_G1_Point, [c[0] for c in cg])
    for (e, exp, is0), g in zip(cg, got):
        assert (g.toAffine() if g is not None else None) == exp and (g is None or g.is0() == is0), e.hex()
    cg2 = decode_cases_g2()
    cg2 += [(bytes.fromhex(r), None, None) for r in load_golden("bls12_381_g2_compressed.json")[12:256]]
    unc = load_golden("bls12_381_multiples.json")["G2_Uncompressed"]
    got = G.fromBytesBatch(G.bls12_381_G2_Point, [c[0] for c in cg2])
    n0 = len(cg2) - 244
    for i, ((e, exp, is0), g) in enumerate(zip(cg2, got)):
        if i >= n0:  # zkcrypto vector (12 + i - n0) * G2: compare with its uncompressed twin
            from oracle.curves import BlsG2
            from oracle.weierstrass import bls_g2_decode_uncompressed
            exp = bls_g2_decode_uncompressed(BlsG2, bytes.fromhex(unc[12 + i - n0])).toAffine()
            assert g is not None and g.toAffine() == exp
        else:
            assert (g.toAffine() if g is not None else None) == exp and (g is None or g.is0() == is0), e.hex()
    encs = [bytes.fromhex(v["vk_bytes"]) for v in load_golden("ed25519_zip215.json")]
    for zip215 in (True, False):
        got = G.fromBytesBatch(G.ed25519_Point, encs, zip215=zip215)
        for e, g in zip(encs, got):
            try:
                exp = Ed25519.fromBytes(e, zip215).toAffine()
            except ValueError:
                exp = None
            assert (g.toAffine() if g is not None else None) == exp, (e.hex(), zip215)


@pytest.mark.gpu
def test_to_bytes_batch_gpu_round_trip():
    """toBytesBatch == the oracle encoders, and fromBytesBatch(toBytesBatch(P)) == P (all four curves)."""
    from noble_curves_amd import curve as G
    from oracle.curves import BlsG1, BlsG2, Ed25519, Secp256k1
    from oracle.weierstrass import bls_g1_encode_compressed, bls_g2_encode_compressed, sec1_encode
    rng = makeRng(0x70B7)
    for Pt, O, encf, n in ((G.secp256k1_Point, Secp256k1, sec1_encode, 64),
                           (G.bls12_381_G1_Point, BlsG1, bls_g1_encode_compressed, 48),
                           (G.bls12_381_G2_Point, BlsG2, bls_g2_encode_compressed, 24),
                           (G.ed25519_Point, Ed25519, lambda p: p.toBytes(), 64)):
        ks = [1, 2] + [rng.rndBelow(Pt.Fn.ORDER - 1) + 1 for _ in range(n - 2)]
        pts = G.multiplyBaseBatch(Pt, ks)
        if Pt is not G.secp256k1_Point:
            pts.append(Pt.ZERO)
        enc = G.toBytesBatch(Pt, pts)
        for p, e in zip(pts, enc):
            x, y = p.toAffine()
            op = O.ZERO if p.is0() else O.fromAffine((x, y))
            assert e == encf(op)
        back = G.fromBytesBatch(Pt, enc)
        assert all(b is not None and b.equals(p) for b, p in zip(back, pts))
    with pytest.raises(ValueError, match="bad point: ZERO"):
        G.toBytesBatch(G.secp256k1_Point, [G.secp256k1_Point.ZERO])


@pytest.mark.gpu
def test_ed25519_fixed_base_table_gpu():
    """BASE.multiply(k) for ed25519 through the dedicated window table (curve.ts:588-606): edge scalars,
    every window digit sign, k = 0 / 1 / L - 1, equality with the variable-base kernel on BASE."""
    from noble_curves_amd import curve as G
    from noble_curves_amd._native import ED25519
    from oracle.curves import ED25519_L, Ed25519
    rng = makeRng(0xEDB45E)
    ks = [0, 1, 2, 3, 127, 128, 129, 255, 256, 257, (1 << 252) + 1, ED25519_L - 1, ED25519_L - 2, 1 << 200,
          (1 << 253) - 1 if (1 << 253) - 1 < ED25519_L else 12345]
    ks += [rng.rndBelow(ED25519_L) for _ in range(200)]
    eng = get_engine()
    sc = scalars_to_wire(ks)
    out, inf = eng.mul_base_batch(ED25519, sc)
    for i, k in enumerate(ks[:60]):
        exp = Ed25519.BASE.multiplyUnsafe(k)
        assert wire_to_affine(ED25519, out[i]) == exp.toAffine() and bool(inf[i]) == exp.is0(), k
    base = np.tile(points_to_wire(ED25519, [Ed25519.BASE]), (len(ks), 1))
    out2, inf2 = eng.mul_var_batch(ED25519, base, sc)
    assert (out == out2).all() and (inf == inf2).all()
    got = G.multiplyBaseBatch(G.ed25519_Point, [k for k in ks if k])
    assert all(p.toAffine() == Ed25519.BASE.multiplyUnsafe(k).toAffine() for p, k in zip(got[:20], [k for k in ks if k][:20]))


@pytest.mark.gpu
def test_torsion_and_cofactor_batches_gpu():
    """isTorsionFree / clearCofactor in batch vs the oracle: ed25519 points with and without a torsion
    component (test/ed25519.test.ts:355-390 style inputs), bls12-381 G1 points inside and outside the
    prime-order subgroup."""
    from noble_curves_amd import curve as G
    from oracle.curves import BLS_P, BlsG1, Ed25519
    from oracle.weierstrass import bls_g1_is_torsion_free
    rng = makeRng(0x7025)
    # ed25519: prime-order points, small-order points, and sums of both
    torsion = []
    for enc in ("0100000000000000000000000000000000000000000000000000000000000000",
                "ecffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f",
                "0000000000000000000000000000000000000000000000000000000000000080",
                "26e8958fc2b227b045c3f489f2ef98f0d5dfac05d3c63339b13802886d53fc05"):
        torsion.append(Ed25519.fromBytes(bytes.fromhex(enc), True))
    prime = [Ed25519.BASE.multiplyUnsafe(rng.rndBelow(1 << 200) + 1) for _ in range(6)]
    mixed = [prime[i].add(torsion[i % len(torsion)]) for i in range(6)]
    opts = prime + torsion + mixed
    gpts = [G.ed25519_Point.fromAffine(p.toAffine()) for p in opts]
    assert G.isTorsionFreeBatch(G.ed25519_Point, gpts) == [p.isTorsionFree() for p in opts]
    cleared = G.clearCofactorBatch(G.ed25519_Point, gpts)
    assert [p.toAffine() for p in cleared] == [p.clearCofactor().toAffine() for p in opts]
    # bls12-381 G1: subgroup points and arbitrary curve points (x from the counter, y = sqrt(x^3 + 4))
    F = BlsG1.Fp
    outside = []
    x = 5
    while len(outside) < 5:
        try:
            y = F.sqrt((pow(x, 3, BLS_P) + 4) % BLS_P)
            outside.append(BlsG1.fromAffine((x, y)))
        except ValueError:
            pass
        x += 1
    inside = [BlsG1.BASE.multiplyUnsafe(rng.rndBelow(1 << 128) + 1) for _ in range(5)]
    opts = inside + outside
    gpts = [G.bls12_381_G1_Point.fromAffine(p.toAffine()) for p in opts]
    exp = [bls_g1_is_torsion_free(BlsG1, p) for p in opts]
    assert exp[:5] == [True] * 5 and not all(exp[5:])
    assert G.isTorsionFreeBatch(G.bls12_381_G1_Point, gpts) == exp
    cleared = G.clearCofactorBatch(G.bls12_381_G1_Point, gpts)
    from oracle.h2c import g1_clear_cofactor
    assert [p.toAffine() for p in cleared] == [g1_clear_cofactor(p).toAffine() for p in opts]
    assert all(G.isTorsionFreeBatch(G.bls12_381_G1_Point, cleared))


@pytest.mark.gpu
def test_aggregate_from_bytes_gpu():
    """bls.aggregatePublicKeys-style sum of encoded points (src/abstract/bls.ts:857-873): G1 and G2 compressed
    keys incl. the infinity encoding, secp256k1 and ed25519 encodings; a bad entry is reported by index."""
    from noble_curves_amd import curve as G
    from oracle.curves import BlsG1, BlsG2, Ed25519, Secp256k1
    from oracle.weierstrass import bls_g1_encode_compressed, bls_g2_encode_compressed, sec1_encode
    rng = makeRng(0xA66)
    for Pt, O, encf, n in ((G.bls12_381_G1_Point, BlsG1, bls_g1_encode_compressed, 300),
                           (G.bls12_381_G2_Point, BlsG2, bls_g2_encode_compressed, 70),
                           (G.secp256k1_Point, Secp256k1, sec1_encode, 200),
                           (G.ed25519_Point, Ed25519, lambda p: p.toBytes(), 200)):
        ks = [rng.rndBelow(1 << 64) + 1 for _ in range(n)]
        base = [O.BASE.multiplyUnsafe(k) for k in ks[:8]]
        pts = [base[i % 8].add(base[(i * 3 + 1) % 8]) for i in range(n)]
        encs = [encf(p) for p in pts]
        if O in (BlsG1, BlsG2):
            encs.append(encf(O.ZERO))
        exp = O.ZERO
        for p in pts:
            exp = exp.add(p)
        got = G.aggregateFromBytes(Pt, encs, zip215=True)
        assert got.toAffine() == exp.toAffine() and got.is0() == exp.is0()
        with pytest.raises(ValueError, match="expected non-empty array"):   # bls.ts:426-431 aNonEmpty
            G.aggregateFromBytes(Pt, [])
        bad = list(encs)
        bad[5] = bytes([bad[5][0] ^ 0xFF]) + bad[5][1:] if O is not Ed25519 else bytes([0xEE] * 31 + [0x7F])
        try:
            O.fromBytes(bad[5]) if O is Ed25519 else None
            ed_ok = O is Ed25519
        except ValueError:
            ed_ok = False
        if not ed_ok:
            with pytest.raises(ValueError, match="invalid point encoding at index 5"):
                G.aggregateFromBytes(Pt, bad)


@pytest.mark.gpu
def test_sec1_point_compress_vectors_gpu():
    """test/secp256k1.test.ts:104-113: the 240 pointCompress vectors through fromBytesBatch / toBytesBatch"""
    from noble_curves_amd import curve as G
    rows = [r for r in load_golden("secp256k1_point_compress.json") if r["compress"]]
    pts = []
    comp_in = [bytes.fromhex(r["P"]) for r in rows if len(r["P"]) == 66]
    dec = iter(G.fromBytesBatch(G.secp256k1_Point, comp_in))
    for r in rows:
        if len(r["P"]) == 66:
            pts.append(next(dec))
        else:
            b = bytes.fromhex(r["P"])
            pts.append(G.secp256k1_Point.fromAffine((int.from_bytes(b[1:33], "big"), int.from_bytes(b[33:], "big"))))
    assert all(p is not None for p in pts)
    enc = G.toBytesBatch(G.secp256k1_Point, pts)
    assert [e.hex() for e in enc] == [r["expected"] for r in rows]


@pytest.mark.gpu
def test_reference_inline_multiply_kats_gpu():
    """Inline known answers of the reference's tests: the four Monero scalar -> point vectors
    (test/ed25519.test.ts:312-333), the bls12-381 MUL_VECTORS / naive double-and-add scalars for G1 and G2
    (test/bls12-381.test.ts:645-702) through the variable-base AND the fixed-base kernels, n * BASE == O."""
    from noble_curves_amd import curve as G
    from oracle.curves import BLS_R, BlsG1, BlsG2
    xmr = [("090af56259a4b6bfbc4337980d5d75fbe3c074630368ff3804d33028e5dbfa77",
            "0f3b913371411b27e646b537e888f685bf929ea7aab93c950ed84433f064480d"),
           ("00364e8711a60780382a5d57b061c126f039940f28a9e91fe039d4d3094d8b88",
            "ad545340b58610f0cd62f17d55af1ab11ecde9c084d5476865ddb4dbda015349"),
           ("0b9bf90ff3abec042752cac3a07a62f0c16cfb9d32a3fc2305d676ec2d86e941",
            "e097c4415fe85724d522b2e449e8fd78dd40d20097bdc9ae36fe8ec6fe12cb8c"),
           ("069d896f02d79524c9878e080308180e2859d07f9f54454e0800e8db0847a46e",
            "f12cb7c43b59971395926f278ce7c2eaded9444fbce62ca717564cb508a0db1d")]
    pts = G.multiplyBaseBatch(G.ed25519_Point, [int(s, 16) for s, _ in xmr])
    assert [e.hex() for e in G.toBytesBatch(G.ed25519_Point, pts)] == [p for _, p in xmr]
    mul_vectors = [0x28B90DEAF189015D3A325908C5E0E4BF00F84F7E639B056FF82D7E70B6EEDE4C,
                   0x13EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
                   0x23EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
                   0x33EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
                   0x43EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
                   0x53EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
                   0x63EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000000]
    ks = mul_vectors + [1, 2, BLS_R - 1, (1 << 128) + 1, 12345]
    for Pt, O in ((G.bls12_381_G1_Point, BlsG1), (G.bls12_381_G2_Point, BlsG2)):
        via_var = G.multiplyBatch(Pt, [Pt.BASE] * len(ks), ks)
        via_base = G.multiplyBaseBatch(Pt, ks)
        for k, a, b in zip(ks, via_var, via_base):
            acc, base, s = O.ZERO, O.BASE, k                  # the reference's naiveMul
            while s > 0:
                if s & 1:
                    acc = acc.add(base)
                if s > 1:
                    base = base.double()
                s >>= 1
            assert a.toAffine() == acc.toAffine() == b.toAffine(), hex(k)
        assert G.isTorsionFreeBatch(Pt, [Pt.BASE, via_var[-1]]) == [True, True]   # n * BASE == O
        fresh = G.multiplyBatch(Pt, [via_var[-1]], [54321])[0]
        assert fresh.toAffine() == O.BASE.multiplyUnsafe(12345 * 54321 % BLS_R).toAffine()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["secp256k1", "bls12_381_G1", "bls12_381_G2", "ed25519"])
def test_pairwise_add_subtract_and_mul_add_gpu(name):
    """Point.add / subtract incl. P = Q, P = -Q and ZERO operands (weierstrass.ts:834-891, edwards.ts:526-545) and
    Point.mulAddUnsafe a*P + b*Q (weierstrass.ts:937-944) in batch, vs the oracle."""
    from noble_curves_amd import curve as G
    from oracle.curves import BlsG1, BlsG2, Ed25519, Secp256k1
    Pt, O = {"secp256k1": (G.secp256k1_Point, Secp256k1), "bls12_381_G1": (G.bls12_381_G1_Point, BlsG1),
             "bls12_381_G2": (G.bls12_381_G2_Point, BlsG2), "ed25519": (G.ed25519_Point, Ed25519)}[name]
    rng = makeRng(0xADD + len(name))
    n = 24 if name != "bls12_381_G2" else 12
    ks = [rng.rndBelow(1 << 100) + 1 for _ in range(n)]
    base = [O.BASE.multiplyUnsafe(k) for k in ks]
    ps = list(base)
    qs = [base[(i + 1) % n] for i in range(n)]
    qs[0] = ps[0]                     # P = Q: doubling
    qs[1] = ps[1].negate()            # P = -Q: ZERO
    ps[2] = O.ZERO                    # ZERO + Q
    qs[3] = O.ZERO                    # P + ZERO
    ps[4], qs[4] = O.ZERO, O.ZERO

    def g(p):
        return Pt.ZERO if p.is0() else Pt.fromAffine(p.toAffine())
    gp, gq = [g(p) for p in ps], [g(q) for q in qs]
    got = G.addBatch(Pt, gp, gq)
    for a, p, q in zip(got, ps, qs):
        e = p.add(q)
        assert a.toAffine() == e.toAffine() and a.is0() == e.is0()
    got = G.subtractBatch(Pt, gp, gq)
    for a, p, q in zip(got, ps, qs):
        e = p.subtract(q)
        assert a.toAffine() == e.toAffine() and a.is0() == e.is0()
    order = Pt.Fn.ORDER
    a_s = [rng.rndBelow(order) for _ in range(n)]
    b_s = [rng.rndBelow(order) for _ in range(n)]
    a_s[5], b_s[6] = 0, 0
    b_s[7] = (order - a_s[7]) % order
    qs2 = list(gq)
    qs2[7] = gp[7]                    # a*P + (n - a)*P = ZERO
    ops, oqs = [p for p in ps], [q for q in qs]
    oqs[7] = ops[7]
    got = G.mulAddUnsafeBatch(Pt, gp, a_s, qs2, b_s)
    for r, p, a, q, b in zip(got, ops, a_s, oqs, b_s):
        e = p.multiplyUnsafe(a).add(q.multiplyUnsafe(b))
        assert r.toAffine() == e.toAffine() and r.is0() == e.is0()
    # ECDSA-verification shape u1*G + u2*P: the first term goes through the fixed-base table
    got = G.mulAddUnsafeBatch(Pt, [Pt.BASE] * n, a_s, qs2, b_s)
    for r, a, q, b in zip(got, a_s, oqs, b_s):
        e = O.BASE.multiplyUnsafe(a).add(q.multiplyUnsafe(b))
        assert r.toAffine() == e.toAffine() and r.is0() == e.is0()


def test_config0_point_multiply_1k_random_scalars():
    """BASELINE configs[0] (benchmark/point.ts:20-32): `P.multiply(k)` / `P.multiplyUnsafe(k)` on one random
    secp256k1 point for the benchmark's literal scalar 2^180 - 15820 and 1 000 random scalars, through the
    shim's batch forms, against the oracle's C restatement (RCB + GLV wNAF-4) and, for a sample, the Python
    oracle."""
    from noble_curves_amd import curve as G
    from oracle import cport
    n = SECP256K1_N
    rng = makeRng(0x6E6F626C6501)
    P = Secp256k1.BASE.multiplyUnsafe(rng.rndBelow(n - 1) + 1)
    ks = [(1 << 180) - 15820] + [rng.rndBelow(n - 1) + 1 for _ in range(1000)]
    gp = G.secp256k1_Point.fromAffine(P.toAffine())
    got_m = G.multiplyBatch(G.secp256k1_Point, [gp] * len(ks), ks)
    got_u = G.multiplyUnsafeBatch(G.secp256k1_Point, [gp] * len(ks), ks)
    exp, _ = cport.multiply_unsafe("secp256k1", points_to_wire(SECP256K1, [P] * len(ks)), scalars_to_wire(ks))
    for i, k in enumerate(ks):
        e = wire_to_affine(SECP256K1, exp[i])
        assert got_m[i].toAffine() == e and got_u[i].toAffine() == e, hex(k)
    for i in (0, 1, 500, 1000):
        assert got_m[i].toAffine() == P.multiply(ks[i]).toAffine()
    with pytest.raises(ValueError, match="invalid scalar: out of range"):
        G.multiplyBatch(G.secp256k1_Point, [gp], [0])            # multiply rejects 0, multiplyUnsafe allows it
    assert G.multiplyUnsafeBatch(G.secp256k1_Point, [gp], [0])[0].is0()
