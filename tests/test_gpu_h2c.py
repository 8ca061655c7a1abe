"""GPU parity for hash-to-curve on bls12-381 G1 / G2 through the C ABI."""
import pytest

from noble_curves_amd import curve as GC
from noble_curves_amd import h2c as G
from oracle.curves import BLS_P, BLS_R, makeRng
from oracle.h2c import G1_hasher, G2_hasher, hash_to_field
from oracle.weierstrass import bls_g1_encode_compressed, bls_g2_encode_compressed

from helpers import load_golden

pytestmark = pytest.mark.gpu


def test_eip2537_map_to_curve_gpu():
    """test/bls12-381.test.ts:1605-1626 through the mirror"""
    eip = load_golden("bls12_381_eip2537.json")
    got = G.bls12_381_G1_hasher.mapToCurveBatch([int(v["Input"], 16) for v in eip["G1"]])
    for v, p in zip(eip["G1"], got):
        x, y = p.toAffine()
        assert "%0128x%0128x" % (x, y) == v["Expected"]
    got = G.bls12_381_G2_hasher.mapToCurveBatch([[int(v["Input"][:128], 16), int(v["Input"][128:], 16)] for v in eip["G2"]])
    for v, p in zip(eip["G2"], got):
        x, y = p.toAffine()
        assert "%0128x%0128x%0128x%0128x" % (x[0], x[1], y[0], y[1]) == v["Expected"]
    t = 1006044755431560595281793557931171729984964515682961911911398807521437683216171091013202870577238485832047490326971
    assert G.bls12_381_G1_hasher.mapToCurve(t).equals(GC.bls12_381_G1_Point.ZERO)
    with pytest.raises(ValueError, match=r"expected bigint \(m=1\)"):
        G.bls12_381_G1_hasher.mapToCurve([1])
    with pytest.raises(ValueError, match="expected array of 2 bigints"):
        G.bls12_381_G2_hasher.mapToCurve([1])


def test_signature_vectors_end_to_end_gpu():
    """sig = priv * hashToCurve(msg): hash on the shim, map + clear on the GPU, multiply on the GPU, encode on the
    GPU; 48 + 48 of the reference's priv:msg:sig vectors (test/bls12-381.test.ts:953-966, :1003-1012)."""
    sig = load_golden("bls12_381_sig_vectors.json")
    for rows, hasher, Pt in ((sig["g2"], G.bls12_381_G2_hasher, GC.bls12_381_G2_Point),
                             (sig["g1"], G.bls12_381_G1_hasher, GC.bls12_381_G1_Point)):
        H = hasher.hashToCurveBatch([bytes.fromhex(r["msg"]) for r in rows])
        S = GC.multiplyBatch(Pt, H, [int(r["priv"], 16) % BLS_R for r in rows])
        enc = GC.toBytesBatch(Pt, S)
        assert [e.hex() for e in enc] == [r["sig"] for r in rows]


def test_hash_and_encode_to_curve_match_oracle_gpu():
    rng = makeRng(0x42C)
    msgs = [b"", b"abc", b"a" * 200] + [bytes(rng.rnd64() & 0xFF for _ in range(1 + (i % 40))) for i in range(61)]
    for hasher, oh in ((G.bls12_381_G1_hasher, G1_hasher), (G.bls12_381_G2_hasher, G2_hasher)):
        got = hasher.hashToCurveBatch(msgs)
        for m, p in zip(msgs[:24], got):
            assert p.toAffine() == oh.hashToCurve(m).toAffine()
        got = hasher.encodeToCurveBatch(msgs[:24])
        for m, p in zip(msgs[:24], got):
            assert p.toAffine() == oh.encodeToCurve(m).toAffine()
        custom = hasher.hashToCurve(b"abc", {"DST": "QUUX-V01-CS02-with-BLS12381G1_XMD:SHA-256_SSWU_RO_"})
        assert custom.toAffine() == oh.hashToCurve(b"abc", b"QUUX-V01-CS02-with-BLS12381G1_XMD:SHA-256_SSWU_RO_").toAffine()
    # the shim's hash_to_field is the reference's: same integers as the oracle
    o = dict(G.bls12_381_G2_hasher.defaults)
    assert G.hash_to_field(b"abc", 2, o) == hash_to_field(b"abc", 2, BLS_P, 2, 128, o["DST"].encode())


def test_map_to_curve_edge_inputs_gpu():
    """u = 0 (SWU step 7 exceptional branch), u = 1, u = p - 1, unreduced u >= p"""
    from test_host_logic import h2c_cases
    for hasher, oh, m in ((G.bls12_381_G1_hasher, G1_hasher, 1), (G.bls12_381_G2_hasher, G2_hasher, 2)):
        rows = h2c_cases(m, 1, 12)
        rows.append([BLS_P + 5] * m)
        got = hasher.mapToCurveBatch([r[0] if m == 1 else r for r in rows])
        for r, p in zip(rows, got):
            exp = oh.mapToCurve(r[0] if m == 1 else r)
            assert p.toAffine() == exp.toAffine() and p.is0() == exp.is0()
