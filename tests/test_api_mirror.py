"""The Python host mirror of the reference's Point / pippenger interface: input validation and
error messages (no GPU needed) and, on a GPU, the reference's own test flow for the path
(test/point.test.ts:264-305, test/secp256k1.test.ts:59-71)."""
import pytest

from noble_curves_amd import curve as G
from oracle.curves import BlsG1, SECP256K1_N, Secp256k1

from helpers import load_golden

K1, G1, G2 = G.secp256k1_Point, G.bls12_381_G1_Point, G.bls12_381_G2_Point


@pytest.mark.parametrize("Pt", [K1, G1, G2])
def test_validation_messages_match_reference(Pt):
    g, n = Pt.BASE, Pt.Fn.ORDER
    with pytest.raises(ValueError, match="invalid point at index 1"):       # curve.ts:393
        G.pippenger(Pt, [g, "x"], [1, 2])
    with pytest.raises(ValueError, match="invalid scalar at index 0"):      # curve.ts:399-402
        G.pippenger(Pt, [g, g], [n, 2])
    with pytest.raises(ValueError, match="invalid scalar at index 1"):
        G.pippenger(Pt, [g, g], [1, -1])
    with pytest.raises(TypeError, match="array of scalars expected"):
        G.pippenger(Pt, [g], 5)
    with pytest.raises(ValueError, match="arrays of points and scalars must have equal length"):  # curve.ts:875
        G.pippenger(Pt, [g, g], [1])
    assert G.pippenger(Pt, [], []) is Pt.ZERO                              # curve.ts:878
    for bad in (0, n, n + 1, -1):                                           # weierstrass.ts:904
        with pytest.raises(ValueError, match="invalid scalar: out of range"):
            g.multiply(bad)
    for bad in (n, -1, 1.5):                                                # weierstrass.ts:920 (0 is allowed)
        with pytest.raises(ValueError, match="invalid scalar: out of range"):
            g.multiplyUnsafe(bad)
    assert Pt.fromAffine(Pt.ZERO.toAffine()) is Pt.ZERO                     # weierstrass.ts:716
    assert g.negate().negate().equals(g) and not g.negate().equals(g)
    with pytest.raises(ValueError, match="invalid affine point"):
        Pt.fromAffine((Pt.Fp.ORDER, 1) if Pt.Fp.degree == 1 else ((Pt.Fp.ORDER, 0), (0, 1)))


@pytest.mark.gpu
def test_reference_flow_on_gpu():
    for k, x, y in load_golden("secp256k1_privates2.json")[:12]:
        assert K1.BASE.multiply(int(k)).toAffine() == (int(x, 16), int(y, 16))
    g = K1.BASE
    assert g.multiplyUnsafe(0) is K1.ZERO and g.multiplyUnsafe(1).equals(g)
    assert g.add(g).equals(g.double()) and g.add(g.negate()).is0() and g.subtract(g).is0()
    a, b = g.multiplyUnsafe(0xABCDEF), g.multiplyUnsafe(SECP256K1_N - 5)
    exp = Secp256k1.BASE.multiplyUnsafe((3 * 0xABCDEF + 7 * (SECP256K1_N - 5)) % SECP256K1_N).toAffine()
    assert G.pippenger(K1, [a, b, K1.ZERO], [3, 7, 99]).toAffine() == exp
    outs = G.multiplyUnsafeBatch(G1, [G1.BASE] * 4, [0, 1, 2, 12345])
    assert [o.toAffine() for o in outs] == [BlsG1.BASE.multiplyUnsafe(k).toAffine() for k in (0, 1, 2, 12345)]
    h = G2.BASE.multiplyUnsafe(5)
    assert G.pippenger(G2, [G2.BASE, h], [5, G2.Fn.ORDER - 1]).is0()
    # ed25519 through the same interface (edwards.ts:555-577, ZERO = (0, 1))
    from oracle.curves import ED25519_L, Ed25519
    E = G.ed25519_Point
    assert E.BASE.multiply(12345).toAffine() == Ed25519.BASE.multiply(12345).toAffine()
    assert E.BASE.multiplyUnsafe(0).is0() and E.ZERO.toAffine() == (0, 1)
    assert E.BASE.add(E.BASE.negate()).is0() and E.BASE.double().equals(E.BASE.add(E.BASE))
    assert G.pippenger(E, [E.BASE, E.BASE.double()], [3, ED25519_L - 1]).equals(E.BASE)
    with pytest.raises(ValueError, match="expected 1 <= sc < curve.n"):
        E.BASE.multiply(0)
    with pytest.raises(ValueError, match="expected 0 <= sc < curve.n"):
        E.BASE.multiplyUnsafe(ED25519_L)
    ks = [1, 2, ED25519_L - 1, 0xDEADBEEF]
    assert [p.toAffine() for p in G.multiplyBaseBatch(E, ks)] == [Ed25519.BASE.multiplyUnsafe(k).toAffine() for k in ks]
    assert [p.toAffine() for p in G.multiplyBaseBatch(K1, ks)] == [Secp256k1.BASE.multiplyUnsafe(k).toAffine() for k in ks]


def test_interleaved_msm_validation_messages():
    """interleavedMSMUnsafe's argument checks happen before anything crosses (curve.ts:943-947,
    test/point.test.ts:317)."""
    for bad in (1, 0, 257, 2.5, True, None):
        with pytest.raises(ValueError, match="invalid window size"):
            G.interleavedMSMUnsafe(K1, [K1.BASE], bad)
    with pytest.raises(ValueError, match="invalid point at index 1"):
        G.interleavedMSMUnsafe(K1, [K1.BASE, 5], 4)
    with pytest.raises(TypeError, match="array expected"):
        G.interleavedMSMUnsafe(K1, None, 4)
    msm = G.interleavedMSMUnsafe(K1, [], 4)           # nothing to upload: no engine needed
    assert msm([]) is K1.ZERO
    with pytest.raises(ValueError, match="array of scalars must not be larger than array of points"):
        msm([1])


@pytest.mark.gpu
def test_interleaved_msm_on_gpu():
    """test/point.test.ts:309-317 (3G + 5*2G + 7*4G + 11*8G = 129G for window sizes 2..10) on every
    curve, :854-858 (L = 3, W = 5), the closure's scalar rules (curve.ts:949-952) and a random fixed set
    against the oracle's pippenger."""
    import random
    from oracle.curve import pippenger as o_pippenger
    from oracle.curves import BlsG2, Ed25519
    rng = random.Random(77)
    for Pt, O in ((K1, Secp256k1), (G1, BlsG1), (G2, BlsG2), (G.ed25519_Point, Ed25519)):
        g = Pt.BASE
        pts = [g, g.multiplyUnsafe(2), g.multiplyUnsafe(4), g.multiplyUnsafe(8)]
        want = O.BASE.multiplyUnsafe(129).toAffine()
        for W in range(2, 11):
            mul = G.interleavedMSMUnsafe(Pt, pts, W)
            assert mul([3, 5, 7, 11]).toAffine() == want
        mul = G.interleavedMSMUnsafe(Pt, pts, 5)
        assert mul([3, 5]).toAffine() == O.BASE.multiplyUnsafe(13).toAffine()      # trailing scalars = 0
        assert mul([]).is0() and mul([0, 0, 0, 0]).is0()
        with pytest.raises(ValueError, match="array of scalars must not be larger than array of points"):
            mul([1] * 5)
        with pytest.raises(ValueError, match="invalid scalar at index 1"):
            mul([1, Pt.Fn.ORDER])
        n = Pt.Fn.ORDER
        ks = [rng.randrange(1, n) for _ in range(37)]
        opts = [O.BASE.multiplyUnsafe(k) for k in ks]
        fixed = G.interleavedMSMUnsafe(Pt, G.multiplyBaseBatch(Pt, ks), 6)
        for _ in range(3):                                                         # same set, new scalars
            ss = [rng.randrange(n) for _ in range(37)]
            assert fixed(ss).toAffine() == o_pippenger(O, opts, ss).toAffine()


def test_h2c_shim_hash_to_field_matches_oracle():
    """The shim's expand_message_xmd / hash_to_field (hash-to-curve.ts:189-228, :312-378) against the oracle's
    restatement (pinned by the reference's signature vectors): short / long messages, empty and oversize
    (> 255 byte) DSTs, both extension degrees; error paths of the reference."""
    from noble_curves_amd import h2c as G
    from oracle.curves import BLS_P, makeRng
    from oracle.h2c import expand_message_xmd, hash_to_field
    rng = makeRng(0x4D5D)
    for i in range(24):
        msg = bytes(rng.rnd64() & 0xFF for _ in range([0, 1, 31, 32, 33, 200][i % 6]))
        dst = bytes(rng.rnd64() & 0xFF for _ in range([1, 16, 43, 255, 256, 300][(i // 2) % 6]))
        n = [1, 32, 33, 64, 128, 256][(i // 3) % 6]
        assert G.expand_message_xmd(msg, dst, n) == expand_message_xmd(msg, dst, n)
        for m in (1, 2):
            opts = {"p": BLS_P, "m": m, "k": 128, "DST": dst}
            assert G.hash_to_field(msg, 2, opts) == hash_to_field(msg, 2, BLS_P, m, 128, dst)
    with pytest.raises(ValueError, match="invalid lenInBytes"):
        G.expand_message_xmd(b"", b"d", 65536)
    with pytest.raises(ValueError, match="expected count >= 1"):
        G.hash_to_field(b"", 0, {"p": BLS_P, "m": 1, "k": 128, "DST": b"d"})
    with pytest.raises(ValueError, match="expected m >= 1"):
        G.hash_to_field(b"", 1, {"p": BLS_P, "m": 0, "k": 128, "DST": b"d"})
    assert G.bls12_381_G2_hasher.defaults["DST"] == "BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_NUL_"
    assert G.bls12_381_G1_hasher.defaults["m"] == 1 and G.bls12_381_G2_hasher.defaults["m"] == 2
