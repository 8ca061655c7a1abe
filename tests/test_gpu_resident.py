"""Resident point sets (include/ncg.h "resident point sets"; the usage pattern of the reference's
interleavedMSMUnsafe closure, src/abstract/curve.ts:907-959): upload once - as points or as compressed
encodings decoded on the device - then pippenger / multiplyUnsafeBatch with only the scalars crossing."""
import numpy as np
import pytest

from noble_curves_amd import curve as G
from oracle import curve as OC
from oracle.curves import BLS_R, BlsG1, SECP256K1_N, Secp256k1, makeRng

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("c,Pt,order", [(G.secp256k1_Point, Secp256k1, SECP256K1_N), (G.bls12_381_G1_Point, BlsG1, BLS_R)])
def test_resident_set_matches_list_api_and_oracle(c, Pt, order):
    rng = makeRng(0x5E7 + order % 97)
    n = 75
    opts = [Pt.BASE.multiplyUnsafe(rng.rndBelow(order - 1) + 1) for _ in range(n)]
    opts[5] = Pt.ZERO
    pts = [c.fromAffine(p.toAffine()) for p in opts]
    sets = [G.uploadPoints(c, pts)]
    enc = G.toBytesBatch(c, [p for i, p in enumerate(pts) if i != 5])
    for _ in range(3):                                           # the same set is reused across calls
        sc = [0 if i % 9 == 4 else rng.rndBelow(order) for i in range(n)]
        exp = OC.pippenger(Pt, opts, sc).toAffine()
        assert G.pippenger(c, sets[0], sc).toAffine() == exp
        got = G.multiplyUnsafeBatch(c, sets[0], sc)
        for g, p, k in zip(got, opts, sc):
            assert g.toAffine() == p.multiplyUnsafe(k).toAffine()
    # from encodings (ZERO has no SEC1 encoding / is left out): same results as the list API
    es = G.uploadEncoded(c, enc)
    sc = [rng.rndBelow(order) for _ in range(n - 1)]
    sub = [p for i, p in enumerate(pts) if i != 5]
    assert G.pippenger(c, es, sc).toAffine() == G.pippenger(c, sub, sc).toAffine()
    with pytest.raises(ValueError, match="equal length"):
        G.pippenger(c, es, sc[:-1])
    with pytest.raises(ValueError, match="invalid scalar at index 0"):
        G.pippenger(c, es, [order] + sc[1:])
    bad = [bytes(e) for e in enc[:4]]
    bad[2] = bytes([7]) + bad[2][1:]
    with pytest.raises(ValueError, match="invalid point encoding at index 2"):
        G.uploadEncoded(c, bad)
    empty = G.uploadPoints(c, [])
    assert G.pippenger(c, empty, []).is0() and G.multiplyUnsafeBatch(c, empty, []) == []
    for s in sets + [es, empty]:
        s.free()
