"""Precomputed resident sets (ncg_points_precompute: the device form of the per-point tables of the reference's
interleavedMSMUnsafe, src/abstract/curve.ts:907-959): window-shifted copies of the set, every window of the MSM adds
into ONE bucket set.  Same group element as pippenger (curve.ts:863-905), bit for bit: compared with the generic
resident path, with the progression identity of test/slow-curves.test.ts:185-252 and - small sets - with the oracle;
identical scalars (benchmark/bls12-381.ts:64-79), zero scalars, n - 1, ZERO members, repeated and negated points,
verified (endomorphism) sets and sets that stay generic."""
import numpy as np
import pytest
import torch

import bench
from helpers import ORACLE_CURVE, wire_to_affine
from noble_curves_amd import get_engine
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2, ED25519, SECP256K1

pytestmark = pytest.mark.gpu


def _ints_to_dev(vals, dev):
    return torch.from_numpy(bench.ints_to_le_bytes(vals).copy()).to(dev)


@pytest.mark.parametrize("curve,lg", [(SECP256K1, 13), (BLS12_381_G1, 14), (BLS12_381_G2, 12)])
def test_precomputed_set_equals_generic_path_and_identity(curve, lg):
    eng = get_engine()
    dev = torch.device("cuda", 0)
    Pt = ORACLE_CURVE[curve]
    order = Pt.Fn.ORDER
    n = 1 << lg
    pts, ks = bench.gen_points(eng, curve, Pt, n, 0xABCDEF1, 0x1357, dev, None)
    pts_h = pts.cpu().numpy().copy()
    pts_h[5] = 0                               # ZERO member
    pts_h[9] = pts_h[8]                        # repeated point
    ks[5], ks[9] = 0, ks[8]
    res = eng.upload_points(curve, pts_h)
    sc_sets = []
    sc = bench.scalars_to_ints(bench.gen_scalars(n, 250, 11, dev))
    sc[::17] = [0] * len(sc[::17])
    sc[3], sc[4] = order - 1, 1
    sc_sets.append(sc)
    sc_sets.append([0xDEADBEEFCAFEF00D1234567 % order] * n)          # identical scalars: every window one bucket
    sc_sets.append([0] * n)
    sc_sets.append([(order - 1) if i % 2 else (1 << (16 * (i % 16))) for i in range(n)])   # window edges
    before = []
    for sc in sc_sets:
        d = _ints_to_dev(sc, dev)
        out, inf = res.msm_dev(d.data_ptr())
        exp = Pt.BASE.multiplyUnsafe(sum(k * s for k, s in zip(ks, sc)) % order)
        assert wire_to_affine(curve, out) == exp.toAffine() and inf == exp.is0()
        before.append((out.copy(), inf))
    assert not res.precomputed
    assert res.precompute() and res.precomputed
    for sc, (o0, i0) in zip(sc_sets, before):
        d = _ints_to_dev(sc, dev)
        out, inf = res.msm_dev(d.data_ptr())
        assert np.array_equal(out, o0) and inf == i0
    # an out-of-range scalar is still refused (validateMSMScalars, curve.ts:398-404)
    bad = list(sc_sets[0])
    bad[7] = order
    with pytest.raises(Exception, match="invalid scalar at index 7"):
        res.msm_dev(_ints_to_dev(bad, dev).data_ptr())
    if curve in (BLS12_381_G1, BLS12_381_G2):   # verified set: the shifted copies of the endomorphism images
        res2 = eng.upload_points(curve, pts_h)
        assert res2.verify_subgroup() == -1 and res2.precompute()
        for sc, (o0, i0) in zip(sc_sets, before):
            out, inf = res2.msm_dev(_ints_to_dev(sc, dev).data_ptr())
            assert np.array_equal(out, o0) and inf == i0
        res2.free()
    res.free()


def test_small_and_edwards_sets_are_left_alone():
    eng = get_engine()
    dev = torch.device("cuda", 0)
    Pt = ORACLE_CURVE[BLS12_381_G1]
    pts, ks = bench.gen_points(eng, BLS12_381_G1, Pt, 100, 3, 5, dev, None)
    res = eng.upload_points(BLS12_381_G1, pts.cpu().numpy())
    assert res.precompute() is False
    sc = [7 * i + 1 for i in range(100)]
    out, _ = res.msm_dev(_ints_to_dev(sc, dev).data_ptr())
    assert wire_to_affine(BLS12_381_G1, out) == Pt.BASE.multiplyUnsafe(sum(k * s for k, s in zip(ks, sc)) % Pt.Fn.ORDER).toAffine()
    res.free()
    Ed = ORACLE_CURVE[ED25519]
    pe, ke = bench.gen_points(eng, ED25519, Ed, 4096, 3, 5, dev, None)
    re_ = eng.upload_points(ED25519, pe.cpu().numpy())
    assert re_.precompute() is False
    re_.free()


def test_interleaved_msm_unsafe_uses_the_precomputation():
    from noble_curves_amd import curve as G
    from oracle.curves import BLS_R, BlsG1
    eng = get_engine()
    dev = torch.device("cuda", 0)
    n = 4096
    pts, ks = bench.gen_points(eng, BLS12_381_G1, BlsG1, n, 77, 13, dev, None)
    P = [G.bls12_381_G1_Point.fromAffine(wire_to_affine(BLS12_381_G1, row)) for row in pts.cpu().numpy()]
    f = G.interleavedMSMUnsafe(G.bls12_381_G1_Point, P, 4)
    sc = [(i * i * 0x9E3779B97F4A7C15 + 1) % BLS_R for i in range(n - 10)]      # fewer scalars: trailing zeros
    got = f(sc)
    assert got.toAffine() == BlsG1.BASE.multiplyUnsafe(sum(k * s for k, s in zip(ks, sc)) % BLS_R).toAffine()
