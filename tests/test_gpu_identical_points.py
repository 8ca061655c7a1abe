"""The reference's 32 768-IDENTICAL-points MSM, at the benchmark's size (benchmark/bls12-381.ts:64-79).

`Array(amount).fill(0).map((i) => _pow1 - BigInt(i))` maps over the ELEMENT (always 0), not the index: every point
is BASE * 2^235 and every scalar is 2^241 (SURVEY 8a parity gotcha 2; the same shape as test/point.test.ts:267-273).
For a bucket method this is the worst shape there is: every window holds ONE occupied bucket with all 32 768
entries, each lane run starts with P + P, and every level of the bucket fix-up / fold tree adds two EQUAL partial
sums - the doubling branch of the incomplete XYZZ addition on every single operation (`k_msm_fixup_long`,
`k_msm_fixup_merge`, `k_msm_reduce_level_coop`).  Expected value without a naive sum:
    sum_i s P = (n * s mod r) * P.
Also the `+i` variant the benchmark MEANT (points (2^235 - i) G, scalars 2^241 + i), and 2^20 identical points once.
Every MSM entry point, with the segment cuts forced to 1, 64 and the default and the serial-run threshold at 0 / 50."""
import numpy as np
import pytest
import torch

import bench
from helpers import ORACLE_CURVE, points_to_wire, wire_to_affine
from noble_curves_amd import get_engine
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2, POINT_BYTES
from oracle.curves import BLS_R

pytestmark = pytest.mark.gpu

POW1, POW2 = 1 << 235, 1 << 241          # benchmark/bls12-381.ts:60-61
AMOUNT = 32768                           # :63


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _identical(curve, n, k_point=POW1, s=POW2):
    Pt = ORACLE_CURVE[curve]
    P = Pt.BASE.multiplyUnsafe(k_point)
    pw = np.tile(points_to_wire(curve, [P]), (n, 1))
    sw = np.tile(np.frombuffer(int(s).to_bytes(32, "little"), dtype=np.uint8), (n, 1))
    exp = P.multiplyUnsafe(n * s % BLS_R)
    return pw, sw, exp.toAffine(), exp.is0()


def _every_entry_point(eng, curve, pw, sw, exp, tag):
    n = pw.shape[0]
    got, inf = eng.msm(curve, pw, sw)                                         # host pointers (parts + pinned staging)
    assert wire_to_affine(curve, got) == exp and not inf, (tag, "ncg_msm")
    dp, ds = _dev(pw), _dev(sw)
    got, _ = eng.msm_dev(curve, n, dp.data_ptr(), ds.data_ptr())
    assert wire_to_affine(curve, got) == exp, (tag, "ncg_msm_dev")
    for parts in (2, 8):
        got, _ = eng.msm_split_windows_dev(curve, n, parts, dp.data_ptr(), ds.data_ptr())
        assert wire_to_affine(curve, got) == exp, (tag, "windows", parts)
        got, _ = eng.msm_split_dev(curve, n, parts, dp.data_ptr(), ds.data_ptr())
        assert wire_to_affine(curve, got) == exp, (tag, "points", parts)
    for lane in (0, 1):
        eng.msm_async_submit(lane, curve, n, dp.data_ptr(), ds.data_ptr())
    for lane in (0, 1):
        got, _ = eng.msm_async_collect(lane, curve)
        assert wire_to_affine(curve, got) == exp, (tag, "async lane", lane)


def _resident_stages(eng, curve, pw, sw, exp, tag):
    n = pw.shape[0]
    ds = _dev(sw)
    rs = eng.upload_points(curve, pw)
    for stage in ("generic", "endo", "precomputed"):
        if stage == "endo":
            assert rs.verify_subgroup() == -1 and rs.in_subgroup
        if stage == "precomputed":
            assert rs.precompute()
        got, _ = rs.msm(sw)
        assert wire_to_affine(curve, got) == exp, (tag, stage, "resident")
        got, _ = rs.msm_dev(ds.data_ptr())
        assert wire_to_affine(curve, got) == exp, (tag, stage, "resident_dev")
        got, _ = eng.msm_split_windows_dev(curve, n, 8, 0, ds.data_ptr(), resident=rs)
        assert wire_to_affine(curve, got) == exp, (tag, stage, "windows/8")
        eng.msm_async_submit(2, curve, n, 0, ds.data_ptr(), resident=rs)
        got, _ = eng.msm_async_collect(2, curve)
        assert wire_to_affine(curve, got) == exp, (tag, stage, "async")
    rs.free()


@pytest.mark.parametrize("curve", [BLS12_381_G1, BLS12_381_G2])
def test_benchmark_msm_32768_identical_points(curve):
    """pippenger(G1, 32768 x (2^235 G), 32768 x 2^241) exactly as benchmark/bls12-381.ts:77-79 builds it."""
    eng = get_engine()
    pw, sw, exp, inf = _identical(curve, AMOUNT)
    assert not inf
    try:
        for seg in (1, 64, 0):
            for run_serial in (0, 50, -1):
                eng.msm_set_tuning(seg, run_serial)
                tag = "seg=%d run_serial=%d" % (seg, run_serial)
                if seg == 0 and run_serial == -1:
                    _every_entry_point(eng, curve, pw, sw, exp, tag)
                    _resident_stages(eng, curve, pw, sw, exp, tag)
                else:
                    dp, ds = _dev(pw), _dev(sw)
                    got, _ = eng.msm_dev(curve, AMOUNT, dp.data_ptr(), ds.data_ptr())
                    assert wire_to_affine(curve, got) == exp, (tag, "ncg_msm_dev")
                    plan = eng.msm_last_plan()
                    if seg:
                        assert plan["seg"] == seg, plan
                    if run_serial >= 0:
                        assert plan["run_serial"] == run_serial, plan
                    got, _ = eng.msm_split_windows_dev(curve, AMOUNT, 8, dp.data_ptr(), ds.data_ptr())
                    assert wire_to_affine(curve, got) == exp, (tag, "windows/8")
                    rs = eng.upload_points(curve, pw)
                    got, _ = rs.msm_dev(ds.data_ptr())
                    assert wire_to_affine(curve, got) == exp, (tag, "resident")
                    assert rs.verify_subgroup() == -1
                    got, _ = rs.msm_dev(ds.data_ptr())
                    assert wire_to_affine(curve, got) == exp, (tag, "endo")
                    rs.free()
    finally:
        eng.msm_set_tuning(0, -1)


@pytest.mark.parametrize("curve", [BLS12_381_G1, BLS12_381_G2])
def test_identical_points_whose_sum_is_zero_or_flips_sign(curve):
    """n identical points with a scalar s such that n*s = 0 mod r cannot be built with n a power of two (r is odd), so
    the ZERO result comes from pairs: half the scalars s, half r - s.  Every bucket pair cancels at the very top."""
    eng = get_engine()
    pw, sw, _, _ = _identical(curve, AMOUNT)
    neg = np.frombuffer(int(BLS_R - POW2).to_bytes(32, "little"), dtype=np.uint8)
    sw = sw.copy()
    sw[AMOUNT // 2:] = neg
    got, inf = eng.msm(curve, pw, sw)
    assert inf and not got.any()
    rs = eng.upload_points(curve, pw)
    assert rs.verify_subgroup() == -1
    got, inf = rs.msm(sw)
    assert inf and not got.any()
    assert rs.precompute()
    got, inf = rs.msm(sw)
    assert inf and not got.any()
    rs.free()


@pytest.mark.parametrize("curve", [BLS12_381_G1, BLS12_381_G2])
def test_benchmark_msm_32768_as_the_benchmark_meant_it(curve):
    """points (2^235 - i) G, scalars 2^241 + i: what `.map((i) => ...)` was written to produce.  Expected by linearity."""
    eng = get_engine()
    dev = torch.device("cuda", 0)
    Pt = ORACLE_CURVE[curve]
    # P_i = (a + i b) G with a = 2^235, b = r - 1  ==  (2^235 - i) G
    pts, ks = bench.gen_points(eng, curve, Pt, AMOUNT, POW1, BLS_R - 1, dev, None)
    assert ks[5] == (POW1 - 5) % BLS_R
    sc = [POW2 + i for i in range(AMOUNT)]
    exp = Pt.BASE.multiplyUnsafe(sum(k * s for k, s in zip(ks, sc)) % BLS_R).toAffine()
    pw = pts.cpu().numpy()
    sw = bench.ints_to_le_bytes(sc).copy()
    _every_entry_point(eng, curve, pw, sw, exp, "meant")
    _resident_stages(eng, curve, pw, sw, exp, "meant")


def test_one_million_identical_points_g1():
    """2^20 copies of 2^235 G with the scalar 2^241 (the BASELINE size): full-width plan, one bucket per window."""
    eng = get_engine()
    n = 1 << 20
    pw, sw, exp, _ = _identical(BLS12_381_G1, n)
    dp, ds = _dev(pw), _dev(sw)
    got, _ = eng.msm_dev(BLS12_381_G1, n, dp.data_ptr(), ds.data_ptr())
    assert wire_to_affine(BLS12_381_G1, got) == exp
    got, _ = eng.msm_split_windows_dev(BLS12_381_G1, n, 8, dp.data_ptr(), ds.data_ptr())
    assert wire_to_affine(BLS12_381_G1, got) == exp
    rs = eng.upload_points(BLS12_381_G1, pw)
    assert rs.verify_subgroup() == -1
    got, _ = rs.msm_dev(ds.data_ptr())
    assert wire_to_affine(BLS12_381_G1, got) == exp
    assert rs.precompute()
    got, _ = rs.msm_dev(ds.data_ptr())
    assert wire_to_affine(BLS12_381_G1, got) == exp
    rs.free()
    got, _ = eng.msm(BLS12_381_G1, pw, sw)
    assert wire_to_affine(BLS12_381_G1, got) == exp


def test_identical_points_g2_2_18():
    n = 1 << 18
    pw, sw, exp, _ = _identical(BLS12_381_G2, n)
    dp, ds = _dev(pw), _dev(sw)
    eng = get_engine()
    got, _ = eng.msm_dev(BLS12_381_G2, n, dp.data_ptr(), ds.data_ptr())
    assert wire_to_affine(BLS12_381_G2, got) == exp
    rs = eng.upload_points(BLS12_381_G2, pw)
    assert rs.verify_subgroup() == -1
    got, _ = rs.msm_dev(ds.data_ptr())
    assert wire_to_affine(BLS12_381_G2, got) == exp
    rs.free()
