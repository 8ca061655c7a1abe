"""GPU parity: Pippenger MSM (ncg_msm) against the CPU oracle's restatement of
curve.ts:863-905 and the arithmetic-progression identity of test/slow-curves.test.ts:185-252."""
import numpy as np
import pytest

from noble_curves_amd import get_engine
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2, SECP256K1
from oracle import curve as C
from oracle.curves import makeRng

from helpers import ORACLE_CURVE, points_to_wire, scalars_to_wire, wire_to_affine

pytestmark = pytest.mark.gpu
CURVES = [SECP256K1, BLS12_381_G1, BLS12_381_G2]


def gpu_msm(curve, pts, scalars):
    out, inf = get_engine().msm(curve, points_to_wire(curve, pts), scalars_to_wire(scalars))
    aff = wire_to_affine(curve, out)
    assert inf == (aff == ORACLE_CURVE[curve].ZERO.toAffine())
    return aff


def progression(Pt, n, seed):
    """P_i = (a + i b) G built with adds; s_i random with every 17th zero; returns expected point."""
    order = Pt.Fn.ORDER
    rng = makeRng(seed)
    a, b = rng.rndBelow(order - 1) + 1, rng.rndBelow(order - 1) + 1
    P = Pt.BASE.multiplyUnsafe(a)
    step = Pt.BASE.multiplyUnsafe(b)
    pts, ks = [], []
    for i in range(n):
        pts.append(P)
        ks.append((a + i * b) % order)
        P = P.add(step)
    pts = C.normalizeZ(Pt, pts)
    sc = [0 if i % 17 == 0 else rng.rndBelow(order) for i in range(n)]
    exp = Pt.BASE.multiplyUnsafe(sum(k * s for k, s in zip(ks, sc)) % order)
    return pts, sc, exp


@pytest.mark.parametrize("curve", CURVES)
def test_msm_edge_cases(curve):
    """test/point.test.ts:264-305: empty, zeros, P/-P/ZERO, repeated points, single point."""
    Pt = ORACLE_CURVE[curve]
    G, Z = Pt.BASE, Pt.ZERO
    order = Pt.Fn.ORDER
    assert gpu_msm(curve, [], []) == Z.toAffine()
    assert gpu_msm(curve, [G], [0]) == Z.toAffine()
    assert gpu_msm(curve, [G], [1]) == G.toAffine()
    assert gpu_msm(curve, [G], [order - 1]) == G.negate().toAffine()
    assert gpu_msm(curve, [G, G.negate(), Z], [5, 5, 7]) == Z.toAffine()
    assert gpu_msm(curve, [G, G], [0, 0]) == Z.toAffine()
    assert gpu_msm(curve, [G] * 5, [3] * 5) == G.multiplyUnsafe(15).toAffine()
    assert gpu_msm(curve, [Z, Z], [1, 2]) == Z.toAffine()
    # identical points and scalars, as the reference's own benchmark feeds (benchmark/bls12-381.ts:64-79)
    k = (order * 2) // 3
    assert gpu_msm(curve, [G.double()] * 70, [k] * 70) == G.multiplyUnsafe(2 * 70 * k % order).toAffine()


@pytest.mark.parametrize("curve,n", [(SECP256K1, 300), (BLS12_381_G1, 300), (BLS12_381_G2, 120)])
def test_msm_matches_oracle_pippenger(curve, n):
    Pt = ORACLE_CURVE[curve]
    pts, sc, exp = progression(Pt, n, 0x6D736D00 + curve)
    assert C.pippenger(Pt, pts, sc).toAffine() == exp.toAffine()   # oracle self-check
    assert gpu_msm(curve, pts, sc) == exp.toAffine()


@pytest.mark.parametrize("curve,n", [(BLS12_381_G1, 5000), (SECP256K1, 3000), (BLS12_381_G2, 1500)])
def test_msm_progression_medium(curve, n):
    """Sizes past what the oracle's pippenger finishes quickly: pinned by the progression identity."""
    Pt = ORACLE_CURVE[curve]
    pts, sc, exp = progression(Pt, n, 0xABCD00 + curve)
    assert gpu_msm(curve, pts, sc) == exp.toAffine()


def test_msm_non_normalised_inputs_and_window_override(monkeypatch):
    """Points that went through add() (Z != 1 on the reference side) are normalised by the host
    shim before crossing (SURVEY 8a gotcha 3); also exercise several window sizes."""
    Pt = ORACLE_CURVE[BLS12_381_G1]
    pts, sc, exp = progression(Pt, 700, 0x77)
    for c in ("3", "7", "11", "16"):
        monkeypatch.setenv("NCG_MSM_C", c)
        assert gpu_msm(BLS12_381_G1, pts, sc) == exp.toAffine()


@pytest.mark.parametrize("seg,run_serial", [(1, 0), (1, 2), (3, 0), (3, 1), (64, -1), (100000, -1), (7, 50)])
def test_msm_segment_lengths_and_skewed_buckets(monkeypatch, seg, run_serial):
    """Lane-segment boundaries cutting buckets in every possible way, including one bucket that holds every entry (all
    scalars equal) and many empty buckets; with run_serial 0 / 1 every cut bucket takes the long-run work list
    (k_msm_fixup_long), with 50 none does (k_msm_fixup_merge adds every piece serially).  The segment is set through
    ncg_msm_set_tuning and ncg_msm_last_plan proves that the launch used it (the NCG_MSM_SEG environment
    knob is honoured by A/B builds only, so the old form of this test ran one configuration four times)."""
    monkeypatch.setenv("NCG_MSM_C", "6")
    eng = get_engine()
    Pt = ORACLE_CURVE[BLS12_381_G1]
    pts, sc, exp = progression(Pt, 900, 0x5E6)
    k = 0x1F                                            # every point lands in the same low bucket
    tot = C.normalizeZ(Pt, [sum_points(Pt, pts)])[0]
    try:
        eng.msm_set_tuning(seg, run_serial)
        assert gpu_msm(BLS12_381_G1, pts, sc) == exp.toAffine()
        lp = eng.msm_last_plan()
        assert lp["seg"] == seg and lp["c"] == 6 and lp["nb"] == 32
        if run_serial >= 0:
            assert lp["run_serial"] == run_serial
        if seg <= 3 and run_serial in (0, 1):           # 900 entries over 32 buckets: every bucket spans many lanes
            assert lp["long_runs"] >= 32, lp
        if run_serial == 50:
            assert lp["long_runs"] == 0, lp
        assert gpu_msm(BLS12_381_G1, pts, [k] * 900) == tot.multiplyUnsafe(k).toAffine()
        lp = eng.msm_last_plan()
        assert lp["seg"] == seg
        if seg < 900 and run_serial in (0, 1, 2):
            assert lp["long_runs"] >= 1, lp             # the one bucket of window 0 is a single long run
        # the other pipelines under the same cut: sharded by points and by windows, shared-bucket sets need >= 4096 points
        pw, sw = points_to_wire(BLS12_381_G1, pts), scalars_to_wire(sc)
        import torch
        dp, ds = torch.from_numpy(pw).cuda(), torch.from_numpy(sw).cuda()
        for parts in (2, 5):
            got, _ = eng.msm_split_windows_dev(BLS12_381_G1, 900, parts, dp.data_ptr(), ds.data_ptr())
            assert wire_to_affine(BLS12_381_G1, got) == exp.toAffine(), ("windows", parts)
            got, _ = eng.msm_split_dev(BLS12_381_G1, 900, parts, dp.data_ptr(), ds.data_ptr())
            assert wire_to_affine(BLS12_381_G1, got) == exp.toAffine(), ("points", parts)
    finally:
        eng.msm_set_tuning(0, -1)
    gpu_msm(BLS12_381_G1, pts[:50], sc[:50])
    assert eng.msm_last_plan()["seg"] != 100000          # the defaults are back


def test_msm_refuses_misaligned_device_buffers():
    """include/ncg.h: device buffers are read with 16-byte accesses; a 4- or 8-byte-aligned pointer is an
    NCG_ERR_INVALID_ARG, not a memory fault."""
    import torch
    eng = get_engine()
    Pt = ORACLE_CURVE[BLS12_381_G1]
    pts, sc, exp = progression(Pt, 40, 0xA11)
    pw, sw = points_to_wire(BLS12_381_G1, pts), scalars_to_wire(sc)
    buf = torch.zeros(pw.size + sw.size + 64, dtype=torch.uint8, device="cuda")
    for off in (4, 8):
        p = buf.data_ptr() + off
        with pytest.raises(Exception, match="16-byte aligned"):
            eng.msm_dev(BLS12_381_G1, 40, p, buf.data_ptr() + 8192)
        with pytest.raises(Exception, match="16-byte aligned"):
            eng.msm_dev(BLS12_381_G1, 40, buf.data_ptr(), p)
        with pytest.raises(Exception, match="16-byte aligned"):
            eng.msm_split_dev(BLS12_381_G1, 40, 2, p, buf.data_ptr())
        with pytest.raises(Exception, match="16-byte aligned"):
            eng.msm_split_windows_dev(BLS12_381_G1, 40, 2, buf.data_ptr(), p)
    assert gpu_msm(BLS12_381_G1, pts, sc) == exp.toAffine()      # the context keeps working
    from noble_curves_amd import fft as gfft
    om = gfft.rootsOfUnity(gfft.bls12_381_Fr, 7).omega(5)
    with pytest.raises(Exception, match="16-byte aligned"):
        eng.ntt_dev(5, 1, om, buf.data_ptr() + 8, buf.data_ptr() + 4096, None)
    with pytest.raises(Exception, match="16-byte aligned"):
        eng.ntt_dev(5, 1, om, buf.data_ptr(), buf.data_ptr() + 4100, None)


def sum_points(Pt, pts):
    acc = Pt.ZERO
    for p in pts:
        acc = acc.add(p)
    return acc
