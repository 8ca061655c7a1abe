"""The input families of the reference's own MSM benchmark, benchmark/msm_timings.ts:10-87, replayed on the GPU.

The benchmark checks `pippenger(G1, p, s)` against sum_i p_i.multiplyUnsafe(s_i) before it times anything (`check`,
:21-26).  Its scalars are bit PATTERNS chosen to stress a windowed method: `ones` = 2^254 - 1 (every window full: the
signed-digit carry of the GPU's window recoding runs through all 16 windows), `'10'` repeated (onezero), `'10000000'`
repeated (one8zero), N - 1 ... N - 400 (top window at its maximum), zeros, ones, and the point at infinity.  Here
every family goes through ncg_msm, ncg_msm_resident (generic, endomorphism and precomputed plans), the window-sharded
and asynchronous entry points and ncg_mul_var_batch; expected values come from oracle.curve.pippenger
(src/abstract/curve.ts:863-905 restated) and from the benchmark's own `sum` of multiplyUnsafe.  The same patterns tiled
to 2^16 points (where the plan is c = 13 / 16 windows wide) are pinned by the linearity of the MSM in the points."""
import numpy as np
import pytest
import torch

import bench
from helpers import ORACLE_CURVE, points_to_wire, scalars_to_wire, wire_to_affine
from noble_curves_amd import get_engine
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2
from oracle import curve as OC
from oracle.curves import BLS_R, BlsG1, BlsG2

pytestmark = pytest.mark.gpu

N = BLS_R
BITS = N.bit_length() - 1                                   # g1.Fn.BITS - 1 = 254 (msm_timings.ts:10)
ONES = int("1" * BITS, 2)                                   # :11
ONEZERO = int("10" * (BITS // 2), 2)                        # :27
ONE8ZERO = int("10000000" * (BITS // 8), 2)                 # :28
assert ONES < N and ONEZERO < N and ONE8ZERO < N


def _sum(Pt, pts, scalars):
    """the benchmark's reference value (:15-19)"""
    res = Pt.ZERO
    for p, s in zip(pts, scalars):
        res = res.add(p.multiplyUnsafe(s))
    return res


def _families(Pt):
    G, Z = Pt.BASE, Pt.ZERO
    single = {                                                # :29-35
        "zero": ([G], [0]), "one": ([G], [1]), "one0": ([Z], [1]), "small": ([G], [123]), "big": ([G], [N - 1]),
    }
    points = [G.multiply(i) for i in (3, 5, 7, 11, 13)]       # :44
    multi = {                                                 # :45-67
        "zero": ([G] * 5, [0] * 5),
        "zero2": ([Z] * 5, [0] * 5),
        "big": (points, [N - 1, N - 100, N - 200, N - 300, N - 400]),
        "same_scalar": (points, [ONES] * 5),
        "same_scalar2": (points, [ONEZERO] * 5),
        "same_scalar3": (points, [1] * 5),
        "same_scalar4": (points, [ONE8ZERO] * 5),
    }
    out = {"single/" + k: v for k, v in single.items()}
    out.update({"multi/" + k: v for k, v in multi.items()})
    return out


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("curve", [BLS12_381_G1, BLS12_381_G2])
def test_msm_timings_families_through_every_msm_entry_point(curve):
    """single-point and 5-point cases of msm_timings.ts:29-67 (G2 gets the same patterns: configs[4])."""
    eng = get_engine()
    Pt = ORACLE_CURVE[curve]
    for name, (pts, sc) in _families(Pt).items():
        want = _sum(Pt, pts, sc)
        assert OC.pippenger(Pt, pts, sc).equals(want), name          # the benchmark's own check (:21-26) on the oracle
        exp = want.toAffine()
        pw, sw = points_to_wire(curve, pts), scalars_to_wire(sc)
        got, inf = eng.msm(curve, pw, sw)
        assert wire_to_affine(curve, got) == exp and inf == want.is0(), name
        dp, ds = _dev(pw), _dev(sw)
        n = len(pts)
        for parts in (2, 8):
            got, _ = eng.msm_split_windows_dev(curve, n, parts, dp.data_ptr(), ds.data_ptr())
            assert wire_to_affine(curve, got) == exp, (name, "windows", parts)
            got, _ = eng.msm_split_dev(curve, n, parts, dp.data_ptr(), ds.data_ptr())
            assert wire_to_affine(curve, got) == exp, (name, "points", parts)
        eng.msm_async_submit(1, curve, n, dp.data_ptr(), ds.data_ptr())
        got, _ = eng.msm_async_collect(1, curve)
        assert wire_to_affine(curve, got) == exp, (name, "async")
        rs = eng.upload_points(curve, pw)
        got, _ = rs.msm(sw)
        assert wire_to_affine(curve, got) == exp, (name, "resident")
        if all(not p.is0() for p in pts):
            assert rs.verify_subgroup() == -1 and rs.in_subgroup     # endomorphism plan (split scalars)
            got, _ = rs.msm(sw)
            assert wire_to_affine(curve, got) == exp, (name, "endo")
            got, _ = eng.msm_split_windows_dev(curve, n, 3, 0, ds.data_ptr(), resident=rs)
            assert wire_to_affine(curve, got) == exp, (name, "endo windows")
        rs.free()


def test_msm_timings_basic_multiply_family():
    """msm_timings.ts:71-87: k*G1 and k*Inf for k in {1, N-1, ones, onezero, one8zero} through the batch ladder."""
    eng = get_engine()
    G, Z = BlsG1.BASE, BlsG1.ZERO
    ks = [1, N - 1, ONES, ONEZERO, ONE8ZERO]
    pts = [G] * 5 + [Z] * 5
    sc = ks + ks
    out, inf = eng.mul_var_batch(BLS12_381_G1, points_to_wire(BLS12_381_G1, pts), scalars_to_wire(sc))
    for i, (p, k) in enumerate(zip(pts, sc)):
        want = p.multiplyUnsafe(k)
        assert wire_to_affine(BLS12_381_G1, out[i]) == want.toAffine() and bool(inf[i]) == want.is0(), i
    assert wire_to_affine(BLS12_381_G1, out[1]) == G.negate().toAffine()          # '(n-1)*G1' (:73)
    assert all(inf[5:])                                                            # k * Inf = Inf (:82-86)
    rs = eng.upload_points(BLS12_381_G1, points_to_wire(BLS12_381_G1, pts))
    out2, inf2 = rs.mul_var_batch(scalars_to_wire(sc))
    assert np.array_equal(out2, out) and np.array_equal(inf2, inf)
    rs.free()


@pytest.mark.parametrize("curve,lg", [(BLS12_381_G1, 16), (BLS12_381_G2, 14)])
def test_msm_timings_patterns_tiled_to_full_width_plans(curve, lg):
    """The same scalar patterns over 2^16 (G2: 2^14) points, where the plan runs 13-bit windows and every pattern
    fills a handful of buckets with thousands of entries each.  Expected value by linearity: the points are
    P_i = (a + i b) G, so sum_i s_i P_i = (sum_i s_i (a + i b)) G."""
    eng = get_engine()
    dev = torch.device("cuda", 0)
    Pt = ORACLE_CURVE[curve]
    n = 1 << lg
    pts, ks = bench.gen_points(eng, curve, Pt, n, 0x1234567, 0x89ABCD, dev, None)
    pats = [ONES, ONEZERO, ONE8ZERO, N - 1, N - 100, N - 200, N - 300, N - 400, 1, 0, 123]
    rs = eng.upload_points(curve, pts.cpu().numpy())
    cases = {
        "each pattern in turn": [pats[i % len(pats)] for i in range(n)],
        "ones everywhere": [ONES] * n,
        "onezero everywhere": [ONEZERO] * n,
        "one8zero everywhere": [ONE8ZERO] * n,
        "N-1 .. N-400 cycling": [N - 1 - 100 * (i % 5) if i % 5 else N - 1 for i in range(n)],
    }
    for stage in ("generic", "endo", "precomputed"):
        if stage == "endo":
            assert rs.verify_subgroup() == -1
        if stage == "precomputed":
            assert rs.precompute()
        for name, sc in cases.items():
            exp = Pt.BASE.multiplyUnsafe(sum(k * s for k, s in zip(ks, sc)) % N).toAffine()
            d = torch.from_numpy(bench.ints_to_le_bytes(sc).copy()).to(dev)
            got, _ = rs.msm_dev(d.data_ptr())
            assert wire_to_affine(curve, got) == exp, (stage, name)
            got, _ = eng.msm_split_windows_dev(curve, n, 8, 0, d.data_ptr(), resident=rs)
            assert wire_to_affine(curve, got) == exp, (stage, name, "windows/8")
            if stage == "generic":
                got, _ = eng.msm_dev(curve, n, pts.data_ptr(), d.data_ptr())
                assert wire_to_affine(curve, got) == exp, (name, "msm_dev")
                got, _ = eng.msm_split_windows_dev(curve, n, 5, pts.data_ptr(), d.data_ptr())
                assert wire_to_affine(curve, got) == exp, (name, "windows/5")
    rs.free()
