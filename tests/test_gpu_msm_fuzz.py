"""Randomised MSM parity (round 5): sizes, window widths and scalar SHAPES chosen by a seeded generator, every case through
several entry points, expected value from the oracle by linearity (P_i = (a + i b) G, so sum_i s_i P_i = (sum_i s_i (a + i b)) G -
the construction of the reference's test/slow-curves.test.ts:185-252).

What it is for: the two-level counting sort (coarse ranges, staged regions, the work list for oversized regions), the spread
top windows and the sparse / dense accumulate kernels all branch on the DATA (how many entries a region holds, how full the top
window is, how far apart the occupied buckets lie), and the fixed-size tests visit only a few of those branches.  The window width
is forced through NCG_MSM_C (a public knob of the library, read per call) so that small inputs meet wide windows - sparse buckets,
short top windows of every width - and large ones meet narrow ones."""
import os
import random

import pytest
import torch

import bench
from helpers import ORACLE_CURVE, wire_to_affine
from noble_curves_amd import get_engine
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2, ED25519, SECP256K1

pytestmark = pytest.mark.gpu

NMAX = {SECP256K1: 1 << 17, ED25519: 1 << 17, BLS12_381_G1: 1 << 17, BLS12_381_G2: 1 << 15}


def _scalars(rnd, shape, n, order):
    bits = order.bit_length()
    if shape == "uniform":
        return [rnd.randrange(order) for _ in range(n)]
    if shape == "small":                      # only the low windows are populated
        return [rnd.randrange(1 << 20) for _ in range(n)]
    if shape == "top":                        # order - small: every window near its maximum, the top window full
        return [order - 1 - rnd.randrange(1 << 16) for _ in range(n)]
    if shape == "few":                        # a handful of distinct values: a few giant buckets per window
        vals = [rnd.randrange(order) for _ in range(3)]
        return [vals[rnd.randrange(3)] for _ in range(n)]
    if shape == "one":                        # benchmark/bls12-381.ts:64-79: identical scalars
        v = rnd.randrange(1, order)
        return [v] * n
    if shape == "sparse_top":                 # all but a few scalars leave the top window empty
        cut = 1 << (bits - 3)
        return [rnd.randrange(order) if rnd.randrange(4096) == 0 else rnd.randrange(cut) for _ in range(n)]
    if shape == "zeros":                      # test/slow-curves.test.ts:215: every 17th scalar zero - here most of them
        return [rnd.randrange(order) if rnd.randrange(8) == 0 else 0 for _ in range(n)]
    if shape == "pow2":                       # single bits: one window holds an entry, in its lowest or highest buckets
        return [1 << rnd.randrange(bits - 1) for _ in range(n)]
    raise AssertionError(shape)


SHAPES = ["uniform", "small", "top", "few", "one", "sparse_top", "zeros", "pow2"]


@pytest.mark.parametrize("curve", [BLS12_381_G1, BLS12_381_G2, SECP256K1, ED25519])
def test_randomised_msm_shapes_and_window_widths(curve):
    eng = get_engine()
    dev = torch.device("cuda", 0)
    Pt = ORACLE_CURVE[curve]
    order = Pt.Fn.ORDER
    rnd = random.Random(0xF0220 + curve)
    nmax = NMAX[curve]
    pts, ks = bench.gen_points(eng, curve, Pt, nmax, 0x51ED + curve, 0x1F3, dev, None)
    pts_h = pts.cpu().numpy()
    sizes = [1, 2, 63, 64, 65, 1000, 4095, 4096, 4097, 20479, 20481, 32768, 65537, nmax]
    old = os.environ.get("NCG_MSM_C")
    done = 0
    try:
        for trial in range(72):
            n = min(nmax, sizes[trial % len(sizes)] if trial < 14 else rnd.randrange(1, nmax + 1))
            shape = SHAPES[trial % len(SHAPES)] if trial < 16 else rnd.choice(SHAPES)
            c = rnd.choice([0, 0, 11, 12, 13, 14, 15, 16]) if n >= 32 else 0
            if c:
                os.environ["NCG_MSM_C"] = str(c)
            else:
                os.environ.pop("NCG_MSM_C", None)
            sc = _scalars(rnd, shape, n, order)
            exp = Pt.BASE.multiplyUnsafe(sum(k * s for k, s in zip(ks, sc)) % order)
            want, want_inf = exp.toAffine(), exp.is0()
            tag = (trial, n, shape, c)
            d = torch.from_numpy(bench.ints_to_le_bytes(sc).copy()).to(dev)
            got, inf = eng.msm_dev(curve, n, pts.data_ptr(), d.data_ptr())
            assert wire_to_affine(curve, got) == want and inf == want_inf, tag + ("ncg_msm_dev",)
            plan = eng.msm_last_plan()
            if c:
                assert plan["c"] == c, (tag, plan)
            kind = trial % 4
            if kind == 0:
                got, inf = eng.msm_split_windows_dev(curve, n, 3, pts.data_ptr(), d.data_ptr())
                assert wire_to_affine(curve, got) == want and inf == want_inf, tag + ("windows/3",)
            elif kind == 1:
                got, inf = eng.msm_split_dev(curve, n, 5, pts.data_ptr(), d.data_ptr())
                assert wire_to_affine(curve, got) == want and inf == want_inf, tag + ("points/5",)
            elif kind == 2:
                got, inf = eng.msm(curve, pts_h[:n], d.cpu().numpy())                      # host pointers (parts from 2^17 on)
                assert wire_to_affine(curve, got) == want and inf == want_inf, tag + ("ncg_msm",)
            else:
                rs = eng.upload_points(curve, pts_h[:n])
                got, inf = rs.msm_dev(d.data_ptr())
                assert wire_to_affine(curve, got) == want and inf == want_inf, tag + ("resident",)
                if curve in (BLS12_381_G1, BLS12_381_G2) and n >= 2:
                    assert rs.verify_subgroup() == -1
                    got, inf = rs.msm_dev(d.data_ptr())
                    assert wire_to_affine(curve, got) == want and inf == want_inf, tag + ("endo",)
                rs.free()
            done += 1
    finally:
        if old is None:
            os.environ.pop("NCG_MSM_C", None)
        else:
            os.environ["NCG_MSM_C"] = old
    assert done == 72
