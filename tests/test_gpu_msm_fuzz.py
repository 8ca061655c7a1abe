"""Randomised MSM parity (round 5): sizes, window widths and scalar SHAPES chosen by a seeded generator, every case through
several entry points, expected value from the oracle by linearity (P_i = (a + i b) G, so sum_i s_i P_i = (sum_i s_i (a + i b)) G -
the construction of the reference's test/slow-curves.test.ts:185-252).

What it is for: the two-level counting sort (coarse ranges, staged regions, the work list for oversized regions), the spread
top windows and the sparse / dense accumulate kernels all branch on the DATA (how many entries a region holds, how full the top
window is, how far apart the occupied buckets lie), and the fixed-size tests visit only a few of those branches.  The window width
is forced through NCG_MSM_C (a public knob of the library, read per call) so that small inputs meet wide windows - sparse buckets,
short top windows of every width - and large ones meet narrow ones.

Round 6: half of the trials also MUTATE a few members of the point set - ZERO members, P / -P pairs with equal
scalars (they cancel inside one bucket) and with unequal ones, repeated points under random scalars, and on the bls12-381 curves
points OUTSIDE the prime-order subgroup (k G + T with T of order 3 / 11 on E(Fp), 13 / 23 on E'(Fp2): legal inputs of the
reference's pippenger, test/point.test.ts:267-291, SURVEY 8a gotchas 1-2).  The expected value is linear on the subgroup part and
the oracle's complete double-and-add on the torsion part."""
import os
import random

import numpy as np

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
FIELD_P = {SECP256K1: (1 << 256) - (1 << 32) - 977, ED25519: (1 << 255) - 19}


def _neg_row(curve, row):
    """wire row of -P (Weierstrass: y -> p - y, componentwise for Fp2; Edwards: x -> p - x); ZERO stays ZERO"""
    from noble_curves_amd._native import FIELD_BYTES
    from oracle.curves import BLS_P
    fb = FIELD_BYTES[curve]
    p = FIELD_P.get(curve, BLS_P)
    vals = [int.from_bytes(bytes(row[i * fb:(i + 1) * fb]), "little") for i in range(len(row) // fb)]
    if curve == ED25519:
        flip = [0]
    elif curve == BLS12_381_G2:
        flip = [2, 3]
    else:
        flip = [1]
    if curve != ED25519 and not any(vals):
        return row.copy()
    for i in flip:
        vals[i] = (p - vals[i]) % p
    return np.frombuffer(b"".join(v.to_bytes(fb, "little") for v in vals), dtype=np.uint8).copy()


def _mutate(rnd, curve, Pt, n, ks, sc, rows):
    """Changes up to 8 members of rows[:n] (and, for the cancelling pairs, their scalars) in place.
    Returns (k_eff, torsion terms [(index, T, q)]): member i is k_eff[i] G (+ T_i)."""
    from helpers import affine_to_wire
    from smallorder import point_of_order
    k_eff = list(ks[:n])
    tors = {}
    kinds = ["zero", "neg_eq", "neg_ne", "repeat"]
    if curve in (BLS12_381_G1, BLS12_381_G2):
        kinds += ["torsion", "torsion"]
    targets = rnd.sample(range(n), min(n // 2, rnd.randrange(1, 9)))
    for j in targets:
        kind = rnd.choice(kinds)
        i = rnd.randrange(n)
        if kind == "zero":
            rows[j] = np.frombuffer(affine_to_wire(curve, Pt.ZERO.toAffine()), dtype=np.uint8) if curve == ED25519 else 0
            k_eff[j] = 0
            tors.pop(j, None)
        elif kind in ("neg_eq", "neg_ne", "repeat"):
            if i == j:
                continue
            rows[j] = rows[i] if kind == "repeat" else _neg_row(curve, rows[i])
            sign = 1 if kind == "repeat" else -1
            k_eff[j] = sign * k_eff[i]
            if i in tors:
                T, q = tors[i]
                tors[j] = (T if sign == 1 else T.negate(), q)
            else:
                tors.pop(j, None)
            if kind == "neg_eq":
                sc[j] = sc[i]                  # s P + s (-P): the pair meets in every bucket of every window and cancels there
        else:
            q = rnd.choice((3, 11) if curve == BLS12_381_G1 else (13, 23))
            if q == 3:
                from smallorder import g1_order3_points
                T = g1_order3_points()[rnd.randrange(2)]
            else:
                T = point_of_order("g1" if curve == BLS12_381_G1 else "g2", q)
                r = rnd.randrange(1, q)
                from smallorder import naive_mul
                T = naive_mul(T, r)
            Pm = Pt.BASE.multiplyUnsafe(k_eff[j] % Pt.Fn.ORDER).add(T) if j not in tors else None
            if Pm is None:
                continue
            rows[j] = np.frombuffer(affine_to_wire(curve, Pm.toAffine()), dtype=np.uint8)
            tors[j] = (T, q)
    return k_eff, tors


@pytest.mark.parametrize("curve", [BLS12_381_G1, BLS12_381_G2, SECP256K1, ED25519])
def test_randomised_msm_shapes_and_window_widths(curve):
    eng = get_engine()
    dev = torch.device("cuda", 0)
    Pt = ORACLE_CURVE[curve]
    order = Pt.Fn.ORDER
    rnd = random.Random(0xF0220 + curve)
    nmax = NMAX[curve]
    pts0, ks = bench.gen_points(eng, curve, Pt, nmax, 0x51ED + curve, 0x1F3, dev, None)
    pts0_h = pts0.cpu().numpy()
    n_mutated = n_torsion = 0
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
            mutated = (trial // 4) % 2 == 1 and n >= 4      # blocks of four trials: every entry point below sees mutated sets
            pts, pts_h, k_eff, tors = pts0, pts0_h, ks, {}
            if mutated:
                rows = pts0_h[:n].copy()
                k_eff, tors = _mutate(rnd, curve, Pt, n, ks, sc, rows)
                pts_h = rows
                pts = torch.from_numpy(rows).to(dev)
                n_mutated += 1
                n_torsion += len(tors)
            exp = Pt.BASE.multiplyUnsafe(sum(k * s for k, s in zip(k_eff, sc)) % order)
            for j, (T, q) in tors.items():
                from smallorder import naive_mul
                exp = exp.add(naive_mul(T, sc[j] % q))
            want, want_inf = exp.toAffine(), exp.is0()
            tag = (trial, n, shape, c, "mutated" if mutated else "plain")
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
                    # the reference's subgroup test on every member: the first one outside the subgroup, or -1 (then the endomorphism plan)
                    assert rs.verify_subgroup() == (min(tors) if tors else -1), tag
                    got, inf = rs.msm_dev(d.data_ptr())
                    assert wire_to_affine(curve, got) == want and inf == want_inf, tag + ("endo" if not tors else "generic after a failed subgroup check",)
                rs.free()
            done += 1
    finally:
        if old is None:
            os.environ.pop("NCG_MSM_C", None)
        else:
            os.environ["NCG_MSM_C"] = old
    assert done == 72 and n_mutated >= 30
    if curve in (BLS12_381_G1, BLS12_381_G2):
        assert n_torsion >= 10
