"""GPU parity on SMALL-ORDER Weierstrass inputs (SURVEY 8a gotcha 1).

`Point.fromAffine` accepts any curve point (src/abstract/weierstrass.ts:696-718) and the reference's
multiplyUnsafe / multiply / mulAddUnsafe / pippenger then run the complete RCB formulas (:793-880,
:900-944; src/abstract/curve.ts:863-905).  bls12-381's curves have cofactors with small prime
factors (G1: 3, 11; G2: 13, 23), so points of those orders - alone or added to a subgroup point -
are legal inputs.  Every entry point that multiplies is fed such points here and compared with
k*P computed by double-and-add on the oracle's complete add / double."""
import pytest

import smallorder
from helpers import ORACLE_CURVE, points_to_wire, scalars_to_wire, wire_to_affine
from noble_curves_amd import curve as shim
from noble_curves_amd import get_engine
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2
from oracle.curves import BLS_R, makeRng

pytestmark = pytest.mark.gpu

CURVES = [(BLS12_381_G1, "g1"), (BLS12_381_G2, "g2")]


def _shim_cls(curve):
    return shim.bls12_381_G1_Point if curve == BLS12_381_G1 else shim.bls12_381_G2_Point


def _to_shim(curve, P):
    return _shim_cls(curve).fromAffine(P.toAffine())


@pytest.mark.parametrize("curve,name", CURVES)
def test_multiply_unsafe_batch_small_order(curve, name):
    cases = smallorder.small_order_cases(name)
    ks = smallorder.small_order_scalars(4) + [15, 17, 31, 33, BLS_R]  # BLS_R: the isTorsionFree scalar
    pts, scal, exp = [], [], []
    for P, _ in cases:
        for k in ks:
            pts.append(P)
            scal.append(k)
            exp.append(smallorder.naive_mul(P, k).toAffine())
    # ordinary subgroup points in the same waves (the complete fallback must not disturb them)
    Pt = ORACLE_CURVE[curve]
    rng = makeRng(0xD15EA5E)
    for _ in range(24):
        P = Pt.BASE.multiplyUnsafe(rng.rndBelow(BLS_R - 1) + 1)
        k = rng.rndBelow(BLS_R)
        pts.append(P)
        scal.append(k)
        exp.append(P.multiplyUnsafe(k).toAffine())
    out, inf = get_engine().mul_var_batch(curve, points_to_wire(curve, pts), scalars_to_wire(scal))
    zero = Pt.ZERO.toAffine()
    for i, e in enumerate(exp):
        assert wire_to_affine(curve, out[i]) == e, (i, hex(scal[i]))
        assert bool(inf[i]) == (e == zero), (i, hex(scal[i]))


@pytest.mark.parametrize("curve,name", CURVES)
def test_shim_entry_points_small_order(curve, name):
    """multiplyBatch, mulAddUnsafeBatch, isTorsionFreeBatch (and clearCofactorBatch on G1)."""
    c = _shim_cls(curve)
    Pt = ORACLE_CURVE[curve]
    cases = smallorder.small_order_cases(name)
    ps = [_to_shim(curve, P) for P, _ in cases]
    ks = [3, 5, 11, 13, 23, 7, 1234567][:len(ps)]
    got = shim.multiplyBatch(c, ps, ks)
    for g, (P, _), k in zip(got, cases, ks):
        assert g.toAffine() == smallorder.naive_mul(P, k).toAffine()
    # a*P + b*Q with small-order P and Q, incl. a*P = -b*Q (sum ZERO) and a*P = b*Q (doubling)
    qs = [_to_shim(curve, cases[(i + 1) % len(cases)][0]) for i in range(len(cases))]
    a_s = [2, 3, 4, 5, 6, 7, BLS_R - 1][:len(ps)]
    b_s = [1, 0, 9, 10, 11, 12, 2][:len(ps)]
    ps2, qs2 = ps + [ps[0], ps[0]], qs + [ps[0], ps[0]]
    a2, b2 = a_s + [2, 1], b_s + [1, 1]                       # 2P + P = O (order 3 on G1), P + P
    got = shim.mulAddUnsafeBatch(c, ps2, a2, qs2, b2)
    src = [P for P, _ in cases]
    srcq = [cases[(i + 1) % len(cases)][0] for i in range(len(cases))]
    src2, srcq2 = src + [src[0], src[0]], srcq + [src[0], src[0]]
    for g, P, a, Q, b in zip(got, src2, a2, srcq2, b2):
        e = smallorder.naive_mul(P, a).add(smallorder.naive_mul(Q, b))
        assert g.toAffine() == e.toAffine()
    # subgroup membership: small-order and mixed points are NOT torsion free; subgroup points are
    sub = [Pt.BASE.multiplyUnsafe(k) for k in (1, 2, 0xDEADBEEF)]
    flags = shim.isTorsionFreeBatch(c, ps + [_to_shim(curve, s) for s in sub])
    assert flags == [False] * len(ps) + [True] * len(sub)
    if curve == BLS12_381_G1:
        h_eff = 0xD201000000010000 + 1
        got = shim.clearCofactorBatch(c, ps)
        for g, (P, _) in zip(got, cases):
            assert g.toAffine() == smallorder.naive_mul(P, h_eff).toAffine()


@pytest.mark.parametrize("curve,name", CURVES)
def test_pippenger_small_order(curve, name):
    """MSM over small-order, mixed and subgroup points with repeated points / scalars: buckets see
    P + P, P + (-P) and sums that pass through ZERO (curve.ts:863-905 on complete adds)."""
    c = _shim_cls(curve)
    Pt = ORACLE_CURVE[curve]
    cases = [P for P, _ in smallorder.small_order_cases(name)]
    rng = makeRng(0x51AB + curve)
    pts, ks = [], []
    for rep in range(6):
        for P in cases:
            pts.append(P)
            ks.append([1, 2, 3, rng.rndBelow(BLS_R), rng.rndBelow(1 << 16), BLS_R - 1][rep])
    for i in range(40):
        pts.append(Pt.BASE.multiplyUnsafe(i + 1))
        ks.append(rng.rndBelow(BLS_R))
    exp = Pt.ZERO
    for P, k in zip(pts, ks):
        exp = exp.add(smallorder.naive_mul(P, k))
    got = shim.pippenger(c, [_to_shim(curve, P) for P in pts], ks)
    assert got.toAffine() == exp.toAffine()
    # all-small-order MSM whose sum is ZERO: k*T + (q-k)*T
    T, q = smallorder.small_order_cases(name)[-4 if name == "g1" else -5]
    assert q is not None
    got = shim.pippenger(c, [_to_shim(curve, T)] * 2, [2, q - 2])
    assert got.is0()
