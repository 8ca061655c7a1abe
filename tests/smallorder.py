"""Small-order points of the bls12-381 curves E(Fp) and E'(Fp2), built with the oracle.

The reference's `Point.fromAffine` / `new Point` only range-check coordinates
(src/abstract/weierstrass.ts:696-718); `multiplyUnsafe`, `multiply`, `mulAddUnsafe` and `pippenger`
then work on ANY curve point through the complete RCB formulas (:793-880).  G1's curve has
cofactor 3 * 11^2 * 10177^2 * 859267^2 * 52437899^2, G2's contains 13^2 * 23^2 - so points of
order 3, 11, 13, 23 are legal inputs, and they are exactly where incomplete additions break.
Test infrastructure only.
"""
import functools

from oracle.curves import BLS_P, BLS_R, BlsG1, BlsG2, Fp2_bls, Fp_bls, bls_G1_CURVE, bls_G2_CURVE, makeRng


def naive_mul(P, k):
    """k*P for any integer k >= 0 by double-and-add on the oracle's complete add / double."""
    R = type(P).ZERO
    Q = P
    while k:
        if k & 1:
            R = R.add(Q)
        Q = Q.double()
        k >>= 1
    return R


def _random_point_g1(rng):
    while True:
        x = rng.rndBelow(BLS_P)
        y2 = (x * x * x + 4) % BLS_P
        y = pow(y2, (BLS_P + 1) // 4, BLS_P)
        if y * y % BLS_P == y2:
            return BlsG1.fromAffine((x, y))


def _random_point_g2(rng):
    F = Fp2_bls
    while True:
        x = (rng.rndBelow(BLS_P), rng.rndBelow(BLS_P))
        y2 = F.add(F.mul(F.sqr(x), x), (4, 4))
        try:
            y = F.sqrt(y2)
        except Exception:
            continue
        if y is not None and F.eql(F.sqr(y), y2):
            return BlsG2.fromAffine((x, y))


@functools.lru_cache(maxsize=None)
def point_of_order(curve_name, q):
    """A point of exact prime order q on E(Fp) ('g1') or E'(Fp2) ('g2')."""
    if curve_name == "g1":
        full = bls_G1_CURVE["h"] * BLS_R
        gen = _random_point_g1
    else:
        full = bls_G2_CURVE["h"] * BLS_R
        gen = _random_point_g2
    assert full % q == 0
    cof = full
    while cof % q == 0:      # strip the whole q-part (the q-Sylow subgroup may be Z_q x Z_q)
        cof //= q
    rng = makeRng(0x5A11 + q)
    while True:
        T = naive_mul(gen(rng), cof)
        if T.is0():
            continue
        while True:
            U = naive_mul(T, q)
            if U.is0():
                return T
            T = U


def g1_order3_points():
    """(0, 2) and (0, p-2): the x = 0 points of y^2 = x^3 + 4, order 3."""
    return [BlsG1.fromAffine((0, 2)), BlsG1.fromAffine((0, BLS_P - 2))]


def small_order_cases(curve_name):
    """[(point, order)] incl. small-order + subgroup mixtures (order = None: large)."""
    if curve_name == "g1":
        Pt, orders = BlsG1, (11,)
        pts = [(p, 3) for p in g1_order3_points()]
    else:
        Pt, orders = BlsG2, (13, 23)
        pts = []
    for q in orders:
        T = point_of_order(curve_name, q)
        pts.append((T, q))
        pts.append((naive_mul(T, q - 2), q))
    G7 = Pt.BASE.multiplyUnsafe(7)
    mixed = [(p.add(G7), None) for p, _ in pts[:3]]
    return pts + mixed


def small_order_scalars(W=5):
    ks = [0, 1, 2, 3, 4, 5, 7, 10, 11, 12, 13, 14, 22, 23, 24, 26, (1 << W) - 1, (1 << W) + 1, 1234567,
          BLS_R - 1, BLS_R - 2, (BLS_R - 1) // 3]
    rng = makeRng(0x5CA1A5)
    return ks + [rng.rndBelow(BLS_R) for _ in range(6)]
