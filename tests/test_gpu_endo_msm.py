"""pippenger on point sets verified to lie in the prime-order subgroup (include/ncg.h "resident point sets":
ncg_points_verify_subgroup / sets decoded by ncg_points_from_encoded): the scalars are split along the curve
endomorphism (csrc/endo.hpp) - the result must be the reference's pippenger (curve.ts:863-905) bit for bit,
and a set holding a point OUTSIDE the subgroup must never take that path."""
import numpy as np
import pytest

from noble_curves_amd import curve as G
from oracle import curve as OC
from oracle.curves import BLS_R, BlsG1, BlsG2, makeRng
from smallorder import small_order_cases

pytestmark = pytest.mark.gpu

Z = 0xD201000000010000
CASES = [(G.bls12_381_G1_Point, BlsG1), (G.bls12_381_G2_Point, BlsG2)]


def edge_scalars():
    x2 = Z * Z
    vals = [0, 1, 2, BLS_R - 1, BLS_R - 2, Z, Z - 1, Z + 1, Z // 2, Z // 2 + 1, x2, x2 - 1, x2 + 1, x2 // 2, x2 // 2 + 1,
            Z ** 3, Z ** 3 - 1, Z ** 3 + Z // 2 + 1, (x2 // 2 + 1) * x2 + x2 // 2 + 1, (x2 - 1) * x2 + x2 - 1, BLS_R // 2]
    vals += [(a * Z ** 3 + b * Z ** 2 + c * Z + d) % BLS_R for a in (0, Z // 2 + 1, Z - 1) for b in (0, Z // 2, Z - 1)
             for c in (Z // 2 + 1, Z - 1) for d in (0, Z // 2 + 1, Z - 1)]
    return [v % BLS_R for v in vals]


@pytest.mark.parametrize("c,Pt", CASES)
def test_verified_set_matches_oracle(c, Pt):
    rng = makeRng(0xE1D0 + c.CURVE_ID)
    edges = edge_scalars()
    n = len(edges) + 40
    opts = [Pt.BASE.multiplyUnsafe(rng.rndBelow(BLS_R - 1) + 1) for _ in range(n)]
    opts[3] = Pt.ZERO
    opts[11] = opts[10]                     # P = Q and P = -Q inside buckets
    opts[13] = opts[12].negate()
    pts = [c.fromAffine(p.toAffine()) for p in opts]
    plain = G.uploadPoints(c, pts)
    fast = G.uploadPoints(c, pts, checkSubgroup=True)
    assert not plain.inSubgroup and fast.inSubgroup
    scalar_sets = [edges + [rng.rndBelow(BLS_R) for _ in range(n - len(edges))],
                   [rng.rndBelow(BLS_R) for _ in range(n)],
                   [0] * n, [1] * n, [BLS_R - 1] * n,
                   [edges[i % len(edges)] for i in range(7, 7 + n)]]
    scalar_sets[1][10], scalar_sets[1][11] = 5, BLS_R - 5          # k P + (r - k) P = O
    scalar_sets[1][12], scalar_sets[1][13] = 77, 77                # k P + k (-P) = O
    for sc in scalar_sets:
        exp = OC.pippenger(Pt, opts, sc).toAffine()
        assert G.pippenger(c, fast, sc).toAffine() == exp
        assert G.pippenger(c, plain, sc).toAffine() == exp
    with pytest.raises(ValueError, match="invalid scalar at index 2"):
        G.pippenger(c, fast, [1, 2, BLS_R] + [0] * (n - 3))
    plain.free(); fast.free()


@pytest.mark.parametrize("c,Pt", CASES)
def test_decoded_set_takes_the_fast_path(c, Pt):
    rng = makeRng(0xDEC0 + c.CURVE_ID)
    n = 300
    opts = [Pt.BASE.multiplyUnsafe(rng.rndBelow(BLS_R - 1) + 1) for _ in range(n)]
    pts = [c.fromAffine(p.toAffine()) for p in opts]
    es = G.uploadEncoded(c, G.toBytesBatch(c, pts))
    assert es.inSubgroup                      # fromBytes ran the subgroup test on every point
    for _ in range(2):
        sc = [rng.rndBelow(BLS_R) for _ in range(n)]
        assert G.pippenger(c, es, sc).toAffine() == OC.pippenger(Pt, opts, sc).toAffine()
    es.free()


@pytest.mark.parametrize("c,Pt", CASES)
def test_set_with_a_point_outside_the_subgroup_keeps_the_generic_path(c, Pt):
    """pippenger accepts any curve point (SURVEY 8a gotcha 1): small-order and mixed-order points make the
    verification fail (naming the first offender) and the MSM stays exact through the generic path."""
    rng = makeRng(0x0BAD + c.CURVE_ID)
    tp = small_order_cases("g1" if Pt is BlsG1 else "g2")[0][0]   # a small-order point of the curve
    opts = [Pt.BASE.multiplyUnsafe(rng.rndBelow(BLS_R - 1) + 1) for _ in range(20)]
    opts[6] = opts[6].add(tp)                 # subgroup + torsion component
    opts[15] = tp
    pts = [c.fromAffine(p.toAffine()) for p in opts]
    s = G.uploadPoints(c, pts, checkSubgroup=True)
    assert not s.inSubgroup and s.resident.verify_subgroup() == 6
    sc = [rng.rndBelow(BLS_R) for _ in range(20)]
    assert G.pippenger(c, s, sc).toAffine() == OC.pippenger(Pt, opts, sc).toAffine()
    s.free()


@pytest.mark.parametrize("c,Pt,log2n", [(G.bls12_381_G1_Point, BlsG1, 16), (G.bls12_381_G2_Point, BlsG2, 14)])
def test_large_verified_set_against_generic_path(c, Pt, log2n):
    """Sizes where the window plan is the production one (c = 16 region): the endomorphism path against the
    generic path on the same device arrays, and a linearity identity sum k_i (a_i G) = (sum k_i a_i) G."""
    from noble_curves_amd import get_engine
    eng = get_engine()
    n = 1 << log2n
    rng = makeRng(0xB16 + log2n)
    a = [rng.rndBelow(BLS_R - 1) + 1 for _ in range(n)]
    pts = G.multiplyBaseBatch(c, a)
    sc = [rng.rndBelow(BLS_R) for _ in range(n)]
    fast = G.uploadPoints(c, pts, checkSubgroup=True)
    plain = G.uploadPoints(c, pts)
    assert fast.inSubgroup
    r_fast, r_plain = G.pippenger(c, fast, sc), G.pippenger(c, plain, sc)
    assert r_fast.toAffine() == r_plain.toAffine()
    assert r_fast.toAffine() == Pt.BASE.multiplyUnsafe(sum(k * x for k, x in zip(sc, a)) % BLS_R).toAffine()
    fast.free(); plain.free()


@pytest.mark.parametrize("c,Pt", CASES)
def test_verified_set_batch_multiply_uses_the_endomorphism_ladder(c, Pt):
    """multiplyUnsafeBatch on a verified set (ncg_mul_var_batch_resident -> mulvar_endo.hip: G1 two streams along
    phi, G2 four streams along psi) against the oracle and against the generic ladder on the same set: edge
    scalars of the split (multiples of z^e, half-way values, r - 1), ZERO, full waves."""
    rng = makeRng(0x61F + c.CURVE_ID)
    edges = edge_scalars()
    n = 3000 if c is G.bls12_381_G1_Point else 1100
    a = [rng.rndBelow(BLS_R - 1) + 1 for _ in range(n)]
    pts = G.multiplyBaseBatch(c, a)
    pts[7] = c.ZERO
    sc = (edges + [rng.rndBelow(BLS_R) for _ in range(n)])[:n]
    fast, plain = G.uploadPoints(c, pts, checkSubgroup=True), G.uploadPoints(c, pts)
    assert fast.inSubgroup and not plain.inSubgroup
    rf, rp = G.multiplyUnsafeBatch(c, fast, sc), G.multiplyUnsafeBatch(c, plain, sc)
    assert [p.toAffine() for p in rf] == [p.toAffine() for p in rp]
    for i in list(range(len(edges))) + [n - 1, n - 2]:
        exp = Pt.ZERO if i == 7 else Pt.BASE.multiplyUnsafe(a[i] * sc[i] % BLS_R)
        assert rf[i].toAffine() == exp.toAffine(), i
    fast.free(); plain.free()


@pytest.mark.parametrize("c,Pt", CASES)
def test_verified_set_batch_multiply_out_of_range_scalars_at_the_c_abi(c, Pt):
    """The shims reject k >= r before crossing (weierstrass.ts:915-928); the C ABI takes any 256-bit value like the
    generic ladder does: the endomorphism ladders must hand those lanes to the complete ladder, not mis-split them."""
    import numpy as np
    rng = makeRng(0xAB1 + c.CURVE_ID)
    ks = [BLS_R, BLS_R + 1, BLS_R + Z ** 3, 2 * BLS_R - 1, 2 ** 255, 2 ** 256 - 1, 2 ** 256 - BLS_R, 5, 0]
    ks += [rng.rndBelow(2 ** 256) for _ in range(130 - len(ks))]
    a = [rng.rndBelow(BLS_R - 1) + 1 for _ in range(len(ks))]
    pts = G.multiplyBaseBatch(c, a)
    fast, plain = G.uploadPoints(c, pts, checkSubgroup=True), G.uploadPoints(c, pts)
    wire = np.frombuffer(b"".join(k.to_bytes(32, "little") for k in ks), np.uint8).reshape(-1, 32)
    (of, inf_f), (op, inf_p) = fast.resident.mul_var_batch(wire), plain.resident.mul_var_batch(wire)
    assert np.array_equal(of, op) and np.array_equal(inf_f, inf_p)
    for i in (0, 1, 2, 5, 8, 20):
        exp = Pt.BASE.multiplyUnsafe(a[i] * ks[i] % BLS_R)
        got = c._from_wire(of[i], bool(inf_f[i]))
        assert got.toAffine() == exp.toAffine(), i
    fast.free(); plain.free()


@pytest.mark.parametrize("c,Pt", CASES)
def test_tiny_verified_sets(c, Pt):
    """Window plans at the small end (c clamps to 3, one sort chunk): 1-, 2-, 5- and 33-point verified sets."""
    rng = makeRng(0x7171 + c.CURVE_ID)
    for n in (1, 2, 5, 33):
        opts = [Pt.BASE.multiplyUnsafe(rng.rndBelow(BLS_R - 1) + 1) for _ in range(n)]
        s = G.uploadPoints(c, [c.fromAffine(p.toAffine()) for p in opts], checkSubgroup=True)
        assert s.inSubgroup
        for sc in ([BLS_R - 1] * n, [rng.rndBelow(BLS_R) for _ in range(n)], [0] * n):
            assert G.pippenger(c, s, sc).toAffine() == OC.pippenger(Pt, opts, sc).toAffine()
        s.free()
