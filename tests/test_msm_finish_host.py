"""Host finish of the MSM (csrc/msm_finish.hpp): Horner over the grouped window sums + toAffine, i.e. the reference's
`sum = sum.add(resI); sum = sum.double() x c` chain (src/abstract/curve.ts:901-902) and weierstrass.ts:951-969.
bls12-381 runs the 64-bit-limb Jacobian form (csrc/bls_host64.hpp); variant 0 is the device templates compiled for
the host.  Both must reproduce the oracle's group element bit for bit - including infinity entries and the
exceptional additions P + P / P + (-P), which the Horner chain meets when window sums repeat."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hosttest  # noqa: E402
from helpers import ORACLE_CURVE, affine_to_wire  # noqa: E402
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2, POINT_BYTES  # noqa: E402
from oracle.curves import BLS_P, makeRng  # noqa: E402

R29 = 1 << 406   # Montgomery radix of the device storage format (14 limbs of 29 bits)
G = 3            # MSM_GROUP


def fe29_words(v):
    m = v * R29 % BLS_P
    return [(m >> (29 * i)) & ((1 << 29) - 1) for i in range(14)]


def fe29_words_lazy(v, k):
    """the same residue as a lazily reduced value v + k p (still below 64 p, limbs below 2^29)"""
    m = v * R29 % BLS_P + k * BLS_P
    assert m < 64 * BLS_P
    w = [(m >> (29 * i)) & ((1 << 29) - 1) for i in range(13)]
    return w + [m >> (29 * 13)]


def acc_words(curve, P, rng, lazy):
    """one XYZZ accumulator (X, Y, ZZ, ZZZ) = (x z^2, y z^3, z^2, z^3) for a random z; infinity = all zero"""
    fw = 14 if curve == BLS12_381_G1 else 28
    if P.is0():
        return [0] * (4 * fw)
    x, y = P.toAffine()
    if curve == BLS12_381_G1:
        z = rng.rndBelow(BLS_P - 1) + 1
        zz, zzz = z * z % BLS_P, z * z * z % BLS_P
        coords = [x * zz % BLS_P, y * zzz % BLS_P, zz, zzz]
        enc = (lambda v: fe29_words_lazy(v, rng.rndBelow(60))) if lazy else fe29_words
        return [w for c in coords for w in enc(c)]
    F2 = ORACLE_CURVE[curve].Fp
    z = (rng.rndBelow(BLS_P - 1) + 1, rng.rndBelow(BLS_P))
    zz = F2.sqr(z)
    zzz = F2.mul(zz, z)
    coords = [F2.mul(x, zz), F2.mul(y, zzz), zz, zzz]
    enc = (lambda v: fe29_words_lazy(v, rng.rndBelow(60))) if lazy else fe29_words
    return [w for c in coords for h in c for w in enc(h)]


def expected(Pt, V, c, nwin, ng):
    acc = Pt.ZERO
    for w in range(nwin - 1, -1, -1):
        for j in range(ng - 1, -1, -1):
            shift = c - G * j if j == ng - 1 else G
            for _ in range(shift):
                acc = acc.double()
            acc = acc.add(V[j][w])
    return acc


@pytest.mark.parametrize("curve", [BLS12_381_G1, BLS12_381_G2])
@pytest.mark.parametrize("c,nwin", [(2, 2), (4, 3), (7, 5), (16, 16), (13, 20), (5, 52)])
def test_host_finish_matches_oracle(curve, c, nwin):
    Pt = ORACLE_CURVE[curve]
    rng = makeRng(0xF1A15 + 31 * c + nwin)
    ng = (c - 1 + G - 1) // G if c >= 2 else 1
    base = Pt.BASE.multiplyUnsafe(rng.rndBelow(1 << 64) + 2)
    V = [[(Pt.ZERO if (j + w) % 5 == 4 else base.multiplyUnsafe(rng.rndBelow(1 << 40) + 1)) for w in range(nwin)] for j in range(ng)]
    if (c, nwin) == (2, 2):
        # ng = 1: acc = V[0][1], two doublings, + V[0][0]: force P + P, then P + (-P)
        cases = [V, [[V[0][1].double().double(), V[0][1]]], [[V[0][1].double().double().negate(), V[0][1]]]]
    else:
        cases = [V]
    for Vs in cases:
        exp = expected(Pt, Vs, c, nwin, ng)
        for lazy in (False, True):
            fin = [wd for j in range(ng) for w in range(nwin) for wd in acc_words(curve, Vs[j][w], rng, lazy)]
            # 2: the 64-bit form with the portable field product forced (1 = MULX / ADX where present); 3: the window sums built by
            # the helper threads (woken inside the call, repeated: claims race differently every time); 4: helper threads off
            for variant in (0, 1, 2, 3, 3, 3, 4):
                out, inf = hosttest.msm_finish(curve, c, nwin, np.array(fin, dtype=np.uint32), POINT_BYTES[curve], variant)
                assert inf == exp.is0(), (variant, lazy)
                want = bytes(POINT_BYTES[curve]) if exp.is0() else affine_to_wire(curve, exp.toAffine())
                assert out.tobytes() == want, (variant, lazy)


def test_plan_twin_matches_expectation():
    # the plan that every rank of a sharded MSM derives from n_max (csrc/msm_plan.hpp)
    p = hosttest.msm_plan(BLS12_381_G1, 1 << 20)
    assert p["c"] == 16 and p["nwin"] == 16 and p["ngroups"] == 5 and p["acc_words"] == 56
    assert hosttest.msm_shard_slot_bytes(BLS12_381_G1) % 256 == 0
    assert hosttest.msm_shard_slot_bytes(BLS12_381_G2) >= 2 * hosttest.msm_shard_slot_bytes(BLS12_381_G1) - 256


def test_host64_product_and_limb_import_at_the_edges():
    """bls_host64.hpp: the fused one-pass Montgomery product (R = 2^384) on 0, 1, p - 1, values with all-ones words, and
    the import of device elements (14 limbs of 29 bits, R = 2^406, lazily reduced up to 64 p) through the 22-bit Montgomery
    step - against big-int arithmetic (modular.ts:956 value of mul on Montgomery representatives)."""
    rng = makeRng(0x64F1)
    p = BLS_P
    rinv = pow(1 << 384, -1, p)
    print("bls_host64 product on MULX / ADX:", hosttest.h64_have_adx())
    edge = [0, 1, 2, p - 1, p - 2, (1 << 380) - 1, (1 << 381) - 1 - ((1 << 381) - 1) // p * 0]
    edge = [e % p for e in edge] + [((1 << 64) - 1) << (64 * i) for i in range(5)]
    vals = edge + [rng.rndBelow(p) for _ in range(120)]
    for a in vals:
        for b in (vals[:8] + vals[-6:]):
            want = (a % p) * (b % p) * rinv % p
            assert hosttest.h64_mul(a % p, b % p) == want
            assert hosttest.h64_mul(a % p, b % p, portable=True) == want
    for trial in range(200):
        x = rng.rndBelow(p) if trial > 2 else [0, 1, p - 1][trial]
        k = rng.rndBelow(64) if trial % 3 else 63
        m = x * R29 % p + k * p
        if m >= 64 * p:
            m -= p
        limbs = [(m >> (29 * i)) & ((1 << 29) - 1) for i in range(13)] + [m >> (29 * 13)]
        assert hosttest.h64_from_fe29(limbs) == x * (1 << 384) % p
