"""CPU execution of the kernels' shared templates (field arithmetic, group law, GLV split,
window recoding, per-lane ladder) against the oracle - catches logic errors without a GPU."""
import pytest

import hosttest
from helpers import ORACLE_CURVE, points_to_wire, scalars_to_wire, wire_to_affine
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2, SECP256K1
from oracle.curves import BLS_P, BLS_R, ED25519_P, SECP256K1_N, SECP256K1_P, Secp256k1, makeRng

FIELDS = [(0, SECP256K1_P, 32), (1, ED25519_P, 32), (2, BLS_P, 48)]


@pytest.mark.parametrize("fid,p,nb", FIELDS)
def test_field_ops_match_bigint(fid, p, nb):
    rng = makeRng(0xF00D + fid)
    vals = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (1 << (8 * nb - 8)) % p] + [rng.rndBelow(p) for _ in range(40)]
    for i, a in enumerate(vals):
        b = vals[(i * 7 + 3) % len(vals)]
        assert hosttest.field_op(fid, 0, a, b, nb) == a * b % p
        assert hosttest.field_op(fid, 1, a, b, nb) == a * a % p
        assert hosttest.field_op(fid, 2, a, b, nb) == (a + b) % p
        assert hosttest.field_op(fid, 3, a, b, nb) == (a - b) % p
        assert hosttest.field_op(fid, 4, a, b, nb) == (-a) % p
    for a in vals[:12]:
        assert hosttest.field_op(fid, 5, a, 0, nb) == (pow(a, -1, p) if a else 0)


def test_glv_split_is_lattice_exact_and_short():
    """Device GLV (reciprocal-multiply rounding) vs weierstrass.ts:121-148 semantics."""
    n = SECP256K1_N
    lam = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
    rng = makeRng(0x61C)
    ks = [0, 1, 2, n - 1, n - 2, n // 2, n // 2 + 1, lam, n - lam, 1 << 128, (1 << 256) - 1, n, n + 1]
    ks += [rng.rndBelow(n) for _ in range(3000)]
    for k in ks:
        k1neg, k1, k2neg, k2 = hosttest.glv_split(k)
        assert k1 < (1 << 128) + (1 << 100) and k2 < (1 << 128) + (1 << 100)
        s1 = -k1 if k1neg else k1
        s2 = -k2 if k2neg else k2
        assert (s1 + lam * s2 - k) % n == 0


def _check(curve, pts, ks):
    out, inf = hosttest.mul_var(curve, points_to_wire(curve, pts), scalars_to_wire(ks))
    Pt = ORACLE_CURVE[curve]
    for i, (p, k) in enumerate(zip(pts, ks)):
        exp = p.multiplyUnsafe(k).toAffine() if k < Pt.Fn.ORDER else None
        if exp is None:
            continue
        assert wire_to_affine(curve, out[i]) == exp, (i, hex(k))
        assert bool(inf[i]) == (exp == Pt.ZERO.toAffine())


def test_lane_ladder_secp256k1():
    n = SECP256K1_N
    rng = makeRng(0x1A0E)
    lam = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
    ks = [0, 1, 2, 3, 4, n - 1, n - 2, 1 << 128, (1 << 128) - 1, lam, lam + 1, n - lam, 15, 16, 17]
    ks += [rng.rndBelow(n) for _ in range(60)]
    pts = [Secp256k1.BASE.multiplyUnsafe(rng.rndBelow(n - 1) + 1) for _ in ks]
    pts[3] = Secp256k1.ZERO
    _check(SECP256K1, pts, ks)


@pytest.mark.parametrize("curve,cnt", [(BLS12_381_G1, 12), (BLS12_381_G2, 6)])
def test_lane_ladder_bls(curve, cnt):
    Pt = ORACLE_CURVE[curve]
    rng = makeRng(0xB0B + curve)
    ks = [0, 1, 2, BLS_R - 1, 7] + [rng.rndBelow(BLS_R) for _ in range(cnt)]
    pts = [Pt.BASE.multiplyUnsafe(rng.rndBelow(BLS_R - 1) + 1) for _ in ks]
    pts[4] = Pt.ZERO
    _check(curve, pts, ks)


def _ed_cases():
    """(sig, msg, pk) triples: valid KATs, corrupted copies, ZIP-215 and edge-case vectors."""
    from helpers import load_golden
    cases = []
    for row in load_golden("ed25519_vectors.json")[:40]:
        pk, msg, sig = (bytes.fromhex(row[k]) for k in ("pk", "msg", "sig"))
        cases.append((sig, msg, pk))
        bad = bytearray(sig); bad[5] ^= 4
        cases.append((bytes(bad), msg, pk))
        bad = bytearray(sig); bad[40] ^= 1
        cases.append((bytes(bad), msg, pk))
        cases.append((sig, msg + b"!", pk))
    for v in load_golden("ed25519_zip215.json"):
        cases.append((bytes.fromhex(v["sig_bytes"]), b"Zcash", bytes.fromhex(v["vk_bytes"])))
    for v in load_golden("ed25519_edge_cases.json"):
        cases.append((bytes.fromhex(v["signature"]), bytes.fromhex(v["message"]), bytes.fromhex(v["pub_key"])))
    # s >= L and s = L-1 style boundaries
    from oracle.curves import ED25519_L
    sig0, msg0, pk0 = cases[0]
    for s_val in (ED25519_L, ED25519_L + 1, (1 << 256) - 1, ED25519_L - 1, 0):
        cases.append((sig0[:32] + int(s_val).to_bytes(32, "little"), msg0, pk0))
    return cases


def test_ed25519_verify_lane_matches_oracle_both_modes():
    from oracle.curves import Ed25519
    from oracle.edwards import eddsa_hash_k, eddsa_verify
    for sig, msg, pk in _ed_cases():
        k = eddsa_hash_k(Ed25519.Fn, sig[:32], pk, msg)
        for zip215 in (True, False):
            exp = eddsa_verify(Ed25519, sig, msg, pk, zip215=zip215)
            assert hosttest.ed25519_verify(sig, pk, k, zip215) == exp, (sig.hex(), pk.hex(), zip215)


def _ed_points_and_scalars():
    from oracle.curves import ED25519_L, Ed25519
    rng = makeRng(0xED25)
    t8 = Ed25519.fromBytes(bytes.fromhex("c7176a703d4dd84fba3c0b760d10670f2a2053fa2c39ccc64ec7fd7792ac03fa"), True)
    ks = [0, 1, 2, 3, 7, 8, 9, ED25519_L - 1, ED25519_L - 2, 1 << 252, (1 << 252) + 1] + [rng.rndBelow(ED25519_L) for _ in range(20)]
    pts = [Ed25519.BASE.multiplyUnsafe(rng.rndBelow(ED25519_L - 1) + 1) for _ in ks]
    pts[2] = Ed25519.ZERO
    pts[5] = t8                                   # small-order point
    pts[6] = Ed25519.BASE.add(t8)                 # torsion component: exact integer multiple required
    pts[12] = pts[13].add(t8)
    return pts, ks


def test_ed25519_mul_var_lane_incl_torsion():
    """edwards.ts:571-577 semantics; torsioned points as in test/ed25519.test.ts:355-390."""
    from helpers import points_to_wire, scalars_to_wire, wire_to_affine
    from noble_curves_amd._native import ED25519
    from oracle.curves import Ed25519
    from oracle import curve as OC
    pts, ks = _ed_points_and_scalars()
    out, inf = hosttest.ed25519_mul_var(points_to_wire(ED25519, pts), scalars_to_wire(ks))
    for i, (p, k) in enumerate(zip(pts, ks)):
        exp = OC.naiveMul(Ed25519, p, k).toAffine()
        assert wire_to_affine(ED25519, out[i]) == exp, (i, hex(k))
        assert bool(inf[i]) == (exp == (0, 1))
