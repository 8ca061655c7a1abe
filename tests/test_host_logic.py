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


def test_ed25519_halved_scalars():
    """ed_halve.hpp: u k == v (mod L) with 0 < |u| < 2^126, 0 <= v < 2^127 and w = |u| s mod L - edge challenges
    (0, 1, around 2^127, L - 1, any 256-bit value, the golden-ratio worst case of the Euclidean algorithm) and
    random ones."""
    from decimal import Decimal, getcontext
    from oracle.curves import ED25519_L as L, makeRng
    getcontext().prec = 120
    gold = int(Decimal(L) * (Decimal(5).sqrt() - 1) / 2)
    rng = makeRng(0xED)
    ks = [0, 1, 2, 3, 2 ** 127 - 1, 2 ** 127, 2 ** 127 + 1, 2 ** 126, L - 1, L - 2, L // 2, (L + 1) // 2, L // 3, gold, L - gold,
          L, L + 1, 2 ** 255, 2 ** 256 - 1, 2 ** 252, 2 ** 251, 2 ** 200 + 1]
    ks += [rng.rndBelow(L) for _ in range(300)]
    for i, k in enumerate(ks):
        s = [0, 1, L - 1, 2 ** 256 - 1][i] if i < 4 else rng.rndBelow(L)
        u, v, w = hosttest.ed_halve(k, s)
        assert 0 < abs(u) < 2 ** 126 and 0 <= v < 2 ** 127, (k, u, v)
        assert (u * k - v) % L == 0, k
        assert w == abs(u) * s % L, (k, s)


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


def decode_cases_secp():
    """(33-byte encoding, expected point or None): reference isPoint vectors + structured rejects."""
    from helpers import load_golden
    from oracle.curves import Secp256k1
    from oracle.weierstrass import sec1_decode
    cases = []
    encs = [bytes.fromhex(v["P"]) for v in load_golden("secp256k1_ispoint_compressed.json")]
    P = SECP256K1_P
    encs += [b"\x02" + (P - 1).to_bytes(32, "big"), b"\x03" + P.to_bytes(32, "big"), b"\x02" + (P + 1).to_bytes(32, "big"),
             b"\x04" + (1).to_bytes(32, "big"), b"\x00" + (1).to_bytes(32, "big"), b"\x02" + bytes(32), b"\x03" + b"\xff" * 32,
             b"\x02" + (5).to_bytes(32, "big"), b"\x03" + (5).to_bytes(32, "big")]
    for e in encs:
        try:
            cases.append((e, sec1_decode(Secp256k1, e).toAffine()))
        except ValueError:
            cases.append((e, None))
    return cases


def decode_cases_g1():
    from helpers import load_golden
    from oracle.curves import BlsG1, BLS_P
    from oracle.weierstrass import bls_g1_decode_compressed, bls_g1_encode_compressed
    encs = [bytes.fromhex(r) for r in load_golden("bls12_381_g1_compressed.json")[:24]]
    rng = makeRng(0xDEC0DE)
    g = bytearray(encs[3])
    flip = bytearray(g); flip[0] ^= 0x20                        # other root: still a subgroup point
    nocomp = bytearray(g); nocomp[0] &= 0x7F                    # compression bit cleared
    inf_bad = bytes([0xC0]) + bytes(46) + b"\x01"               # infinity with payload
    inf_sort = bytes([0xE0]) + bytes(47)                        # invalid flag combination
    big = bytearray((BLS_P + 5).to_bytes(48, "big")); big[0] |= 0x80
    encs += [bytes(flip), bytes(nocomp), inf_bad, inf_sort, bytes(big), bytes([0xC0]) + bytes(47)]
    # random x: mostly non-residues or points outside the prime-order subgroup
    for _ in range(14):
        x = rng.rndBelow(BLS_P)
        b = bytearray(x.to_bytes(48, "big")); b[0] |= 0x80 | (0x20 if rng.rnd64() & 1 else 0)
        encs.append(bytes(b))
    cases = []
    for e in encs:
        try:
            p = bls_g1_decode_compressed(BlsG1, e)
            cases.append((e, p.toAffine(), p.is0()))
        except ValueError:
            cases.append((e, None, False))
    assert sum(1 for c in cases if c[1] is None) >= 10 and sum(1 for c in cases if c[1] is not None) >= 20
    return cases


def decode_cases_g2():
    from helpers import load_golden
    from oracle.curves import BlsG2, BLS_P
    from oracle.weierstrass import bls_g2_decode_compressed
    encs = [bytes.fromhex(r) for r in load_golden("bls12_381_g2_compressed.json")[:12]]
    rng = makeRng(0xDEC0DE2)
    g = bytearray(encs[3])
    flip = bytearray(g); flip[0] ^= 0x20
    nocomp = bytearray(g); nocomp[0] &= 0x7F
    inf_bad = bytes([0xC0]) + bytes(94) + b"\x01"
    inf_sort = bytes([0xE0]) + bytes(95)
    big1 = bytearray((BLS_P + 5).to_bytes(48, "big") + bytes(g[48:])); big1[0] |= 0x80
    big0 = bytearray(bytes(g[:48]) + BLS_P.to_bytes(48, "big"))
    encs += [bytes(flip), bytes(nocomp), inf_bad, inf_sort, bytes(big1), bytes(big0), bytes([0xC0]) + bytes(95)]
    # random x: non-squares, points outside the prime-order subgroup, and x with c1 = 0 / c0 = 0
    for k in range(14):
        x1 = 0 if k % 5 == 3 else rng.rndBelow(BLS_P)
        x0 = 0 if k % 5 == 4 else rng.rndBelow(BLS_P)
        b = bytearray(x1.to_bytes(48, "big") + x0.to_bytes(48, "big"))
        b[0] |= 0x80 | (0x20 if rng.rnd64() & 1 else 0)
        encs.append(bytes(b))
    cases = []
    for e in encs:
        try:
            p = bls_g2_decode_compressed(BlsG2, e)
            cases.append((e, p.toAffine(), p.is0()))
        except ValueError:
            cases.append((e, None, False))
    assert sum(1 for c in cases if c[1] is None) >= 10 and sum(1 for c in cases if c[1] is not None) >= 12
    return cases


def test_fp2_sqrt_lane_decides_like_the_oracle():
    """Fp2.sqrt (tower.ts:476-500): the single-power device variant finds a root exactly when the
    reference does, and its root is +-the reference's; covers c1 == 0 (both Legendre cases) and 0."""
    from oracle.curves import BlsG2, BLS_P
    F2 = BlsG2.Fp
    rng = makeRng(0x5152)
    vals = [(rng.rndBelow(BLS_P), rng.rndBelow(BLS_P)) for _ in range(10)]
    vals += [F2.sqr(v) for v in vals[:4]]
    vals += [(rng.rndBelow(BLS_P), 0) for _ in range(6)] + [(0, rng.rndBelow(BLS_P)) for _ in range(3)]
    vals += [(0, 0), (1, 0), (BLS_P - 1, 0), (4, 4)]
    n_sq = 0
    for v in vals:
        try:
            exp = F2.sqrt(v)
        except ValueError:
            exp = None
        ok, r = hosttest.fp2_sqrt(*v)
        assert ok == (exp is not None), v
        if exp is not None:
            n_sq += 1
            assert r in (exp, F2.neg(exp)), v
    assert 8 < n_sq < len(vals)


def test_decode_lanes_g2():
    """96-byte compressed decode + psi subgroup check, lane logic vs oracle."""
    import numpy as np
    from helpers import wire_to_affine
    from noble_curves_amd._native import BLS12_381_G2
    cg = decode_cases_g2()
    out, ok, inf = hosttest.decode_points(BLS12_381_G2, np.array([np.frombuffer(c[0], np.uint8) for c in cg]), 192)
    for i, (e, exp, is0) in enumerate(cg):
        assert ok[i] == (exp is not None), (i, e.hex())
        assert wire_to_affine(BLS12_381_G2, out[i]) == (exp if exp else ((0, 0), (0, 0))) and inf[i] == is0


def test_decode_lanes_secp256k1_g1_ed25519():
    """decompression + validity, lane logic on the CPU vs the oracle."""
    import numpy as np
    from helpers import load_golden, wire_to_affine
    from noble_curves_amd._native import ED25519
    from oracle.curves import Ed25519
    cs = decode_cases_secp()
    out, ok, _ = hosttest.decode_points(SECP256K1, np.array([np.frombuffer(c[0], np.uint8) for c in cs]), 64)
    for i, (e, exp) in enumerate(cs):
        assert ok[i] == (exp is not None), e.hex()
        assert wire_to_affine(SECP256K1, out[i]) == (exp if exp else (0, 0))
    cg = decode_cases_g1()
    out, ok, inf = hosttest.decode_points(BLS12_381_G1, np.array([np.frombuffer(c[0], np.uint8) for c in cg]), 96)
    for i, (e, exp, is0) in enumerate(cg):
        assert ok[i] == (exp is not None), e.hex()
        assert wire_to_affine(BLS12_381_G1, out[i]) == (exp if exp else (0, 0)) and inf[i] == is0
    encs = [bytes.fromhex(v["vk_bytes"]) for v in load_golden("ed25519_zip215.json")]
    encs += [bytes.fromhex(r["pk"]) for r in load_golden("ed25519_vectors.json")[:20]]
    for zip215 in (True, False):
        out, ok, _ = hosttest.decode_points(ED25519, np.array([np.frombuffer(e, np.uint8) for e in encs]), 64,
                                            1 if zip215 else 0)
        for i, e in enumerate(encs):
            try:
                exp = Ed25519.fromBytes(e, zip215).toAffine()
            except ValueError:
                exp = None
            assert ok[i] == (exp is not None), (e.hex(), zip215)
            if exp:
                assert wire_to_affine(ED25519, out[i]) == exp


def test_encode_lanes_all_curves():
    """Point.toBytes (compressed) lane logic vs the oracle encoders; decode(encode(P)) == P on the host lanes."""
    import numpy as np
    from helpers import load_golden, points_to_wire
    from noble_curves_amd._native import BLS12_381_G2, ED25519
    from oracle.curves import BlsG1, BlsG2, Ed25519, Secp256k1
    from oracle.weierstrass import bls_g1_encode_compressed, bls_g2_encode_compressed, sec1_encode
    rng = makeRng(0xE7C0DE)
    ks = [1, 2, 3] + [rng.rndBelow(1 << 64) + 1 for _ in range(9)]
    sp = [Secp256k1.BASE.multiplyUnsafe(k) for k in ks]
    enc, ok = hosttest.encode_points(SECP256K1, points_to_wire(SECP256K1, sp + [Secp256k1.ZERO]), 33)
    assert list(ok) == [True] * len(sp) + [False]
    assert [enc[i].tobytes() for i in range(len(sp))] == [sec1_encode(p) for p in sp]
    g1 = [BlsG1.BASE.multiplyUnsafe(k) for k in ks] + [BlsG1.ZERO]
    enc, ok = hosttest.encode_points(BLS12_381_G1, points_to_wire(BLS12_381_G1, g1), 48)
    assert ok.all() and [enc[i].tobytes() for i in range(len(g1))] == [bls_g1_encode_compressed(p) for p in g1]
    assert [enc[i].tobytes().hex() for i in range(3)] == load_golden("bls12_381_g1_compressed.json")[1:4]
    g2 = [BlsG2.BASE.multiplyUnsafe(k) for k in ks[:6]] + [BlsG2.ZERO]
    enc, ok = hosttest.encode_points(BLS12_381_G2, points_to_wire(BLS12_381_G2, g2), 96)
    assert ok.all() and [enc[i].tobytes() for i in range(len(g2))] == [bls_g2_encode_compressed(p) for p in g2]
    assert [enc[i].tobytes().hex() for i in range(3)] == load_golden("bls12_381_g2_compressed.json")[1:4]
    dec, dok, dinf = hosttest.decode_points(BLS12_381_G2, enc, 192)
    assert dok.all() and (dec == points_to_wire(BLS12_381_G2, g2)).all() and list(dinf) == [False] * 6 + [True]
    ed = [Ed25519.BASE.multiplyUnsafe(k) for k in ks] + [Ed25519.ZERO]
    enc, ok = hosttest.encode_points(ED25519, points_to_wire(ED25519, ed), 32)
    assert ok.all() and [enc[i].tobytes() for i in range(len(ed))] == [p.toBytes() for p in ed]


def test_ntt_lane_code_matches_oracle_all_orderings():
    """the device butterflies + table walk (ntt.hip, executed on the CPU) against
    the oracle FFT for every (inverse, brpInput, brpOutput) combination and the reference's KAT."""
    from helpers import load_golden
    from oracle.curves import Fr_bls
    from oracle.fft import FFT, RootsOfUnity
    roots = RootsOfUnity(Fr_bls, 7)
    f = FFT(roots, Fr_bls)
    kat = load_golden("fft_kat.json")
    assert hosttest.ntt(3, [int(x) for x in kat["basic_input"]], roots.omega(3), 0) == [int(x) for x in kat["basic_exp"]]
    rng = makeRng(0x177)
    for bits in (0, 1, 2, 5, 7):
        x = [rng.rndBelow(Fr_bls.ORDER) for _ in range(1 << bits)]
        if bits >= 2:
            x[0], x[1] = 0, Fr_bls.ORDER - 1
        for flags in range(8):
            inv, bi, bo = bool(flags & 1), bool(flags & 2), bool(flags & 4)
            exp = (f.inverse if inv else f.direct)(x, bi, bo)
            assert hosttest.ntt(bits, x, roots.omega(bits), flags) == exp, (bits, flags)


def _fr29_val(limbs):
    return sum(int(x) << (29 * i) for i, x in enumerate(limbs))


def _fr29_limbs(x):
    return [(x >> (29 * i)) & ((1 << 29) - 1) for i in range(8)] + [x >> 232]


def test_fr29_butterfly_arithmetic_at_the_bounds():
    """fr29.hpp (the NTT butterflies' radix-2^29 lazy form of Fr, modular.ts:940-982 values at the pass
    boundaries): Montgomery product with the left operand at the loosest limbs it admits (6 * 2^29, limb 8
    all ones) and the right one at r - 1 / all-ones limbs, the fold below 2^256 from 33 r, the final
    conditional subtraction and the word <-> limb conversions, against big-int arithmetic; the host
    twin counts every 64-bit column and 32-bit limb overflow (must be none)."""
    from oracle.curves import Fr_bls
    r = Fr_bls.ORDER
    M = (1 << 29) - 1
    rinv = pow(1 << 261, -1, r)
    rng = makeRng(0xF29)
    for trial in range(120):
        a = [rng.rndBelow(6 << 29) for _ in range(8)] + [rng.rndBelow(1 << 32)]
        w = _fr29_limbs(rng.rndBelow(r))
        if trial == 0:
            a = [(6 << 29) - 1] * 8 + [(1 << 32) - 1]
        if trial < 2:
            w = _fr29_limbs(r - 1)
        if trial == 2:
            a, w = [(6 << 29) - 1] * 8 + [(1 << 32) - 1], [M] * 8 + [(1 << 23) - 1]
        out, ovf = hosttest.fr29_op(0, a, w)
        assert ovf == 0
        assert _fr29_val(out) % r == _fr29_val(a) * _fr29_val(w) * rinv % r
        assert all(x <= M for x in out[:8]) and _fr29_val(out) < _fr29_val(a) * _fr29_val(w) // (1 << 261) + r + 1
    for trial in range(120):
        v = rng.rndBelow(33 * r) if trial else 33 * r - 1
        out, ovf = hosttest.fr29_op(4, _fr29_limbs(v))
        assert ovf == 0 and _fr29_val(out) % r == v % r and _fr29_val(out) < (1 << 256) and all(x <= M for x in out[:8])
        a = [rng.rndBelow(7 << 29) for _ in range(8)] + [rng.rndBelow(1 << 28)]     # loose limbs, value < 33 r
        out, ovf = hosttest.fr29_op(4, a)
        assert ovf == 0 and _fr29_val(out) % r == _fr29_val(a) % r and _fr29_val(out) < (1 << 256)
    for trial in range(60):
        v = [0, r - 1, r, 2 * r - 1][trial] if trial < 4 else rng.rndBelow(2 * r)
        out, _ = hosttest.fr29_op(5, _fr29_limbs(v))
        assert _fr29_val(out) == v % r
        out, _ = hosttest.fr29_op(6, [(v >> (32 * i)) & 0xFFFFFFFF for i in range(8)] + [0])
        assert _fr29_val(out) == v and all(x <= M for x in out[:8])
        out, _ = hosttest.fr29_op(7, _fr29_limbs(v))
        assert sum(x << (32 * i) for i, x in enumerate(out[:8])) == v
    # a - t through the 3 r bias, a + t, and the weak normalisation keep the value
    for trial in range(60):
        a = [rng.rndBelow(5 << 29) for _ in range(8)] + [rng.rndBelow(1 << 28)]
        t = _fr29_limbs(rng.rndBelow(29 * r // 10) if trial else (0x015BC8F4 << 232) + (1 << 232) - 1)  # limb 8 at BIAS[8]
        d, ovf = hosttest.fr29_op(2, a, t)
        assert ovf == 0 and _fr29_val(d) == _fr29_val(a) + 3 * r - _fr29_val(t)
        sm, ovf = hosttest.fr29_op(1, a, t)
        assert ovf == 0 and _fr29_val(sm) == _fr29_val(a) + _fr29_val(t)
        wk, ovf = hosttest.fr29_op(3, d)
        assert ovf == 0 and _fr29_val(wk) == _fr29_val(d) and all(x < (1 << 29) + 8 for x in wk[:8])


def test_ntt_multi_pass_schedules_and_full_size_passes():
    """The device pass schedule on the host twin with the passes shrunk so that small transforms run three and
    more passes (tiles with a contiguous run of 4, the fold of the bit reversal through the workspace, the
    1/N scale on the last pass), and with the real 10-stage passes at 2^10 / 2^11 on random and all-(r - 1)
    input (the largest lazy values a pass can build up) - every ordering against the oracle FFT
    (fft.ts:518-577), no column or limb overflow."""
    from oracle.curves import Fr_bls
    from oracle.fft import FFT, RootsOfUnity
    r = Fr_bls.ORDER
    roots = RootsOfUnity(Fr_bls, 11)
    f = FFT(roots, Fr_bls)
    rng = makeRng(0x1729)
    for bits, passes in ((7, (3, 2)), (6, (2, 2)), (5, (2, 1)), (7, (4, 3)), (8, (2, 1)), (9, (3, 3))):
        x = [rng.rndBelow(r) for _ in range(1 << bits)]
        x[0], x[1] = 0, r - 1
        for flags in range(8):
            inv, bi, bo = bool(flags & 1), bool(flags & 2), bool(flags & 4)
            assert hosttest.ntt(bits, x, roots.omega(bits), flags, passes) == (f.inverse if inv else f.direct)(x, bi, bo), (bits, passes, flags)
    for bits in (10, 11):
        for x in ([rng.rndBelow(r) for _ in range(1 << bits)], [r - 1] * (1 << bits)):
            for flags in (0, 3, 6) if bits == 11 else (0, 1, 2, 5, 7):
                inv, bi, bo = bool(flags & 1), bool(flags & 2), bool(flags & 4)
                assert hosttest.ntt(bits, x, roots.omega(bits), flags) == (f.inverse if inv else f.direct)(x, bi, bo), (bits, flags)


def test_ntt_pass_plan_covers_every_stage_once():
    for n in range(1, 29):
        plan = hosttest.ntt_plan(n)
        s = 1
        for k, (s_lo, t) in enumerate(plan):
            assert s_lo == s and 1 <= t <= (10 if k == 0 else 8)
            s += t
        assert s == n + 1
    assert hosttest.ntt_plan(0) == []


def h2c_cases(m, count, n):
    """n random field-element tuples for hash_to_field-shaped input plus edge values."""
    from oracle.curves import BLS_P
    rng = makeRng(0x42C0 + 16 * m + count)
    rows = []
    for i in range(n):
        rows.append([rng.rndBelow(BLS_P) for _ in range(m * count)])
    rows[0] = [0] * (m * count)                       # u = 0: tv2 == 0 branch of SWU step 7
    rows[1] = [1] + [0] * (m * count - 1)
    rows[2] = [BLS_P - 1] * (m * count)
    return rows


def test_map_to_curve_lanes_match_oracle():
    """SWU + isogeny + add + clearCofactor lane code vs the oracle hasher for
    G1 and G2, count = 1 (mapToCurve / encodeToCurve) and 2 (hashToCurve); EIP-2537 vectors."""
    import numpy as np
    from helpers import load_golden, wire_to_affine
    from noble_curves_amd._native import BLS12_381_G2
    from oracle.curves import BlsG1, BlsG2
    from oracle.h2c import G1_hasher, G2_hasher
    eip = load_golden("bls12_381_eip2537.json")
    u = np.array([np.frombuffer(int(v["Input"], 16).to_bytes(48, "little"), np.uint8) for v in eip["G1"]])
    out, inf = hosttest.map_to_curve(BLS12_381_G1, u, 1, 96)
    for i, v in enumerate(eip["G1"]):
        x, y = wire_to_affine(BLS12_381_G1, out[i])
        assert "%0128x%0128x" % (x, y) == v["Expected"] and not inf[i]
    u = np.array([np.frombuffer(int(v["Input"][:128], 16).to_bytes(48, "little") + int(v["Input"][128:], 16).to_bytes(48, "little"), np.uint8)
                  for v in eip["G2"]])
    out, inf = hosttest.map_to_curve(BLS12_381_G2, u, 1, 192)
    for i, v in enumerate(eip["G2"]):
        x, y = wire_to_affine(BLS12_381_G2, out[i])
        assert "%0128x%0128x%0128x%0128x" % (x[0], x[1], y[0], y[1]) == v["Expected"] and not inf[i]
    # kernel of the isogeny maps to ZERO (test/bls12-381.test.ts:1621-1625)
    t = 1006044755431560595281793557931171729984964515682961911911398807521437683216171091013202870577238485832047490326971
    out, inf = hosttest.map_to_curve(BLS12_381_G1, np.array([np.frombuffer(t.to_bytes(48, "little"), np.uint8)]), 1, 96)
    assert inf[0] and not out.any()
    for curve, hasher, Pt, m, pb in ((BLS12_381_G1, G1_hasher, BlsG1, 1, 96), (BLS12_381_G2, G2_hasher, BlsG2, 2, 192)):
        for count in (1, 2):
            rows = h2c_cases(m, count, 6 if m == 1 else 4)
            u = np.array([np.frombuffer(b"".join(v.to_bytes(48, "little") for v in r), np.uint8) for r in rows])
            out, inf = hosttest.map_to_curve(curve, u, count, pb)
            for i, r in enumerate(rows):
                pts = [hasher.map(r[j * m:(j + 1) * m]) for j in range(count)]
                exp = hasher.clear(pts[0] if count == 1 else pts[0].add(pts[1]))
                assert wire_to_affine(curve, out[i]) == exp.toAffine() and inf[i] == exp.is0(), (curve, count, i)


@pytest.mark.parametrize("curve,name", [(BLS12_381_G1, "g1"), (BLS12_381_G2, "g2")])
def test_lane_ladder_small_order_points(curve, name):
    """Inputs are arbitrary curve points (SURVEY 8a gotcha 1; weierstrass.ts:696-718, :915-928):
    points of order 3 / 11 (G1) and 13 / 23 (G2), alone and mixed with a subgroup component, through
    the per-lane ladder vs k*P by double-and-add on the oracle's complete formulas."""
    import smallorder
    cases = smallorder.small_order_cases(name)
    ks = smallorder.small_order_scalars(3)
    if name == "g2":
        ks = ks[:14] + ks[-3:]
    pts, scal, exp = [], [], []
    for P, _ in cases:
        for k in ks:
            pts.append(P)
            scal.append(k)
            exp.append(smallorder.naive_mul(P, k).toAffine())
    out, inf = hosttest.mul_var(curve, points_to_wire(curve, pts), scalars_to_wire(scal))
    zero = ORACLE_CURVE[curve].ZERO.toAffine()
    for i, e in enumerate(exp):
        assert wire_to_affine(curve, out[i]) == e, (i, hex(scal[i]))
        assert bool(inf[i]) == (e == zero), (i, hex(scal[i]))


@pytest.mark.parametrize("fid,p", [(0, SECP256K1_P), (1, ED25519_P)])
def test_fe9_lazy_field_ops_at_their_bounds(fid, p):
    """fe9.hpp (radix 2^29, lazy limb bounds) against big-int arithmetic, with operands whose limbs sit
    at the top of what each bound type admits (limbs < B*U, U = 2^29 + 2^19): no 64-bit column may
    overflow, outputs must come back below U, every value must match mod p (modular.ts:940-982)."""
    U = (1 << 29) + (1 << 19)
    rng = makeRng(0xFE9 + fid)

    def val(l):
        return sum(x << (29 * i) for i, x in enumerate(l))

    def limbs_of(x):
        return [(x >> (29 * i)) & ((1 << 29) - 1) for i in range(9)]

    def operand(B, kind):
        if kind == 0:
            return [B * U - 1] * 9
        if kind == 1:
            return [0] * 8 + [B * U - 1]
        if kind == 2:
            return [B * U - 1] + [0] * 8
        return [rng.rndBelow(B * U) if rng.rnd64() % 4 else B * U - 1 for _ in range(9)]

    for variant in (11, 12, 17, 71, 23, 32, 22, 15, 33, 77, 46):
        A, B = variant // 10, variant % 10
        for kind in range(12):
            a, b = operand(A, min(kind, 3)), operand(B, min((kind * 7 + 1) % 5, 3))
            va, vb = val(a), val(b)
            assert hosttest.fe9_op(fid, 0, variant, a, b) == va * vb % p
            assert hosttest.fe9_op(fid, 9, variant, a, b) < U
            assert hosttest.fe9_op(fid, 1, variant, a, b) == va * va % p
            if A + B <= 7:
                assert hosttest.fe9_op(fid, 2, variant, a, b) == (va + vb) % p
            if A + B + 1 <= 7:
                assert hosttest.fe9_op(fid, 3, variant, a, b) == (va - vb) % p
            if A + 1 <= 7:
                assert hosttest.fe9_op(fid, 4, variant, a, b) == (-va) % p
            if 2 * A <= 7:
                assert hosttest.fe9_op(fid, 10, variant, a, b) == 2 * va % p
            assert hosttest.fe9_op(fid, 6, variant, a, b) == va % p
            assert hosttest.fe9_op(fid, 8, variant, a, b) < U
            assert hosttest.fe9_op(fid, 7, variant, a, b) == (1 if va % p == 0 else 0)
    # zero tests on exact multiples of p, written with loose limbs; and literal zero
    for j in (0, 1, 2, 5, 31, 33, 64, 100, 200):
        l = limbs_of(j * p) if j * p < (1 << 261) else None
        if l is None:
            hi = (j * p) >> (29 * 8)
            if hi >= 7 * U:
                continue
            l = limbs_of(j * p)[:8] + [hi]
        assert val(l) == j * p
        assert hosttest.fe9_op(fid, 7, 71, l, [0] * 9) == 1
        l2 = list(l)
        l2[3] ^= 1
        assert hosttest.fe9_op(fid, 7, 71, l2, [0] * 9) == 0
    # canonicalisation edge values and inversion
    for x in (0, 1, 2, p - 1, p, p + 1, 2 * p - 1, 2 * p, (1 << 256) - 1, (1 << 256), (1 << 261) - 1):
        assert hosttest.fe9_op(fid, 6, 11, limbs_of(x), [0] * 9) == x % p
    for x in (1, 2, 3, p - 1, p - 2, rng.rndBelow(p), rng.rndBelow(p), 0):
        assert hosttest.fe9_op(fid, 5, 11, limbs_of(x), [0] * 9) == (pow(x, -1, p) if x else 0)


def test_sha512_challenge_matches_hashlib():
    """csrc/sha512.hpp (SHA-512 of R || A || M, then LE mod L) vs hashlib for every padding / block-count
    boundary (total = 64 + len: one block up to len 47, two up to 175, ...) and extreme digests."""
    import hashlib
    from oracle.curves import ED25519_L
    rng = makeRng(0x5A512)
    for ln in list(range(0, 6)) + [46, 47, 48, 49, 63, 64, 111, 112, 113, 174, 175, 176, 177, 255, 256, 300, 1023, 2000]:
        sig = bytes(rng.rnd64() & 0xFF for _ in range(64))
        pk = bytes(rng.rnd64() & 0xFF for _ in range(32))
        msg = bytes(rng.rnd64() & 0xFF for _ in range(ln))
        exp = int.from_bytes(hashlib.sha512(sig[:32] + pk + msg).digest(), "little") % ED25519_L
        assert hosttest.ed25519_challenge(sig, pk, msg) == exp, ln


def test_bls_endomorphism_scalar_split():
    """csrc/endo.hpp (the digit kernel of the verified-set MSM): k = k1 + k2 z^2 (G1) and k = sum d_e z^e (G2)
    modulo r = z^4 - z^2 + 1, every sub-scalar balanced so that the signed windows need no sign handling -
    on edge values around every split boundary and on random scalars."""
    import random
    z = 0xD201000000010000
    X, r = z * z, z ** 4 - z ** 2 + 1
    assert r == BLS_R
    rng = random.Random(1)
    ks = [0, 1, 2, r - 1, r - 2, X, X - 1, X + 1, X // 2, X // 2 + 1, z, z - 1, z + 1, z // 2, z // 2 + 1, z ** 3, z ** 3 - 1, r // 2]
    ks += [(a * X + b) % r for a in (0, 1, X // 2 - 1, X // 2, X // 2 + 1, X - 2, X - 1) for b in (0, 1, X // 2 - 1, X // 2, X // 2 + 1, X - 2, X - 1)]
    ks += [(a * z ** 3 + b * z ** 2 + c * z + d) % r for a in (0, z // 2, z // 2 + 1, z - 1) for b in (0, z // 2, z // 2 + 1, z - 1)
           for c in (0, z // 2 + 1, z - 1) for d in (0, 1, z // 2 + 1, z - 1)]
    ks += [rng.randrange(r) for _ in range(3000)]
    for k in ks:
        k1, k2 = hosttest.bls_endo_split(2, k)
        assert (k1 + k2 * X - k) % r == 0 and max(abs(k1), abs(k2)) <= X // 2 + 1
        d = hosttest.bls_endo_split(4, k)
        assert (sum(d[e] * z ** e for e in range(4)) - k) % r == 0 and max(abs(x) for x in d) <= z // 2 + 2


def test_lane_ladder_g1_subgroup_glv():
    from oracle.curves import BlsG1
    """The GLV ladder for bls12-381 G1 points known to lie in the subgroup (CurveG1E: k = k1 + k2 z^2, second
    stream on (beta x, y); csrc/curves.hpp, mulvar_endo.hip) through the host twin: edge scalars around the
    split boundaries, ZERO, and random pairs against the oracle's multiplyUnsafe."""
    z = 0xD201000000010000
    X = z * z
    rng = makeRng(0x61E)
    ks = [0, 1, 2, 3, BLS_R - 1, BLS_R - 2, X, X - 1, X + 1, X // 2, X // 2 + 1, (X // 2) * X + X // 2 + 1, (X - 1) * X % BLS_R,
          z, z ** 3, BLS_R // 2, 15, 16, 17]
    ks += [rng.rndBelow(BLS_R) for _ in range(16)]
    pts = [BlsG1.BASE.multiplyUnsafe(rng.rndBelow(BLS_R - 1) + 1) for _ in ks]
    pts[5] = BlsG1.ZERO
    out, inf = hosttest.mul_var(12, points_to_wire(BLS12_381_G1, pts), scalars_to_wire(ks))
    for i, (p, k) in enumerate(zip(pts, ks)):
        exp = p.multiplyUnsafe(k).toAffine()
        assert wire_to_affine(BLS12_381_G1, out[i]) == exp, (i, hex(k))
        assert bool(inf[i]) == (exp == BlsG1.ZERO.toAffine())


def test_ecdsa_scalar_side_lane_code():
    """csrc/ecdsa.hip through the host twin: range checks on (r, s), the low-S rule, h = bits2int_modN and
    u1 = h s^-1, u2 = r s^-1 modulo n (Montgomery arithmetic for a modulus with its top bit set, windowed Fermat
    inversion) against Python big ints - weierstrass.ts:1596-1606."""
    n = SECP256K1_N
    rng = makeRng(0xEC0)
    cases = [(1, 1, 0), (n - 1, n - 1, (1 << 256) - 1), (n - 1, 1, n), (2, n >> 1, n - 1), (3, (n >> 1) + 1, 5), (1 << 255, 7, 1 << 255)]
    cases += [(rng.rndBelow(n - 1) + 1, rng.rndBelow(n - 1) + 1, rng.rndBelow(1 << 256)) for _ in range(200)]
    for r, s, h in cases:
        sig = r.to_bytes(32, "big") + s.to_bytes(32, "big")
        for low in (True, False):
            ok, u1, u2 = hosttest.ecdsa_prepare(sig, h.to_bytes(32, "big"), low)
            exp_ok = not (low and s > n >> 1)
            assert ok == exp_ok, (hex(r), hex(s), low)
            if ok:
                inv = pow(s, -1, n)
                assert u1 == (h % n) * inv % n and u2 == r * inv % n
    for r, s in ((0, 5), (5, 0), (n, 5), (5, n), ((1 << 256) - 1, 5)):
        assert not hosttest.ecdsa_prepare(r.to_bytes(32, "big") + s.to_bytes(32, "big"), bytes(32), False)[0]


def test_short_top_window_spreading_keeps_every_digit_weight():
    """MsmPlan::top_tb (round 5): a top window that holds few bits sends digit d >= 1 of point i to bucket
    ((i & submask) << tb) | (d - 1), and the tail keeps only the fold's pending sums of the levels below tb, which weighs bucket b
    by (b mod 2^tb) + 1.  Replayed with Python integers on the SHIPPED plan (csrc/msm_plan.hpp through the host twin): for random
    and extreme scalars below the group order the weight of the bucket is the digit, the bucket exists, the spread digit fits the
    int16 the kernel stores, and sum_w digit_w 2^(c w) - H' gives the scalar back (the reference's window sum, curve.ts:886-902)."""
    import random
    from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2, ED25519, SECP256K1
    from oracle.curves import BLS_R, ED25519_L, SECP256K1_N
    rnd = random.Random(0x5EED)
    cases = [(BLS12_381_G2, 1 << 18, 0, BLS_R, True), (BLS12_381_G1, 1 << 18, 0, BLS_R, True), (BLS12_381_G1, 1 << 20, 0, BLS_R, False),
             (BLS12_381_G1, 1 << 16, 0, BLS_R, True), (SECP256K1, 1 << 20, 0, SECP256K1_N, True), (ED25519, 1 << 20, 0, ED25519_L, False),
             (BLS12_381_G1, 1 << 15, 0, BLS_R, True), (BLS12_381_G2, 1 << 12, 14, BLS_R, True), (BLS12_381_G1, 4096, 12, BLS_R, True)]
    for curve, n, c_over, order, expect_spread in cases:
        p = hosttest.msm_plan_top(curve, n, c_over)
        c, nwin, tb, sub, nb, hp = p["c"], p["nwin"], p["top_tb"], p["top_submask"], p["nb"], p["hprime"]
        assert (tb > 0) == expect_spread, (curve, n, p)
        half = 1 << (c - 1)
        assert hp == sum(1 << (c * w + c - 1) for w in range(nwin)) and nb == half
        ks = [0, 1, order - 1, order - 2, order >> 1, (1 << (order.bit_length() - 1)) - 1, 1 << (order.bit_length() - 1)]
        ks += [rnd.randrange(order) for _ in range(300)]
        top_max = 0
        for i, k in enumerate(ks):
            t = k + hp
            assert t < 1 << (c * nwin)
            digits = [((t >> (c * w)) & ((1 << c) - 1)) - half for w in range(nwin)]
            assert sum(d << (c * w) for w, d in enumerate(digits)) == k          # signed windows: sum_w d_w 2^(c w) = k
            d = digits[-1]
            assert 0 <= d <= p["vmax"] - half + 1                                   # never negative in the top window
            top_max = max(top_max, d)
            if tb and d > 0:
                idx = (i * 2654435761) & 0xFFFFF                                    # any point index
                spread = ((idx & sub) << tb) + min(d, 1 << tb)                     # the digit kernel's mapping
                assert 1 <= spread <= nb and spread < 1 << 15                      # a bucket of the window; fits int16
                bucket = spread - 1
                assert (bucket % (1 << tb)) + 1 == d                               # the weight the cut fold gives it
                assert sub + 1 >= 8
        if tb:
            assert top_max <= 1 << tb


def test_msm_lane_segment_small_plans_are_priced_by_latency():
    """csrc/msm_plan.hpp msm_seg (round 6): a plan that leaves SIMDs empty gets SHORT lane segments (the accumulate kernel's time is
    then seg x the latency of one addition), a plan that fills the chip keeps the segment that makes its lanes one full round; the
    picks measured in profiles/r06_msm_small_sweep_c*.json / r06_msm_timing.json are pinned here."""
    import hosttest
    from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2, ED25519, SECP256K1
    for curve, ls in ((BLS12_381_G1, 0), (BLS12_381_G2, 1), (SECP256K1, 0), (ED25519, 0)):
        for lg in range(6, 21):
            p = hosttest.msm_seg(curve, 1 << lg)
            cap = 65536 * p["accum_waves"]
            assert 4 <= p["seg"] <= 160
            assert p["nseg"] == -(-(1 << lg) // p["seg"]) and p["lanes"] == p["nseg"] << ls
            total = p["nwin"] * p["lanes"]
            if total > cap:                       # several rounds: the measured full-chip segments (16 and up) only
                assert p["seg"] >= 16
    g1 = {lg: hosttest.msm_seg(BLS12_381_G1, 1 << lg) for lg in (10, 13, 14, 16, 17, 20)}
    assert [g1[lg]["seg"] for lg in (10, 13, 14)] == [4, 4, 8]
    assert 11 <= g1[16]["seg"] <= 14 and g1[16]["c"] == 10          # 2^16: c = 10 since round 6 (0.76 against 0.87 ms at c = 13)
    assert g1[17]["seg"] == 21 and g1[20]["seg"] == 128 and g1[20]["nwin"] * g1[20]["lanes"] == 131072   # one full round at two waves per SIMD
    g2 = {lg: hosttest.msm_seg(BLS12_381_G2, 1 << lg) for lg in (12, 14, 18)}
    assert g2[12]["seg"] == 4 and 12 <= g2[14]["seg"] <= 16
    assert g2[18]["nwin"] * g2[18]["lanes"] <= 131072 < g2[18]["nwin"] * ((-(-(1 << 18) // (g2[18]["seg"] - 1))) << 1)   # the shortest segment that is one round
    # a forced width changes the bucket size the merge chain is priced with
    assert hosttest.msm_seg(BLS12_381_G1, 1 << 14, 12)["seg"] <= hosttest.msm_seg(BLS12_381_G1, 1 << 14, 8)["seg"]
