#!/usr/bin/env python3
"""Throughput of the hot-path entry points that bench.py's headline lines do not cover
(fixed-base multiply, batch decode, normalisation, other curves' multiply / MSM).  Device-resident
inputs, HIP-event timing on an explicit stream; prints one JSON object (also to --out)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from noble_curves_amd import get_engine  # noqa: E402
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2, ED25519, SECP256K1  # noqa: E402
from oracle.curves import BLS_R, BlsG1, BlsG2, ED25519_L, Ed25519, SECP256K1_N, Secp256k1, makeRng  # noqa: E402


def timeit(fn, steps=3, warmup=1):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    s = st.cuda_stream
    eng = get_engine(0)
    res = {}
    P = lambda t: t.data_ptr()  # noqa: E731

    def rate(name, n, ms, unit):
        res[name] = {"n": n, "ms": round(ms, 3), "per_s": n / (ms * 1e-3), "unit": unit}
        print("%-44s n=2^%-2d %9.3f ms  %.3e %s/s" % (name, n.bit_length() - 1, ms, n / (ms * 1e-3), unit), flush=True)

    # ---- secp256k1: fixed-base, decode, normalise, MSM
    n = 1 << 20
    sc = bench.gen_scalars(n, 255, 5, dev)
    out = torch.empty((n, 64), dtype=torch.uint8, device=dev)
    inf = torch.empty((n,), dtype=torch.uint8, device=dev)
    ok = torch.empty((n,), dtype=torch.uint8, device=dev)
    eng.mul_base_batch_dev(SECP256K1, n, P(sc), P(out), P(inf), s)
    rate("secp256k1 fixed-base multiply", n, timeit(lambda: eng.mul_base_batch_dev(SECP256K1, n, P(sc), P(out), P(inf), s)), "scalar-mults")
    pts = out.clone()
    enc = torch.empty((n, 33), dtype=torch.uint8, device=dev)            # SEC1-compress on the device with torch ops
    enc[:, 0] = 2 + (pts[:, 32] & 1)
    enc[:, 1:] = torch.flip(pts[:, :32], dims=[1])
    dec = torch.empty((n, 64), dtype=torch.uint8, device=dev)
    f = lambda: eng._check(eng.lib.ncg_decode_points_batch_dev(eng.h, SECP256K1, n, P(enc), 0, P(dec), P(ok), P(inf), s))  # noqa: E731
    f()
    torch.cuda.synchronize()
    assert bool((dec == pts).all().item()) and int(ok.sum().item()) == n, "secp256k1 decode mismatch"
    rate("secp256k1 SEC1 decode (sqrt)", n, timeit(f), "points")
    a_ = makeRng(9)
    a, b = a_.rndBelow(SECP256K1_N - 1) + 1, a_.rndBelow(SECP256K1_N - 1) + 1
    kpts, _ = bench.gen_points(eng, SECP256K1, Secp256k1, n, a, b, dev, s)
    rate("secp256k1 MSM", n, timeit(lambda: eng.msm_dev(SECP256K1, n, P(kpts), P(sc), s)), "points")
    ones = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
    ones[:, 0] = 1
    rate("secp256k1 point sum (MSM, unit scalars)", n, timeit(lambda: eng.msm_dev(SECP256K1, n, P(kpts), P(ones), s)), "points")
    padd = lambda: eng._check(eng.lib.ncg_add_pairs_batch_dev(eng.h, SECP256K1, n, P(kpts), P(pts), 0, P(dec), P(inf), s))  # noqa: E731
    rate("secp256k1 pairwise point add", n, timeit(padd), "points")
    t1 = torch.empty((n, 64), dtype=torch.uint8, device=dev)
    t2 = torch.empty((n, 64), dtype=torch.uint8, device=dev)

    def ecdsa_shape():                                                   # u1*G + u2*P per item (weierstrass.ts:1609)
        eng.mul_base_batch_dev(SECP256K1, n, P(sc), P(t1), P(inf), s)
        eng.mul_var_batch_dev(SECP256K1, n, P(kpts), P(sc), P(t2), P(inf), s)
        eng._check(eng.lib.ncg_add_pairs_batch_dev(eng.h, SECP256K1, n, P(t1), P(t2), 0, P(dec), P(inf), s))
    rate("secp256k1 u1*G + u2*P (mulAddUnsafe, ECDSA-verify shape)", n, timeit(ecdsa_shape), "double-mults")
    # full ECDSA verify (weierstrass.ts:1571-1620) of 2^18 signatures under 64 keys: key decompression, s^-1 mod n,
    # u1 G + u2 P and the comparison on the device; 1/16 of the rows corrupted, verdicts checked
    ne = 1 << 18
    from oracle.curves import SECP256K1_N as NN
    erng = makeRng(0xEC)
    dkeys = [erng.rndBelow(NN - 1) + 1 for _ in range(64)]
    kk = [erng.rndBelow(NN - 1) + 1 for _ in range(ne)]
    ksc = torch.from_numpy(np.frombuffer(b"".join(k.to_bytes(32, "little") for k in kk), np.uint8).reshape(ne, 32).copy()).to(dev)
    dsc = torch.from_numpy(np.frombuffer(b"".join(d.to_bytes(32, "little") for d in dkeys), np.uint8).reshape(64, 32).copy()).to(dev)
    raff = torch.empty((ne, 64), dtype=torch.uint8, device=dev)
    rinf = torch.empty((ne,), dtype=torch.uint8, device=dev)
    eng.mul_base_batch_dev(SECP256K1, ne, P(ksc), P(raff), P(rinf), s)
    kaff = torch.empty((64, 64), dtype=torch.uint8, device=dev)
    eng.mul_base_batch_dev(SECP256K1, 64, P(dsc), P(kaff), P(rinf), s)
    torch.cuda.synchronize()
    kenc, _ = eng.encode_points_batch(SECP256K1, kaff.cpu().numpy())
    rx = raff.cpu().numpy()[:, :32]
    import hashlib
    from noble_curves_amd.schnorr import challenge as bip340_challenge
    ry = raff.cpu().numpy()[:, 32:]
    kx = kaff.cpu().numpy()
    sig_rows, hash_rows, msg_rows, ssig_rows, spk_rows = [], [], [], [], []
    for i in range(ne):
        rx_i = int.from_bytes(bytes(rx[i]), "little")
        r = rx_i % NN
        msg = (i * 0x9E3779B97F4A7C15 + 12345).to_bytes(32, "big")
        hb = hashlib.sha256(msg).digest()
        h = int.from_bytes(hb, "big") % NN
        d = dkeys[i % 64]
        sv = pow(kk[i], -1, NN) * (h + r * d) % NN
        if sv > NN >> 1:
            sv = NN - sv
        sig_rows.append(r.to_bytes(32, "big") + sv.to_bytes(32, "big"))
        hash_rows.append(hb)
        msg_rows.append(msg)
        # BIP-340 with the same nonce point and key (negated where y is odd)
        py_odd = kx[i % 64][32] & 1
        ry_odd = ry[i][0] & 1
        dd = NN - d if py_odd else d
        k2 = NN - kk[i] if ry_odd else kk[i]
        pkb = bytes(kx[i % 64][31::-1])
        rb = rx_i.to_bytes(32, "big")
        ssig_rows.append(rb + ((k2 + bip340_challenge(rb, pkb, msg) * dd) % NN).to_bytes(32, "big"))
        spk_rows.append(pkb)
    S = np.frombuffer(b"".join(sig_rows), np.uint8).reshape(ne, 64).copy()
    S[::16, 40] ^= 1
    # 2^18 distinct signatures tiled to 2^20 rows (the work per row does not depend on its neighbours)
    T = 4
    Sd = torch.from_numpy(S).to(dev).repeat(T, 1).contiguous()
    Hd = torch.from_numpy(np.frombuffer(b"".join(hash_rows), np.uint8).reshape(ne, 32).copy()).to(dev).repeat(T, 1).contiguous()
    Md = torch.from_numpy(np.frombuffer(b"".join(msg_rows), np.uint8).reshape(ne, 32).copy()).to(dev).repeat(T, 1).contiguous()
    Od = (torch.arange(T * ne + 1, dtype=torch.int64, device=dev) * 32).contiguous()
    Kd = torch.from_numpy(np.ascontiguousarray(kenc[np.arange(ne) % 64])).to(dev).repeat(T, 1).contiguous()
    okd = torch.empty((T * ne,), dtype=torch.uint8, device=dev)
    exp_ok = np.ones((ne,), np.uint8)
    exp_ok[::16] = 0
    ev = lambda: eng.ecdsa_verify_batch_dev(T * ne, P(Sd), P(Hd), P(Kd), True, P(okd), s)  # noqa: E731
    ms_e = timeit(ev)
    assert np.array_equal(okd.cpu().numpy(), np.tile(exp_ok, T)), "ECDSA verdicts"
    rate("secp256k1 ECDSA verify batch (compressed keys, 64 distinct)", T * ne, ms_e, "verifies")
    evm = lambda: eng._check(eng.lib.ncg_ecdsa_verify_batch_msgs_dev(eng.h, SECP256K1, T * ne, P(Sd), P(Md), P(Od), P(Kd), 1, P(okd), s))  # noqa: E731
    ms_m = timeit(evm)
    assert np.array_equal(okd.cpu().numpy(), np.tile(exp_ok, T)), "ECDSA verdicts (device hash)"
    rate("secp256k1 ECDSA verify from 32-byte messages (SHA-256 on the device)", T * ne, ms_m, "verifies")
    SS = np.frombuffer(b"".join(ssig_rows), np.uint8).reshape(ne, 64).copy()
    SS[::16, 40] ^= 1
    SSd = torch.from_numpy(SS).to(dev).repeat(T, 1).contiguous()
    PKd = torch.from_numpy(np.frombuffer(b"".join(spk_rows), np.uint8).reshape(ne, 32).copy()).to(dev).repeat(T, 1).contiguous()
    evs = lambda: eng._check(eng.lib.ncg_schnorr_verify_batch_msgs_dev(eng.h, T * ne, P(SSd), P(Md), P(Od), P(PKd), P(okd), s))  # noqa: E731
    ms_s = timeit(evs)
    assert np.array_equal(okd.cpu().numpy(), np.tile(exp_ok, T)), "Schnorr verdicts"
    rate("secp256k1 BIP-340 Schnorr verify from 32-byte messages (tagged hash on the device)", T * ne, ms_s, "verifies")
    proj = torch.cat([kpts, torch.zeros((n, 32), dtype=torch.uint8, device=dev)], dim=1)
    proj[:, 64] = 1                                                        # Z = 1
    fn = lambda: eng._check(eng.lib.ncg_normalize_batch_dev(eng.h, SECP256K1, n, P(proj), P(dec), P(inf), s))  # noqa: E731
    rate("secp256k1 normalizeZ batch", n, timeit(fn), "points")

    # ---- ed25519: variable-base multiply, MSM, decode
    esc = bench.gen_scalars(n, 252, 6, dev)
    rng = makeRng(11)
    a, b = rng.rndBelow(ED25519_L - 1) + 1, rng.rndBelow(ED25519_L - 1) + 1
    epts, _ = bench.gen_points(eng, ED25519, Ed25519, n, a, b, dev, s)
    eout = torch.empty((n, 64), dtype=torch.uint8, device=dev)
    rate("ed25519 variable-base multiply", n, timeit(lambda: eng.mul_var_batch_dev(ED25519, n, P(epts), P(esc), P(eout), P(inf), s)), "scalar-mults")
    rate("ed25519 fixed-base multiply", n, timeit(lambda: eng.mul_base_batch_dev(ED25519, n, P(esc), P(eout), P(inf), s)), "scalar-mults")
    rate("ed25519 MSM", n, timeit(lambda: eng.msm_dev(ED25519, n, P(epts), P(esc), s)), "points")
    eenc = epts[:, 32:].clone()
    eenc[:, 31] |= (epts[:, 0] & 1) << 7
    fe = lambda: eng._check(eng.lib.ncg_decode_points_batch_dev(eng.h, ED25519, n, P(eenc), 1, P(eout), P(ok), P(inf), s))  # noqa: E731
    fe()
    torch.cuda.synchronize()
    assert bool((eout == epts).all().item()), "ed25519 decode mismatch"
    rate("ed25519 decode (zip215)", n, timeit(fe), "points")

    # ---- bls12-381: G1/G2 variable-base + fixed-base multiply, G1 decode with subgroup check
    m = 1 << 18
    rng = makeRng(13)
    a, b = rng.rndBelow(BLS_R - 1) + 1, rng.rndBelow(BLS_R - 1) + 1
    gsc = bench.gen_scalars(m, 254, 7, dev)
    g1, _ = bench.gen_points(eng, BLS12_381_G1, BlsG1, m, a, b, dev, s)
    g1o = torch.empty((m, 96), dtype=torch.uint8, device=dev)
    rate("bls12-381 G1 variable-base multiply", m, timeit(lambda: eng.mul_var_batch_dev(BLS12_381_G1, m, P(g1), P(gsc), P(g1o), P(inf), s), 2), "scalar-mults")
    rs1 = eng.upload_points(BLS12_381_G1, g1.cpu().numpy())
    assert rs1.verify_subgroup() == -1 and rs1.in_subgroup
    ref1 = g1o.clone()
    rs1.mul_var_batch_dev(P(gsc), P(g1o), P(inf), s)
    torch.cuda.synchronize()
    assert bool((g1o == ref1).all().item()), "G1 verified-set ladder mismatch"
    rate("bls12-381 G1 variable-base multiply, verified set (GLV ladder)", m, timeit(lambda: rs1.mul_var_batch_dev(P(gsc), P(g1o), P(inf), s), 2), "scalar-mults")
    rs1.free()
    rate("bls12-381 G1 fixed-base multiply", m, timeit(lambda: eng.mul_base_batch_dev(BLS12_381_G1, m, P(gsc), P(g1o), P(inf), s), 2), "scalar-mults")
    genc = torch.empty((m, 48), dtype=torch.uint8, device=dev)              # compressed: x big-endian + flags
    fenc = lambda: eng._check(eng.lib.ncg_encode_points_batch_dev(eng.h, BLS12_381_G1, m, P(g1), P(genc), P(ok), s))  # noqa: E731
    rate("bls12-381 G1 encode (compressed)", m, timeit(fenc, 2), "points")
    fg = lambda: eng._check(eng.lib.ncg_decode_points_batch_dev(eng.h, BLS12_381_G1, m, P(genc), 0, P(g1o), P(ok), P(inf), s))  # noqa: E731
    fg()
    torch.cuda.synchronize()
    assert bool((g1o == g1).all().item()) and int(ok[:m].sum().item()) == m, "G1 decode mismatch"
    rate("bls12-381 G1 decode + subgroup check", m, timeit(fg, 2), "points")
    g2, _ = bench.gen_points(eng, BLS12_381_G2, BlsG2, m, a, b, dev, s)
    g2o = torch.empty((m, 192), dtype=torch.uint8, device=dev)
    rate("bls12-381 G2 variable-base multiply", m, timeit(lambda: eng.mul_var_batch_dev(BLS12_381_G2, m, P(g2), P(gsc), P(g2o), P(inf), s), 2), "scalar-mults")
    rs2 = eng.upload_points(BLS12_381_G2, g2.cpu().numpy())
    assert rs2.verify_subgroup() == -1 and rs2.in_subgroup
    ref2 = g2o.clone()
    rs2.mul_var_batch_dev(P(gsc), P(g2o), P(inf), s)
    torch.cuda.synchronize()
    assert bool((g2o == ref2).all().item()), "G2 verified-set ladder mismatch"
    rate("bls12-381 G2 variable-base multiply, verified set (psi ladder)", m, timeit(lambda: rs2.mul_var_batch_dev(P(gsc), P(g2o), P(inf), s), 2), "scalar-mults")
    rs2.free()
    g2enc = torch.empty((m, 96), dtype=torch.uint8, device=dev)
    eng._check(eng.lib.ncg_encode_points_batch_dev(eng.h, BLS12_381_G2, m, P(g2), P(g2enc), P(ok), s))
    fg2 = lambda: eng._check(eng.lib.ncg_decode_points_batch_dev(eng.h, BLS12_381_G2, m, P(g2enc), 0, P(g2o), P(ok), P(inf), s))  # noqa: E731
    fg2()
    torch.cuda.synchronize()
    assert bool((g2o == g2).all().item()) and int(ok[:m].sum().item()) == m, "G2 decode mismatch"
    rate("bls12-381 G2 decode + subgroup check", m, timeit(fg2, 2), "points")
    # ---- hash-to-curve (device part): 2 field elements per point -> SWU, isogeny, add, clearCofactor
    hm = 1 << 18
    u1 = torch.randint(0, 256, (hm, 2 * 48), dtype=torch.uint8, device=dev)
    u1[:, 47] &= 0x0F
    u1[:, 95] &= 0x0F                                                     # < 2^380 < p
    rate("bls12-381 G1 hashToCurve map (count=2)", hm,
         timeit(lambda: eng.map_to_curve_batch_dev(BLS12_381_G1, hm, 2, P(u1), P(g1o), P(inf), s), 2), "points")
    hm2 = 1 << 16
    u2 = torch.randint(0, 256, (hm2, 4 * 48), dtype=torch.uint8, device=dev)
    for k in range(4):
        u2[:, 48 * k + 47] &= 0x0F
    rate("bls12-381 G2 hashToCurve map (count=2)", hm2,
         timeit(lambda: eng.map_to_curve_batch_dev(BLS12_381_G2, hm2, 2, P(u2), P(g2o), P(inf), s), 2), "points")
    if args.out:
        with open(args.out, "w") as fjs:
            json.dump(res, fjs, indent=1)


if __name__ == "__main__":
    main()
