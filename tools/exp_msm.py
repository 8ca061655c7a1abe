#!/usr/bin/env python3
"""A few MSMs of one curve and size (for a kernel trace) and their wall time.  python tools/exp_msm.py <secp|ed|g1|g2> <log2n> [scalar bits]
(knobs of -DNCG_AB_BUILD libraries through the environment, NCG_LIB selects the library)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from helpers import wire_to_affine
from noble_curves_amd import get_engine
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2, ED25519, SECP256K1
from oracle.curves import BlsG1, BlsG2, Ed25519, Secp256k1
dev = torch.device("cuda", 0); eng = get_engine(0)
name = sys.argv[1] if len(sys.argv) > 1 else "g1"
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cid, O = {"secp": (SECP256K1, Secp256k1), "ed": (ED25519, Ed25519), "g1": (BLS12_381_G1, BlsG1), "g2": (BLS12_381_G2, BlsG2)}[name]
n = 1 << lg
order = O.Fn.ORDER
pts, ks = bench.gen_points(eng, cid, O, n, 77, 5, dev, None)
bits = int(sys.argv[3]) if len(sys.argv) > 3 else order.bit_length() - 2
sc = bench.gen_scalars(n, bits, 5, dev)
sci = bench.scalars_to_ints(sc)
exp = O.BASE.multiplyUnsafe(sum(k * s for k, s in zip(ks, sci)) % order).toAffine()
r = eng.msm_dev(cid, n, pts.data_ptr(), sc.data_ptr())
assert wire_to_affine(cid, r[0]) == exp
for _ in range(3): eng.msm_dev(cid, n, pts.data_ptr(), sc.data_ptr())
ts = []
for _ in range(10):
    t0 = time.perf_counter(); eng.msm_dev(cid, n, pts.data_ptr(), sc.data_ptr()); ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print(name, lg, "wall_ms min %.3f median %.3f" % (ts[0], ts[5]), eng.msm_last_plan())
