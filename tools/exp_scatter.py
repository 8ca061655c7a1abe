#!/usr/bin/env python3
"""A few 2^20-point G1 MSMs (for a kernel trace of the sort kernels) and their wall time; NCG_EXP_* knobs select variants."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from helpers import wire_to_affine
from noble_curves_amd import get_engine
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2
from oracle.curves import BLS_R, BlsG1, BlsG2
dev = torch.device("cuda", 0); eng = get_engine(0)
g2 = len(sys.argv) > 1 and sys.argv[1] == "g2"
cid, O, n = (BLS12_381_G2, BlsG2, 1 << 18) if g2 else (BLS12_381_G1, BlsG1, 1 << 20)
pts, ks = bench.gen_points(eng, cid, O, n, 77, 5, dev, None)
sc = bench.gen_scalars(n, 254, 5, dev)
sci = bench.scalars_to_ints(sc)
exp = O.BASE.multiplyUnsafe(sum(k * s for k, s in zip(ks, sci)) % BLS_R).toAffine()
r = eng.msm_dev(cid, n, pts.data_ptr(), sc.data_ptr())
assert wire_to_affine(cid, r[0]) == exp
for _ in range(3): eng.msm_dev(cid, n, pts.data_ptr(), sc.data_ptr())
ts = []
for _ in range(10):
    t0 = time.perf_counter(); eng.msm_dev(cid, n, pts.data_ptr(), sc.data_ptr()); ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print("wall_ms min %.3f median %.3f" % (ts[0], ts[5]))
