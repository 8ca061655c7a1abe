#!/usr/bin/env python3
"""Window-width sweep of the single-GPU MSM: median wall time per (curve, log2 n, c) with NCG_MSM_C forcing the width;
feeds the plan table in csrc/msm_plan.hpp.   python tools/msm_csweep.py [--curves g1,g2] [--min 12] [--max 20]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, bench
from helpers import wire_to_affine
from noble_curves_amd import get_engine
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2
from oracle.curves import BLS_R, BlsG1, BlsG2
ap = argparse.ArgumentParser()
ap.add_argument("--curves", default="g1,g2")
ap.add_argument("--min", type=int, default=12)
ap.add_argument("--max", type=int, default=20)
ap.add_argument("--reps", type=int, default=9)
ap.add_argument("--out", default=None)
args = ap.parse_args()
dev = torch.device("cuda", 0); st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); s = st.cuda_stream
eng = get_engine(0)
rows = []
for name, cid, O, top in (("g1", BLS12_381_G1, BlsG1, args.max), ("g2", BLS12_381_G2, BlsG2, args.max - 2)):
    if name not in args.curves.split(","):
        continue
    nmax = 1 << top
    pts, ks = bench.gen_points(eng, cid, O, nmax, 0x1234567 + 7, 0x6789, dev, s)
    sc = bench.gen_scalars(nmax, 254, 5, dev)
    sc[::17] = 0
    sci = bench.scalars_to_ints(sc)
    for lg in range(args.min - (0 if name == "g1" else 2), top + 1):
        n = 1 << lg
        expect = O.BASE.multiplyUnsafe(sum(k * x for k, x in zip(ks[:n], sci[:n])) % BLS_R).toAffine()
        c0 = max(2, min(16, lg - (3 if name == "g2" else 4)))
        row = {"curve": name, "log2n": lg, "default_c": c0, "ms": {}}
        for c in range(max(4, c0 - 3), min(16, c0 + 4) + 1):
            os.environ["NCG_MSM_C"] = str(c)
            f = lambda: eng.msm_dev(cid, n, pts.data_ptr(), sc.data_ptr(), s)
            r = f()
            assert wire_to_affine(cid, r[0]) == expect, (name, lg, c)
            f()
            ts = []
            for _ in range(args.reps):
                t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
            ts.sort()
            row["ms"][c] = round(ts[len(ts) // 2], 4)
        os.environ.pop("NCG_MSM_C", None)
        row["best_c"] = min(row["ms"], key=row["ms"].get)
        rows.append(row)
        print(json.dumps(row), flush=True)
if args.out:
    json.dump(rows, open(args.out, "w"), indent=1)
