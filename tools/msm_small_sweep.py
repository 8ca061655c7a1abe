#!/usr/bin/env python3
"""(window width, lane segment) sweep of the single-GPU MSM at the sizes where the accumulate kernel does not fill the chip
(a lane adds `seg` sorted entries one after the other: with few lanes the kernel's time is seg x the latency of one addition on a
lone wave).  NCG_MSM_C forces the width, ncg_msm_set_tuning the segment; every result is checked.  Feeds msm_plan.hpp / msm_seg.
    python tools/msm_small_sweep.py [--curves g1,g2] [--min 10] [--max 17] [--out file.json]"""
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
ap.add_argument("--min", type=int, default=10)
ap.add_argument("--max", type=int, default=17)
ap.add_argument("--reps", type=int, default=9)
ap.add_argument("--segs", default="0,4,6,8,12")
ap.add_argument("--out", default=None)
args = ap.parse_args()
dev = torch.device("cuda", 0); st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); s = st.cuda_stream
eng = get_engine(0)
segs = [int(x) for x in args.segs.split(",")]
rows = []
for name, cid, O in (("g1", BLS12_381_G1, BlsG1), ("g2", BLS12_381_G2, BlsG2)):
    if name not in args.curves.split(","):
        continue
    nmax = 1 << args.max
    pts, ks = bench.gen_points(eng, cid, O, nmax, 0x1234567 + 7, 0x6789, dev, s)
    sc = bench.gen_scalars(nmax, 254, 5, dev)
    sc[::17] = 0
    sci = bench.scalars_to_ints(sc)
    for lg in range(args.min, args.max + 1):
        n = 1 << lg
        expect = O.BASE.multiplyUnsafe(sum(k * x for k, x in zip(ks[:n], sci[:n])) % BLS_R).toAffine()
        row = {"curve": name, "log2n": lg, "ms": {}}
        f = lambda: eng.msm_dev(cid, n, pts.data_ptr(), sc.data_ptr(), s)
        def med():
            r = f()
            assert wire_to_affine(cid, r[0]) == expect, (name, lg)
            f()
            ts = []
            for _ in range(args.reps):
                t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
            ts.sort()
            return round(ts[len(ts) // 2], 4)
        os.environ.pop("NCG_MSM_C", None)
        eng.msm_set_tuning(seg=0)
        row["default_ms"] = med()
        row["default_plan"] = eng.msm_last_plan()
        for c in range(max(4, lg - 7), min(16, lg) + 1):
            os.environ["NCG_MSM_C"] = str(c)
            for seg in segs:
                eng.msm_set_tuning(seg=seg)
                row["ms"]["c%d_s%d" % (c, seg)] = med()
        os.environ.pop("NCG_MSM_C", None)
        eng.msm_set_tuning(seg=0)
        best = min(row["ms"], key=row["ms"].get)
        row["best"] = best; row["best_ms"] = row["ms"][best]
        rows.append(row)
        print(json.dumps({k: row[k] for k in ("curve", "log2n", "default_ms", "default_plan", "best", "best_ms")}), flush=True)
        top = sorted(row["ms"].items(), key=lambda kv: kv[1])[:8]
        print("   ", top, flush=True)
if args.out:
    json.dump(rows, open(args.out, "w"), indent=1)
