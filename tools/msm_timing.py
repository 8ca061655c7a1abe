#!/usr/bin/env python3
"""MSM wall time per size on one GPU (inputs resident): generic `pippenger` path, resident set, verified (endomorphism) set.
Every result is checked with the progression identity of test/slow-curves.test.ts:185-252.
    python tools/msm_timing.py [--curves g1,g2] [--min 14] [--max 20] [--reps 20] [--out file.json] [--mark]
--mark prints one line per size to stderr (NCG_TIMING-style) so a kernel trace of the same run can be cut per size.
The 2^17-point row at the plan of a 2^17 shard is what one GPU of an 8-GPU strong-scaling run executes."""
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
ap.add_argument("--min", type=int, default=14)
ap.add_argument("--max", type=int, default=20)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--out", default=None)
ap.add_argument("--no-resident", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda", 0); st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); s = st.cuda_stream
eng = get_engine(0)
rows = []
for name, cid, O, top in (("g1", BLS12_381_G1, BlsG1, args.max), ("g2", BLS12_381_G2, BlsG2, args.max - 2)):
    if name not in args.curves.split(","):
        continue
    nmax = 1 << top
    a, b = 0x1234567 + 7, 0x6789
    pts, ks = bench.gen_points(eng, cid, O, nmax, a, b, dev, s)
    sc = bench.gen_scalars(nmax, 254, 5, dev)
    sc[::17] = 0
    sci = bench.scalars_to_ints(sc)
    for lg in range(max(8, args.min - (0 if name == "g1" else 2)), top + 1):
        n = 1 << lg
        expect = O.BASE.multiplyUnsafe(sum(k * x for k, x in zip(ks[:n], sci[:n])) % BLS_R).toAffine()
        row = {"curve": name, "log2n": lg}

        def timed(f, key):
            r = f(); torch.cuda.synchronize()
            assert wire_to_affine(cid, r[0]) == expect, (name, lg, key)
            for _ in range(3): f()
            torch.cuda.synchronize()
            ts = []
            for _ in range(args.reps):
                t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
            ts.sort()
            row[key] = {"min_ms": round(ts[0], 4), "median_ms": round(ts[len(ts) // 2], 4), "max_ms": round(ts[-1], 4)}
        timed(lambda: eng.msm_dev(cid, n, pts.data_ptr(), sc.data_ptr(), s), "generic")
        if not args.no_resident:
            res = eng.upload_points(cid, pts[:n].cpu().numpy())
            timed(lambda: res.msm_dev(sc.data_ptr(), s), "resident")
            if res.precompute():
                timed(lambda: res.msm_dev(sc.data_ptr(), s), "resident_precomp")
            res.free()
            res = eng.upload_points(cid, pts[:n].cpu().numpy())
            assert res.verify_subgroup() == -1
            timed(lambda: res.msm_dev(sc.data_ptr(), s), "resident_verified")
            if res.precompute():
                timed(lambda: res.msm_dev(sc.data_ptr(), s), "verified_precomp")
            res.free()
        rows.append(row)
        print(json.dumps(row), flush=True)
if args.out:
    json.dump(rows, open(args.out, "w"), indent=1)
