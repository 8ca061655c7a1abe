#!/usr/bin/env python3
"""Entries per accumulate lane (`seg`, ncg_msm_set_tuning) against wall time, for the whole MSM and for one rank's share of a
window-sharded one:  python tools/seg_sweep.py [--curve g1] [--log2n 20] [--parts 1,8] [--segs 16,24,32,...] [--kind resident]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, bench
from noble_curves_amd import get_engine
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2
from oracle.curves import BlsG1, BlsG2
ap = argparse.ArgumentParser()
ap.add_argument("--curve", default="g1"); ap.add_argument("--log2n", type=int, default=20)
ap.add_argument("--parts", default="1,8"); ap.add_argument("--kind", default="resident"); ap.add_argument("--reps", type=int, default=8)
ap.add_argument("--segs", default="0,16,24,32,48,64,96,128,192,256")
ap.add_argument("--out", default=None)
a = ap.parse_args()
dev = torch.device("cuda", 0); st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); s = st.cuda_stream
eng = get_engine(0)
cid, O = (BLS12_381_G1, BlsG1) if a.curve == "g1" else (BLS12_381_G2, BlsG2)
n = 1 << a.log2n
pts, ks = bench.gen_points(eng, cid, O, n, 0x1234567 + 7, 0x6789, dev, s)
sc = bench.gen_scalars(n, 254, 5, dev); sc[::17] = 0
rs = None
if a.kind != "generic":
    rs = eng.upload_points(cid, pts.cpu().numpy())
    if a.kind.startswith("verified"): assert rs.verify_subgroup() == -1
    if a.kind.endswith("precomputed") or a.kind.endswith("precomp"): assert rs.precompute()
P = 0 if rs is not None else pts.data_ptr()
rows = []
for G in [int(x) for x in a.parts.split(",")]:
    ref = None
    for seg in [int(x) for x in a.segs.split(",")]:
        eng.msm_set_tuning(seg, -1)
        if G <= 1:
            f = (lambda: rs.msm_dev(sc.data_ptr(), s)) if rs is not None else (lambda: eng.msm_dev(cid, n, P, sc.data_ptr(), s))
        else:
            f = lambda: eng.msm_shard_windows_local_dev(cid, n, 0, G, P, sc.data_ptr(), s, rs)
        r = f(); torch.cuda.synchronize()
        r0 = r[0] if G <= 1 else r
        if ref is None: ref = r0.copy()
        assert G > 1 or (r0 == ref).all(), (G, seg)   # the cut never changes the result (a slot holds projective sums: only the MSM itself is compared)
        for _ in range(3): f()
        ts = []
        for _ in range(a.reps):
            t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        lp = eng.msm_last_plan()
        row = {"parts": G, "seg_req": seg, "seg": lp["seg"], "run_serial": lp["run_serial"], "long_runs": lp["long_runs"], "min_ms": round(ts[0], 4), "median_ms": round(ts[len(ts) // 2], 4)}
        rows.append(row); print(json.dumps(row), flush=True)
eng.msm_set_tuning(0, -1)
if a.out: json.dump(rows, open(a.out, "w"), indent=1)
