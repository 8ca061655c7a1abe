#!/usr/bin/env python3
"""Latency of ONE rank's local phase of a window-sharded MSM (part 0 of G, resident set) plus a correctness check of the whole
split MSM, for A/B runs of the share-mode knobs (NCG_MSM_MERGE_UNITS / NCG_MSM_TOTALS_SPLIT / NCG_MSM_SHARE_QBLOCKS in -DNCG_AB_BUILD
libraries: NCG_LIB=tools/_build/libncg_ab.so).   python tools/share_ab.py [--curve g1] [--parts 8] [--reps 30] [--tag x]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, bench
from helpers import wire_to_affine
from noble_curves_amd import get_engine
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2
from oracle.curves import BLS_R, BlsG1, BlsG2
ap = argparse.ArgumentParser()
ap.add_argument("--curve", default="g1"); ap.add_argument("--log2n", type=int, default=0)
ap.add_argument("--parts", type=int, default=8); ap.add_argument("--reps", type=int, default=30); ap.add_argument("--tag", default="")
a = ap.parse_args()
dev = torch.device("cuda", 0); st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); s = st.cuda_stream
eng = get_engine(0)
cid, O, lg = (BLS12_381_G1, BlsG1, 20) if a.curve == "g1" else (BLS12_381_G2, BlsG2, 18)
n = 1 << (a.log2n or lg)
pts, ks = bench.gen_points(eng, cid, O, n, 0x1234567 + 7, 0x6789, dev, s)
sc = bench.gen_scalars(n, 254, 5, dev); sc[::17] = 0
exp = O.BASE.multiplyUnsafe(sum(k * x for k, x in zip(ks, bench.scalars_to_ints(sc))) % BLS_R).toAffine()
rs = eng.upload_points(cid, pts.cpu().numpy())
got, _ = eng.msm_split_windows_dev(cid, n, a.parts, 0, sc.data_ptr(), s, rs)
assert wire_to_affine(cid, got) == exp, "split MSM mismatch"
def wall(f, reps):
    for _ in range(5): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return round(ts[0], 4), round(ts[len(ts) // 2], 4)
local = wall(lambda: eng.msm_shard_windows_local_dev(cid, n, 0, a.parts, 0, sc.data_ptr(), s, rs), a.reps)
whole = wall(lambda: rs.msm_dev(sc.data_ptr(), s), max(5, a.reps // 3))
print(json.dumps({"tag": a.tag, "curve": a.curve, "n": n, "parts": a.parts, "local_part0_ms_min_med": local, "one_gpu_ms_min_med": whole,
                  "plan": eng.msm_last_plan(), "knobs": {k: v for k, v in os.environ.items() if k.startswith("NCG_MSM")}}))
