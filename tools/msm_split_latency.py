#!/usr/bin/env python3
"""Latency of ONE MSM when its windows are cut into P parts that run CONCURRENTLY on the engine's lanes of one GPU (the sort of one
part overlaps the accumulate of another, the dependent fold / tail of one part the accumulate of the next), against the plain
synchronous call.  Uses the public pieces: ncg_msm_async_submit(NCG_MSM_ASYNC_PART) + collect_slot + ncg_msm_shard_combine.
    python tools/msm_split_latency.py [--curve g1] [--log2n 20]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, bench
from helpers import wire_to_affine
from noble_curves_amd import get_engine
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2
from oracle.curves import BLS_R, BlsG1, BlsG2
ap = argparse.ArgumentParser()
ap.add_argument("--curve", default="g1"); ap.add_argument("--log2n", type=int, default=20); ap.add_argument("--reps", type=int, default=12)
args = ap.parse_args()
dev = torch.device("cuda", 0); st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); s = st.cuda_stream
eng = get_engine(0)
cid, O = (BLS12_381_G1, BlsG1) if args.curve == "g1" else (BLS12_381_G2, BlsG2)
n = 1 << args.log2n
pts, ks = bench.gen_points(eng, cid, O, n, 0x1234567 + 7, 0x6789, dev, s)
sc = bench.gen_scalars(n, 254, 5, dev); sc[::17] = 0
sci = bench.scalars_to_ints(sc)
expect = O.BASE.multiplyUnsafe(sum(k * x for k, x in zip(ks, sci)) % BLS_R).toAffine()


def wall(f, reps=args.reps, warm=4):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return round(ts[0], 4), round(ts[len(ts) // 2], 4)


out = {"curve": args.curve, "log2n": args.log2n}
f_sync = lambda: eng.msm_dev(cid, n, pts.data_ptr(), sc.data_ptr(), s)
assert wire_to_affine(cid, f_sync()[0]) == expect
out["sync"] = wall(f_sync)
for P in (2, 3, 4):
    def f(P=P):
        for p in range(P):
            eng.msm_async_submit(p, cid, n, pts.data_ptr(), sc.data_ptr(), None, None, eng.async_part(p, P))
        slots = [eng.msm_async_collect_slot(p, cid) for p in range(P)]
        return eng.msm_shard_combine(cid, n, np.stack(slots), s)
    assert wire_to_affine(cid, f()[0]) == expect, P
    out["split%d" % P] = wall(f)
print(json.dumps(out))
