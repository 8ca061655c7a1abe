#!/usr/bin/env python3
"""Strong-scaling budget of ONE 2^20-point MSM, measured on one GPU.

* `sync`:            one MSM at a time (ncg_msm_dev / ncg_msm_resident_dev), wall clock per MSM.
* `pipelined`:       ncg_msm_async_submit / _collect over L lanes: the steady-state time per MSM when the dependent tail
                     of one MSM overlaps the sort / accumulate kernels of the next.
* `share(G)`:        what ONE of G ranks executes in the window-sharded mode - part 0 of G through
                     ncg_msm_shard_windows_local_dev (latency form) and every part in turn through the lanes
                     (NCG_MSM_ASYNC_PART; steady state, average over the G parts), plus the combine / finish every rank
                     runs on the G gathered slots (ncg_msm_shard_combine: upload of the slots instead of the xGMI
                     all-gather, header check, concatenation, host Horner).  The G slots of one MSM are combined and
                     the result is checked against the progression identity, so the timed code is the shipped path.
* `point_share(G)`:  the same for the point-sharded mode (a 2^20 / G-point MSM at the shard plan), for comparison.
Not included: the all-gather itself (8 x 29 KB over xGMI).
    python tools/msm_share.py [--curve g1] [--log2n 20] [--reps 10] [--out file.json]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, bench
from helpers import wire_to_affine
from noble_curves_amd import get_engine
from noble_curves_amd._native import BLS12_381_G1, BLS12_381_G2
from oracle.curves import BLS_R, BlsG1, BlsG2

ap = argparse.ArgumentParser()
ap.add_argument("--curve", default="g1")
ap.add_argument("--log2n", type=int, default=20)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--parts", default="2,4,8")
ap.add_argument("--out", default=None)
args = ap.parse_args()
dev = torch.device("cuda", 0); st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); s = st.cuda_stream
eng = get_engine(0)
cid, O = (BLS12_381_G1, BlsG1) if args.curve == "g1" else (BLS12_381_G2, BlsG2)
n = 1 << args.log2n
pts, ks = bench.gen_points(eng, cid, O, n, 0x1234567 + 7, 0x6789, dev, s)
sc = bench.gen_scalars(n, 254, 5, dev); sc[::17] = 0
sci = bench.scalars_to_ints(sc)
expect = O.BASE.multiplyUnsafe(sum(k * x for k, x in zip(ks, sci)) % BLS_R).toAffine()
L = eng.msm_async_lanes()
res = {"curve": args.curve, "log2n": args.log2n, "lanes": L, "plan": eng.msm_plan_info(cid, n)}


def wall(f, reps=args.reps, warm=3):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return {"min_ms": round(ts[0], 4), "median_ms": round(ts[len(ts) // 2], 4)}


def pipelined(submit, collect, depth, jobs):
    """`jobs` submissions over `depth` lanes, collecting lane i just before it is reused; returns ms per job (steady state)."""
    for warm in (True, False):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = None
        for i in range(jobs):
            lane = i % depth
            if i >= depth: last = collect(lane)
            submit(lane, i)
        for i in range(jobs, jobs + depth): last = collect(i % depth)
        dt = (time.perf_counter() - t0) * 1e3
    return dt / jobs, last


for kind in ("generic", "resident", "precomputed", "verified", "verified_precomp"):
    rs = None
    if kind != "generic":
        rs = eng.upload_points(cid, pts.cpu().numpy())
        if kind.startswith("verified"): assert rs.verify_subgroup() == -1
        if kind.endswith("precomputed") or kind.endswith("precomp"): assert rs.precompute()
    row = {}
    P = 0 if rs is not None else pts.data_ptr()
    f_sync = (lambda: rs.msm_dev(sc.data_ptr(), s)) if rs is not None else (lambda: eng.msm_dev(cid, n, pts.data_ptr(), sc.data_ptr(), s))
    got = f_sync(); assert wire_to_affine(cid, got[0]) == expect, kind
    row["sync"] = wall(f_sync)
    for depth in (1, 2, 3, 4):
        if depth > L: break
        ms, last = pipelined(lambda lane, i: eng.msm_async_submit(lane, cid, n, P, sc.data_ptr(), None, rs),
                             lambda lane: eng.msm_async_collect(lane, cid), depth, 6 * depth + 6)
        assert wire_to_affine(cid, last[0]) == expect, (kind, depth)
        row["pipelined_depth%d_ms" % depth] = round(ms, 4)
    for G in [int(x) for x in args.parts.split(",")]:
        sh = {}
        slots = [eng.msm_shard_windows_local_dev(cid, n, r, G, P, sc.data_ptr(), s, rs) for r in range(G)]
        stack = np.stack(slots)
        got = eng.msm_shard_combine(cid, n, stack, s)
        assert wire_to_affine(cid, got[0]) == expect, (kind, G)
        sh["local_part0_sync"] = wall(lambda: eng.msm_shard_windows_local_dev(cid, n, 0, G, P, sc.data_ptr(), s, rs))
        sh["local_last_part_sync"] = wall(lambda: eng.msm_shard_windows_local_dev(cid, n, G - 1, G, P, sc.data_ptr(), s, rs))
        sh["combine_finish"] = wall(lambda: eng.msm_shard_combine(cid, n, stack, s))
        sh["latency_ms"] = round(sh["local_part0_sync"]["median_ms"] + sh["combine_finish"]["median_ms"], 4)
        for depth in (2, 3, 4):
            if depth > L: break
            got_slots = {}
            def submit(lane, i, G=G):
                eng.msm_async_submit(lane, cid, n, P, sc.data_ptr(), None, rs, eng.async_part(i % G, G))
                submit.part[lane] = i % G
            submit.part = {}
            def collect(lane):
                sl = eng.msm_async_collect_slot(lane, cid)
                got_slots[submit.part[lane]] = sl
                return sl
            ms, _ = pipelined(submit, collect, depth, 4 * G + 2 * depth)
            got = eng.msm_shard_combine(cid, n, np.stack([got_slots[r] for r in range(G)]), s)
            assert wire_to_affine(cid, got[0]) == expect, (kind, G, depth)
            sh["pipelined_part_depth%d_ms" % depth] = round(ms, 4)
        row["share%d" % G] = sh
        if kind == "generic":
            m = n // G
            f = lambda: eng.msm_shard_local_dev(cid, m, pts.data_ptr(), sc.data_ptr(), s, m)
            row["point_share%d_local_sync" % G] = wall(f)
    res[kind] = row
    print(kind, json.dumps(row), flush=True)
    if rs is not None: rs.free()
print(json.dumps(res))
if args.out:
    json.dump(res, open(args.out, "w"), indent=1)
