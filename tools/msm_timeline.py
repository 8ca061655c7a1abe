#!/usr/bin/env python3
"""Per-kernel timeline of the LAST MSM of each (curve, shape) in a rocprofv3 --kernel-trace CSV:
    python tools/msm_timeline.py <dir with *kernel_trace.csv> [--seq]
(--seq: every launch in order with its start offset and duration instead of the per-kernel sums.)  An MSM is the kernel sequence from k_msm_digits* to k_msm_tail*."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
msms, cur = [], None
for nm, t0, t1 in seq:
    short = nm.split("(")[0].replace("void ncg::", "").replace("ncg::", "")
    if "k_msm_digits" in nm or ("k_points_to_mont" in nm and cur is None):
        if cur is None:
            cur = {"t0": t0, "k": []}
    if cur is not None:
        cur["k"].append((short, t0, t1))
        if "k_msm_tail" in nm:
            cur["t1"] = t1
            msms.append(cur)
            cur = None
last = {}
for m in msms:
    tag = next((k[0] for k in m["k"] if "k_msm_accum" in k[0]), "?")
    grid = tuple(sorted(set(k[0] for k in m["k"])))
    last[(tag, len(m["k"]), round((m["t1"] - m["t0"]) / 2e4))] = m
for (tag, nk, _), m in last.items():
    print("%s  %d launches  span %.1f us" % (tag, nk, (m["t1"] - m["t0"]) / 1e3))
    if "--seq" in sys.argv:
        for nm, t0, t1 in m["k"]:
            print("   +%8.1f us  %-58s %9.1f us" % ((t0 - m["t0"]) / 1e3, nm[:58], (t1 - t0) / 1e3))
        continue
    agg = {}
    for nm, t0, t1 in m["k"]:
        a = agg.setdefault(nm, [0, 0.0]); a[0] += 1; a[1] += (t1 - t0) / 1e3
    busy = sum(v[1] for v in agg.values())
    for nm, (c, t) in agg.items():
        print("   %-58s x%-3d %9.1f us" % (nm[:58], c, t))
    print("   %-58s      %9.1f us" % ("(gaps between kernels)", (m["t1"] - m["t0"]) / 1e3 - busy))
