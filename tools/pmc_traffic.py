#!/usr/bin/env python3
"""Builds profiles/r01_pmc_traffic.json from rocprofv3 counter-collection output.

Usage (on the GPU box, counters in their own passes - never combined with trace domains other
than --kernel-trace):
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python bench.py ...
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python bench.py ...
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write --out gpurun_out/pmc_traffic.json
Values are KB per launch averaged over launches; corrected bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024
as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts half of wide coalesced reads)."""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict


def collect(d):
    """{kernel: {counter: (avg per launch, launches)}} plus the same split by grid size under
    the key (kernel, grid)."""
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name") or r.get("Kernel-Name") or ""
            cn = r.get("Counter_Name") or ""
            try:
                val = float(r.get("Counter_Value") or 0)
            except ValueError:
                continue
            acc[name][(cn, r.get("Dispatch_Id"), r.get("Grid_Size"))].append(val)
    out = defaultdict(dict)
    for name, per in acc.items():
        by_counter, by_grid = defaultdict(list), defaultdict(list)
        for (cn, _, grid), vals in per.items():
            by_counter[cn].append(sum(vals))          # sum over XCD / instance rows of one dispatch
            by_grid[(cn, grid)].append(sum(vals))
        for cn, vals in by_counter.items():
            out[name][cn] = (sum(vals) / len(vals), len(vals))
        for (cn, grid), vals in by_grid.items():
            out[(name, grid)][cn] = (sum(vals) / len(vals), len(vals))
    return out


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--out", required=True)
    ap.add_argument("--note", default="")
    args = ap.parse_args()
    merged, grids = defaultdict(dict), defaultdict(lambda: defaultdict(dict))
    for d in args.dirs:
        for name, cs in collect(d).items():
            for cn, (avg, n) in cs.items():
                if isinstance(name, tuple):
                    grids[short(name[0])][name[1]][cn + "_KB"] = round(avg, 1)
                    grids[short(name[0])][name[1]]["launches_" + cn + "_KB"] = n
                else:
                    merged[short(name)][cn + "_KB"] = round(avg, 1)
                    merged[short(name)]["launches_" + cn + "_KB"] = n

    def corrected(v):
        if "FETCH_SIZE_KB" in v and "WRITE_SIZE_KB" in v:
            v["hbm_bytes_per_launch_corrected"] = int((2 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024)
    kernels = {}
    for name, v in merged.items():
        if not name.startswith("ncg::"):
            continue
        corrected(v)
        if len(grids[name]) > 1:                       # launches of different sizes: keep the split
            for g in grids[name].values():
                corrected(g)
            v["by_grid"] = dict(grids[name])
        kernels[name] = v
    note = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; values are KB per launch averaged over "
            "launches; corrected bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md (FETCH_SIZE counts half "
            "of wide coalesced reads on gfx950; other access widths are uncalibrated). " + args.note)
    json.dump({"note": note, "kernels": kernels}, open(args.out, "w"), indent=1)
    print("%d kernels -> %s" % (len(kernels), args.out))


if __name__ == "__main__":
    main()
