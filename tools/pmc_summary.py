#!/usr/bin/env python3
"""Summarises rocprofv3 counter-collection passes (one directory per pass) into one JSON:
per kernel, the average of every collected counter per launch, the average launch duration from
the pass's kernel trace, and the derived figures DESIGN.md section 7 defines:

  valu_wave_insts_per_launch      SQ_INSTS_VALU (wave-level VALU instructions issued)
  valu_lane_ops_per_s             SQ_INSTS_VALU * 64 / duration
  (valu_issue_frac, the lane-op rate over 16 lanes per clock and SIMD, was dropped in round 6: gfx950 issues a plain wave64
   instruction in under 4 cycles - 2.89 measured - so the figure could pass 1; bench.py reports mad_frac / plain_frac instead)
  valu_busy_frac                  SQ_ACTIVE_INST_VALU * 4 / (SQ_BUSY_CYCLES-normalised SIMD cycles), raw
                                  counters kept alongside so the formula can be re-derived
  hbm_bytes_per_launch            FETCH_SIZE / WRITE_SIZE (KB) scaled by the calibration factors measured
                                  with tools/pmc_calib (bytes moved / counter bytes, per access pattern)

Usage (each pass on the GPU box, counters never combined with trace domains other than --kernel-trace):
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d gpurun_out/pmc_valu -- python bench.py ...
  python tools/pmc_summary.py gpurun_out/pmc_valu gpurun_out/pmc_fetch ... --out profiles/r02_pmc.json
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

VALU_PEAK_LANE_OPS = 256 * 4 * 16 * 2.4e9


def short(name):
    return name.replace("void ", "").split("(")[0]


def read_pass(d):
    """-> ({kernel: {counter: [per-dispatch sums]}}, {kernel: [durations ns]})"""
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = short(r.get("Kernel_Name") or "")
            try:
                per[name][r.get("Counter_Name")][r.get("Dispatch_Id")] += float(r.get("Counter_Value") or 0)
            except ValueError:
                pass
    counters = {k: {c: list(v.values()) for c, v in cs.items()} for k, cs in per.items()}
    durs = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            try:
                durs[short(r.get("Kernel_Name") or "")].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
            except (KeyError, ValueError):
                pass
    return counters, durs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--out", required=True)
    ap.add_argument("--calib", default=None, help="JSON written by an earlier run over tools/pmc_calib passes")
    ap.add_argument("--prefix", default="ncg::", help="keep kernels whose name starts with this ('' = all)")
    ap.add_argument("--note", default="")
    args = ap.parse_args()
    kernels = defaultdict(dict)
    for d in args.dirs:
        counters, durs = read_pass(d)
        for k, cs in counters.items():
            if args.prefix and not k.startswith(args.prefix) and not k.startswith("calib_"):
                continue
            e = kernels[k]
            for c, vals in cs.items():
                e[c] = sum(vals) / len(vals)
                e["launches_" + c] = len(vals)
            if durs.get(k):
                # duration under THIS pass's counters (profiled launches run slower than bare ones)
                key = "avg_ms_in_pass_" + "+".join(sorted(cs))[:60]
                e[key] = sum(durs[k]) / len(durs[k]) / 1e6
                if "SQ_INSTS_VALU" in cs:
                    e["avg_ms_valu_pass"] = e[key]
    calib = {}
    if args.calib and os.path.exists(args.calib):
        calib = json.load(open(args.calib)).get("calibration", {})
    out = {"note": ("rocprofv3 --pmc passes (counters alone with --kernel-trace); values are averages per launch summed "
                    "over XCDs / SEs. " + args.note), "kernels": {}, "calibration": {}}
    BYTES = 2 << 30
    for k, e in kernels.items():
        if k.startswith("calib_"):
            c = {}
            if "FETCH_SIZE" in e and "read" in k or "gather" in k and "FETCH_SIZE" in e:
                c["fetch_bytes_per_counter_byte"] = BYTES / (e["FETCH_SIZE"] * 1024)
            if "WRITE_SIZE" in e and "write" in k:
                c["write_bytes_per_counter_byte"] = BYTES / (e["WRITE_SIZE"] * 1024)
            c.update({x: e[x] for x in ("FETCH_SIZE", "WRITE_SIZE") if x in e})
            out["calibration"][k] = c
            continue
        if "SQ_INSTS_VALU" in e and e.get("avg_ms_valu_pass"):
            sec = e["avg_ms_valu_pass"] * 1e-3
            e["valu_lane_ops_per_s"] = e["SQ_INSTS_VALU"] * 64 / sec
        if "SQ_ACTIVE_INST_VALU" in e and e.get("SQ_BUSY_CYCLES"):
            e["valu_busy_raw_ratio"] = e["SQ_ACTIVE_INST_VALU"] / e["SQ_BUSY_CYCLES"]
        if "SQ_WAVE_CYCLES" in e and e.get("SQ_ACTIVE_INST_VALU"):
            e["valu_active_over_wave_cycles"] = e["SQ_ACTIVE_INST_VALU"] / e["SQ_WAVE_CYCLES"]
        out["kernels"][k] = e
    if not out["calibration"]:
        out["calibration"] = calib
    cal = out["calibration"]
    fg = cal.get("calib_gather64", {}).get("fetch_bytes_per_counter_byte")
    fs = cal.get("calib_stream_read16", {}).get("fetch_bytes_per_counter_byte")
    wg = cal.get("calib_write64", {}).get("write_bytes_per_counter_byte")
    # the same per-pattern factors bench.py applies (FETCH_FACTOR / KERNELS there): kernels that stream 16 bytes per lane read
    # FETCH_SIZE in units that count each byte half (factor 2.0), gather-dominated kernels (item-major 64-byte records through an
    # index) one to one (0.99) - tools/pmc_calib.hip
    STREAMING = ("k_ntt_pass", "k_points_to_mont", "k_msm_digits", "k_msm_hist", "k_msm_bucket_totals", "k_msm_scan", "k_encode", "k_ntt_fill_table")
    for k, e in out["kernels"].items():
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            stream = any(t in k for t in STREAMING)
            f = (fs or 2.0) if stream else (fg or fs or 2.0)
            w = wg or 1.0
            e["hbm_bytes_per_launch"] = int(e["FETCH_SIZE"] * 1024 * f + e["WRITE_SIZE"] * 1024 * w)
            e["hbm_bytes_factors"] = {"fetch": f, "write": w, "pattern": "stream" if stream else "gather"}
    json.dump(out, open(args.out, "w"), indent=1, sort_keys=True)
    print("%d kernels -> %s" % (len(out["kernels"]), args.out))


if __name__ == "__main__":
    main()
