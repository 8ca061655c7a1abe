"""The ONE line bench.py prints on stdout: a compact (< 6 KB) view of the full result object.

The driver keeps an 8 KB tail of stdout and parses the last line (a 36.7 KB line came back
`parsed: null`).  The full object - step lists, notes, every sub-measurement - goes to `--out`
(default gpurun_out/bench_full.json); this module cuts it down to the contract's keys plus one flat
block per workload.  tests/test_bench_line.py builds the compact line from a committed full object and
asserts its size and keys.
"""
import json

LIMIT = 6144          # bytes; the driver's tail is 8 KB


def _r(x, sig=6):
    """floats to `sig` significant digits (the line is a record, not a checkpoint); ints and the rest untouched"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (sig, x))
    return x


def _pick(d, keys):
    return {k: _r(d[k]) for k in keys if isinstance(d, dict) and k in d and not isinstance(d[k], (dict, list))}


def _cut(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 1].rstrip() + "~"


def roofline_block(rf, full=True):
    if not isinstance(rf, dict):
        return None
    out = _pick(rf, ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel_ms"))
    if full and "kernel" in rf:
        out["kernel"] = _cut(rf["kernel"], 96)
    v = rf.get("valu")
    if isinstance(v, dict):
        out["valu"] = _pick(v, ("mad_frac", "executed_mad_per_s", "mad_peak_per_s") + (("plain_frac", "pipe_demand_sum", "issue_frac") if full else ()))
    return out


def cpu_block(cb, full=True):
    if not isinstance(cb, dict):
        return None
    out = _pick(cb, ("value", "unit", "cores", "kind") + (("cpu_model", "logical_cores") if full else ()))
    if "sample" in cb:
        out["sample"] = _cut(cb["sample"], 110 if full else 60)
    p = cb.get("port")
    if isinstance(p, dict):
        out["port"] = _pick(p, ("value", "cores"))
        at = p.get("all_threads")
        if isinstance(at, dict):
            out["port"]["all_threads"] = _pick(at, ("value", "cores"))
    return out


def _dist3(st):
    """{"event_ms": {"min","median","max","list"}, ...} -> {"event_ms": [min, median, max], ...}"""
    if not isinstance(st, dict):
        return None
    out = {}
    for k, v in st.items():
        if isinstance(v, dict) and "median" in v:
            out[k] = [_r(v.get("min"), 5), _r(v.get("median"), 5), _r(v.get("max"), 5)]
    return out or None


def msm_block(e):
    out = _pick(e, ("value", "unit", "ms_per_msm", "total_points", "points_per_gpu", "scaling", "mode", "rccl_ranks", "world",
                    "ms_per_msm_n1", "speedup_vs_n1"))
    wp = e.get("window_plan")
    if isinstance(wp, dict):
        out["plan"] = _pick(wp, ("c", "nwin", "nb"))
    rf = roofline_block(e.get("roofline"), full=False)
    if rf:
        out["roofline"] = rf
    cb = cpu_block(e.get("cpu_baseline"), full=False)
    if cb:
        out["cpu_baseline"] = cb
    ps = e.get("power_state")
    if isinstance(ps, dict) and isinstance(ps.get("sclk_mhz"), list):
        out["sclk_mhz_med"] = _r(ps["sclk_mhz"][1], 5)
        out["power_w_med"] = _r((ps.get("power_w") or [None, None])[1], 5)
    ee = e.get("end_to_end")
    if isinstance(ee, dict) and "ms_per_msm" in ee:
        out["end_to_end_ms"] = _r(ee["ms_per_msm"])
    pl = e.get("pipelined")
    if isinstance(pl, dict):
        best = None
        for k, v in pl.items():
            if isinstance(v, dict) and "ms_per_msm" in v and (best is None or v["ms_per_msm"] < best[1]["ms_per_msm"]):
                best = (k, v)
        if best is None and "ms_per_msm" in pl:
            best = ("depth%s" % pl.get("depth", ""), pl)
        if best:
            out["in_flight"] = {"ms_per_msm": _r(best[1]["ms_per_msm"]), "lanes": best[0]}
            if "speedup_vs_n1" in best[1]:
                out["in_flight"]["speedup_vs_n1"] = _r(best[1]["speedup_vs_n1"], 4)
    bp = e.get("by_points")
    if isinstance(bp, dict) and "ms_per_msm" in bp:
        out["by_points_ms"] = _r(bp["ms_per_msm"])
    ws = e.get("window_share")
    if isinstance(ws, dict):
        sh = {}
        for g, v in ws.items():
            if isinstance(v, dict) and "latency_ms" in v:
                sh[g] = [_r(v["latency_ms"], 4), _r(v.get("pipelined_part_ms"), 4)]
        if sh:
            out["share_ms_latency_inflight_emulated_1gpu"] = sh
    return out


def batch_block(e, ms_key):
    out = _pick(e, ("value", "unit", ms_key, "log2n", "sigs_per_gpu"))
    rf = roofline_block(e.get("roofline"), full=False)
    if rf:
        out["roofline"] = rf
    cb = cpu_block(e.get("cpu_baseline"), full=False)
    if cb:
        out["cpu_baseline"] = cb
    ko = e.get("kernel_only")
    if isinstance(ko, dict) and "ms_per_batch" in ko:
        out["kernel_only_ms"] = _r(ko["ms_per_batch"])
    ps = e.get("power_state")
    if isinstance(ps, dict) and isinstance(ps.get("sclk_mhz"), list):
        out["sclk_mhz_med"] = _r(ps["sclk_mhz"][1], 5)
        out["power_w_med"] = _r((ps.get("power_w") or [None, None])[1], 5)
    return out


def compact_line(full, full_path=None):
    """the driver-facing object; every key of the bench contract, `roofline`, `cpu_baseline`, one flat block per workload"""
    out = {}
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype"):
        if k in full:
            out[k] = _r(full[k], 8)
    for k in ("ms_per_msm", "ms_per_batch", "ms_per_transform", "total_points", "mode"):   # single-workload runs (tools/ab_*.sh)
        if k in full and not isinstance(full[k], (dict, list)):
            out[k] = _r(full[k])
    rs = full.get("resident_subgroup_set")
    if isinstance(rs, dict) and "ms_per_msm" in rs:
        out["resident_subgroup_set"] = {"ms_per_msm": _r(rs["ms_per_msm"])}
    out["data"] = _cut(full.get("data", "synthetic"), 100)
    cfg = full.get("config")
    if isinstance(cfg, dict):
        out["config"] = {k: (_cut(v, 100) if isinstance(v, str) else v) for k, v in cfg.items()}
    st = _dist3(full.get("step_times"))
    if st:
        out["step_ms_min_med_max"] = st
    rf = roofline_block(full.get("roofline"))
    if rf:
        out["roofline"] = rf
    cb = cpu_block(full.get("cpu_baseline"))
    if cb:
        out["cpu_baseline"] = cb
    ee = full.get("end_to_end")
    if isinstance(ee, dict) and "ms_per_batch" in ee:
        out["end_to_end_ms"] = _r(ee["ms_per_batch"])
    ps = full.get("power_state")
    if isinstance(ps, dict) and isinstance(ps.get("sclk_mhz"), list):
        out["power_state"] = {"sclk_mhz_med": _r(ps["sclk_mhz"][1], 5), "power_w_med": _r((ps.get("power_w") or [None, None])[1], 5),
                              "ms_per_step_x_sclk": _r(ps.get("ms_per_step_x_sclk"), 5)}
    extra = full.get("extra") or {}
    for key in ("msm_g1", "msm_g1_strong", "msm_g2", "msm_g2_strong"):
        e = extra.get(key) or full.get(key)
        if isinstance(e, dict):
            out[key] = msm_block(e)
    if isinstance(extra.get("ed25519_verify"), dict):
        out["ed25519"] = batch_block(extra["ed25519_verify"], "ms_per_batch")
    if isinstance(extra.get("ntt_fr"), dict):
        out["ntt"] = batch_block(extra["ntt_fr"], "ms_per_transform")
    c0 = extra.get("configs0_point_multiply")
    if isinstance(c0, dict):
        b = {"unit": "ops/s"}
        ref = c0.get("reference")
        if isinstance(ref, dict):
            b["reference"] = _pick(ref, ("Point_mul", "Point_mulUns"))
        b["port"] = {k: _r(c0[k]["value"]) for k in ("Point_multiply", "Point_multiplyUnsafe")
                     if isinstance(c0.get(k), dict) and "value" in c0[k]}
        out["configs0_cpu"] = b
    for k in ("pmc", "dist_dry_run"):
        if k in full:
            out[k] = _cut(full[k], 60)
    if full_path:
        out["full"] = full_path
    # last resort, never expected: drop the widest optional blocks until the line fits
    for victim in ("configs0_cpu", "step_ms_min_med_max", "ntt", "ed25519", "msm_g2_strong", "msm_g2"):
        if len(json.dumps(out, separators=(",", ":"))) <= LIMIT:
            break
        out.pop(victim, None)
    return out


def dumps(obj):
    return json.dumps(obj, separators=(",", ":"))


if __name__ == "__main__":
    import sys
    full = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    line = dumps(compact_line(full))
    print(line)
    print(len(line), "bytes", file=sys.stderr)
