"""Key figures of a bench.py --out file, one line per workload (A/B tooling)."""
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    print("==", f)
    if d.get("metric", "").startswith("secp"):
        print("  secp256k1 ms_per_step %.3f  end_to_end %.3f" % (d["ms_per_step"], (d.get("end_to_end") or {}).get("ms_per_batch", float("nan"))))
    ex = d.get("extra", {})
    for k in ("msm_g1", "msm_g2"):
        e = ex.get(k) or (d if d.get("metric", "").endswith(k[-2:] + "_msm_points_per_sec") else None)
        if not e:
            continue
        ws = e.get("window_share", {})
        print("  %s sync %.3f  e2e %.3f  pipelined %s  share(lat,inflight) %s  resident %.3f precomp %.3f" % (
            k, e["ms_per_msm"], (e.get("end_to_end") or {}).get("ms_per_msm", float("nan")),
            {p: round(v["ms_per_msm"], 3) for p, v in e.get("pipelined", {}).items() if p.startswith("depth")},
            {g: (round(v["latency_ms"], 3), round(v["pipelined_part_ms"], 3)) for g, v in ws.items() if g.startswith("G")},
            (e.get("resident_subgroup_set") or {}).get("ms_per_msm", float("nan")),
            ((e.get("resident_subgroup_set") or {}).get("precomputed") or {}).get("ms_per_msm", float("nan"))))
    for k in ("ed25519_verify", "ntt_fr"):
        e = ex.get(k)
        if e:
            print("  %s %s" % (k, {kk: round(v, 4) for kk, v in e.items() if isinstance(v, float) and "ms" in kk}))
