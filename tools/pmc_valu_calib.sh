#!/bin/bash
# tools/pmc_valu_calib.sh <tag> : what the SQ VALU counters read on kernels whose instruction streams are KNOWN (tools/valu_rates.hip:
# 8192 identical wave-instructions per wave between two s_memtime reads) - which counter counts v_mad_u64_u32 (SQ_INSTS_VALU_INT64 /
# _INT32 / neither), and what VALUBusy = 4 * SQ_ACTIVE_INST_VALU / (SIMDs * GRBM_GUI_ACTIVE) reads on a pipe that is saturated by
# multiply-adds or by plain adds at 1, 2, 4, 8 waves per SIMD.  Counters alone with --kernel-trace, as the guide prescribes.
TAG=${1:-valu_calib}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
[ -x tools/_build/valu_rates ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o tools/_build/valu_rates
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
   --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc -- $GRAFT_REPO_ROOT/tools/_build/valu_rates > $GRAFT_REPO_ROOT/$OUT/valu_rates.jsonl 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err); echo "rc=$?"
python - <<PY
import csv, glob, json, collections
per = collections.defaultdict(lambda: collections.defaultdict(dict))
order = []
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Dispatch_Id"]))
        per[k[0]][k[1]][r["Counter_Name"]] = per[k[0]][k[1]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
rows = []
for name, disp in per.items():
    ids = sorted(disp)
    # run<KIND>: warm-up + timed launch for W = 1, 2, 4, 8 -> 8 dispatches; keep the timed ones
    for j, did in enumerate(ids):
        if j % 2 == 0:
            continue
        c = disp[did]
        waves = c.get("SQ_WAVES", 0)
        rows.append({"kernel": name, "waves_per_simd": int(round(waves / 1024)) if waves else None,
                     "insts_valu_per_wave": c.get("SQ_INSTS_VALU", 0) / max(1, waves),
                     "int64_per_wave": c.get("SQ_INSTS_VALU_INT64", 0) / max(1, waves), "int32_per_wave": c.get("SQ_INSTS_VALU_INT32", 0) / max(1, waves),
                     "valu_busy": 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0) / (1024.0 * max(1.0, c.get("GRBM_GUI_ACTIVE", 0))),
                     "active_valu_over_wave_cycles": c.get("SQ_ACTIVE_INST_VALU", 0) / max(1.0, c.get("SQ_WAVE_CYCLES", 0)),
                     "gui_active_cycles": c.get("GRBM_GUI_ACTIVE", 0), "active_quad_per_inst": c.get("SQ_ACTIVE_INST_VALU", 0) / max(1.0, c.get("SQ_INSTS_VALU", 0))})
json.dump({"what": "tools/pmc_valu_calib.sh: SQ VALU counters on tools/valu_rates.hip (8192 known wave-instructions per wave)", "rows": rows}, open("$OUT/valu_calib.json", "w"), indent=1)
for r in rows:
    if r["kernel"].endswith("<0>") or r["kernel"].endswith("<2>") or r["kernel"].endswith("<17>") or r["kernel"].endswith("<4>"):
        print(r)
PY
find $OUT -name "*.csv" -size +3M -delete; find $OUT -name "*.db" -delete
