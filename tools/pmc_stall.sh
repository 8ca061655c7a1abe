#!/bin/bash
# tools/pmc_stall.sh <tag> : where the waves of the dominant kernels spend their cycles - issue, waiting on s_waitcnt, waiting for an
# instruction (fetch / arbitration) - and the instruction-fetch latency (SQ_IFETCH_LEVEL / SQ_IFETCH), per kernel.  Two counter
# passes, each alone with --kernel-trace (no other trace domain), on the short bench run.
TAG=${1:-stall}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocm-smi --showserial 2>/dev/null | grep Serial > $OUT/gpu.txt
BENCH_ARGS="--steps 4 --warmup 1 --no-cpu-baseline --no-live-pmc --quick-verify"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_wait -- python $GRAFT_REPO_ROOT/bench.py $BENCH_ARGS > $GRAFT_REPO_ROOT/$OUT/pmc_wait.log 2>&1); echo "wait rc=$?"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_ifetch -- python $GRAFT_REPO_ROOT/bench.py $BENCH_ARGS > $GRAFT_REPO_ROOT/$OUT/pmc_ifetch.log 2>&1); echo "ifetch rc=$?"
python tools/pmc_summary.py $OUT/pmc_wait $OUT/pmc_ifetch --out $OUT/stall.json 2>&1 | tail -1
find $OUT -name "*.csv" -size +3M -delete; find $OUT -name "*.db" -delete
python - <<PY
import json
d = json.load(open("$OUT/stall.json"))["kernels"]
rows = sorted(d.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]
print("%-52s %8s %8s %8s %8s %9s %9s" % ("kernel", "valu/wc", "wait/wc", "winst/wc", "busy", "ifetch_lat", "vmem_lat"))
for k, v in rows:
    wc = v.get("SQ_WAVE_CYCLES") or 1
    lat = v.get("SQ_IFETCH_LEVEL", 0) / max(1, v.get("SQ_IFETCH", 1))
    vl = v.get("SQ_INST_LEVEL_VMEM", 0) / max(1, v.get("SQ_INSTS_VMEM_RD", 1))
    print("%-52s %8.3f %8.3f %8.3f %8.3g %9.1f %9.1f" % (k.replace("ncg::", "")[:52], v.get("SQ_ACTIVE_INST_VALU", 0) / wc, v.get("SQ_WAIT_ANY", 0) / wc,
          v.get("SQ_WAIT_INST_ANY", 0) / wc, v.get("SQ_BUSY_CYCLES", 0), lat, vl))
PY
