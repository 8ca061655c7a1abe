#!/bin/bash
# tools/ab_ntt.sh <lib>... : NTT timings (tools/bench_ntt.py) of alternative builds, alternating, on one box
mkdir -p gpurun_out
for rep in 1 2; do for lib in "$@"; do
  tag=$(basename $lib .so)
  NCG_LIB=$PWD/$lib timeout 300 python tools/bench_ntt.py --sizes 16,20,22,24 --steps 10 --out gpurun_out/ntt_${tag}_$rep.json > gpurun_out/ntt_${tag}_$rep.log 2>&1
  python - <<P
import json
d=json.load(open("gpurun_out/ntt_${tag}_$rep.json"))
print("$tag", $rep, {k: v["ms"] for k, v in d.items()})
P
done; done
