#!/bin/bash
# tools/ab_extra.sh <libA> <libB> : tools/bench_extra.py (every entry point beside the bench line) with two builds, side by side
mkdir -p gpurun_out
for lib in "$@"; do
  tag=$(basename $lib .so)
  NCG_LIB=$PWD/$lib timeout 900 python tools/bench_extra.py --out gpurun_out/extra_$tag.json > gpurun_out/extra_$tag.log 2>&1 || tail -3 gpurun_out/extra_$tag.log
done
python - "$@" <<'P'
import json, os, sys
tags = [os.path.basename(x)[:-3] for x in sys.argv[1:]]
d = [json.load(open("gpurun_out/extra_%s.json" % t)) for t in tags]
for k in d[0]:
    if k in d[1] and isinstance(d[0][k], dict) and "ms" in d[0][k]:
        a, b = d[0][k]["ms"], d[1][k]["ms"]
        print("%-52s %9.3f %9.3f  %+5.1f %%" % (k, a, b, (b - a) / a * 100))
P
