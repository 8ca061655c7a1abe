#!/bin/bash
# tools/ab_lib.sh <workload> <libA> <libB> : alternate two builds on one bench workload
W=$1; shift
for rep in 1 2; do for lib in "$@"; do
  NCG_LIB=$PWD/$lib timeout 300 python bench.py --workload $W --no-cpu-baseline --no-live-pmc --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('resident_subgroup_set') or {}; print('$W', '$lib'.split('/')[-1], round(d.get('ms_per_msm') or d['ms_per_step'],3), round(r.get('ms_per_msm',0),3))"
done; done
