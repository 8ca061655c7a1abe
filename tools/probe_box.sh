#!/bin/bash
# tools/probe_box.sh [libB] : prints the secp256k1 step time of this box (fast boxes read about 8.9 ms, slow ones 10.1) and,
# when a second build is named, alternates it with the shipped one on the G2 MSM
t=$(timeout 300 python bench.py --workload secp256k1 --no-cpu-baseline --no-live-pmc --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))")
echo "secp256k1 2^20 step: $t ms"
bash tools/box_info.sh 2>&1 | grep -E "clock level|Power|Partition" | head -8
if [ -n "$1" ]; then bash tools/ab_lib.sh msm_g2 noble-curves_amd/libncg.so $1 | tail -4; fi
