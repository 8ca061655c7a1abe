#!/bin/bash
# tools/slow_box_hunt.sh : quick G2 MSM on this box; if it is of the slow kind (> 3.8 ms), run the diagnostics while we have it:
# variant builds (LDS-DMA prefetch) and the wave-cycle / fetch-level counters.
rocm-smi --showserial 2>/dev/null | grep Serial
t=$(timeout 200 python bench.py --workload msm_g2 --no-cpu-baseline --no-live-pmc --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_msm'],3))")
echo "G2 MSM 2^18: $t ms"
if python -c "import sys; sys.exit(0 if float('$t') > 3.8 else 1)"; then
  echo "SLOW BOX"
  bash tools/probe_variants.sh 2>&1 | grep -E "msm_g"
  bash tools/pmc_stall.sh slowbox
fi
