#!/bin/bash
# tools/slow_box_hunt2.sh [force] : ONE gpurun call = one box.  The secp256k1 ladder's time with the clock / power the box holds under
# it (tools/clock_probe.sh), what the box sustains on pure vector arithmetic (tools/_build/sustain_valu) and - on a box of the slow
# kind (ladder >= 8.95 ms), or with `force` - the ladder variants of the A/B build side by side (tools/_build/libncg_ab.so, NCG_SECP_W:
# 243 = shipped: W = 4, table in device memory, 3 waves per SIMD; 42 = W = 4 table in LDS, 2 waves (no table traffic at all);
# 244 = 4 waves; 253 = W = 5), alternating twice.  Output: one block per box, appended by the caller to profiles/r06_slow_box_ab.txt.
out=$(bash tools/clock_probe.sh secp256k1 2>/dev/null)
echo "$out"
ms=$(echo "$out" | sed -n 's/.*secp256k1 \([0-9.]*\) ms.*/\1/p')
tools/_build/sustain_valu 2>&1 | sed 's/^/   /'
cyc=$(echo "$out" | sed -n 's/.*x sclk = \([0-9]*\)).*/\1/p')
# the slow kind: the ladder at 8.95 ms or more, or 2 % more cycles per step than the typical boxes' 19.4-19.7 k
if [ "$1" = force ] || python -c "import sys; sys.exit(0 if float('${ms:-0}') >= 8.95 or float('${cyc:-0}') >= 19900 else 1)"; then
  [ "$1" = force ] || echo "   SLOW BOX"
  for rep in 1 2; do for w in 243 253; do
    NCG_SECP_W=$w NCG_LIB=$PWD/tools/_build/libncg_ab.so timeout 300 python bench.py --workload secp256k1 --no-cpu-baseline --no-live-pmc --quick-verify --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   NCG_SECP_W=$w  %.3f ms' % d['ms_per_step'])"
  done; done
  # the ladder with its table entries prefetched through LDS (tools/_build/libncg_lpf.so: -DNCG_LADDER_PREFETCH=1) against the same
  # tree without (libncg_base.so), alternating three times
  if [ -f tools/_build/libncg_lpf.so ]; then
    for rep in 1 2 3; do for v in base lpf; do
      NCG_LIB=$PWD/tools/_build/libncg_$v.so timeout 300 python bench.py --workload secp256k1 --no-cpu-baseline --no-live-pmc --quick-verify --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   libncg_$v  %.3f ms' % d['ms_per_step'])"
    done; done
  fi
  for wl in msm_g1 ntt ed25519; do bash tools/clock_probe.sh $wl 2>/dev/null | sed 's/^/   /'; done
fi
