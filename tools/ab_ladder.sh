#!/bin/bash
# tools/ab_ladder.sh <libA> <libB> [reps] : the secp256k1 batch multiply of two builds, alternating, 300 untimed-verification steps each (tools/ladder_time.py),
# with the clock / power the box held under each (tools/clock_probe_cmd.sh) - only worth reading on a box whose repeats agree to 0.3 %
rocm-smi --showserial 2>/dev/null | grep -o "Serial Number: .*" | head -1
R=${3:-4}
for rep in $(seq 1 $R); do for lib in $1 $2; do
  NCG_LIB=$PWD/$lib bash tools/clock_probe_cmd.sh python tools/ladder_time.py 300 2>&1 | grep -E "ms per step|under load" | tr '\n' ' ' | sed "s|$PWD/tools/_build/||; s|secp256k1 2^20 batch multiply: ||; s|per step over 300 steps ||; s|under load ([0-9 of]* samples): ||"; echo
done; done
