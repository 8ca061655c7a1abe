#!/bin/bash
# tools/clock_probe_cmd.sh <command...> : shader clock / package power of the visible GPU (hwmon, ~20 Hz) while <command> runs; prints the
# command's output, then one line with the samples of the loaded phase (power above 60 % of the maximum seen).
bdf=$(rocm-smi --showbus 2>/dev/null | grep -o "[0-9a-fA-F]\{4\}:[0-9a-fA-F]\{2\}:[0-9a-fA-F]\{2\}\.[0-9]" | head -1)
H=$(ls -d /sys/bus/pci/devices/${bdf,,}/hwmon/hwmon* 2>/dev/null | head -1)
S=$(mktemp)
"$@" &
BP=$!
while kill -0 $BP 2>/dev/null; do
  echo "$(date +%s.%N) $(cat $H/freq1_input 2>/dev/null || echo 0) $(cat $H/power1_input 2>/dev/null || echo 0)" >> $S
  sleep 0.05
done
wait $BP
python - "$S" <<'P'
import sys
rows = [l.split() for l in open(sys.argv[1])]
p = sorted(float(r[2]) / 1e6 for r in rows)
hot = [r for r in rows if float(r[2]) / 1e6 > 0.6 * p[-1]]
f = sorted(float(r[1]) / 1e6 for r in hot); w = sorted(float(r[2]) / 1e6 for r in hot)
print("under load (%d of %d samples): sclk MHz min/med/max %.0f/%.0f/%.0f  power W min/med/max %.0f/%.0f/%.0f" % (len(hot), len(rows), f[0], f[len(f) // 2], f[-1], w[0], w[len(w) // 2], w[-1]))
P
rm -f $S
