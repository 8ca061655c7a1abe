#!/bin/bash
# tools/clock_probe.sh [workload] : the shader clock and package power the box sustains UNDER one workload of bench.py, sampled from
# sysfs (hwmon freq1_input / power1_average|power1_input, ~20 Hz) while 400 steps run, next to the time per step - what makes one box's
# ladder 7 % slower than another's (profiles/r06_box_to_box.md): a lower sustained clock, or the same clock and a slower memory path?
WL=${1:-secp256k1}
ser=$(rocm-smi --showserial 2>/dev/null | grep -o "Serial Number: .*" | head -1 | awk '{print $3}')
# the ONE GPU this container sees, by its PCI address (sysfs lists every card of the host)
bdf=$(rocm-smi --showbus 2>/dev/null | grep -o "[0-9a-fA-F]\{4\}:[0-9a-fA-F]\{2\}:[0-9a-fA-F]\{2\}\.[0-9]" | head -1)
H=$(ls -d /sys/bus/pci/devices/${bdf,,}/hwmon/hwmon* 2>/dev/null | head -1)
[ -z "$H" ] && H=$(ls -d /sys/bus/pci/devices/${bdf^^}/hwmon/hwmon* 2>/dev/null | head -1)
cap=$(cat $H/power1_cap 2>/dev/null)
echo "# device $bdf hwmon $H power cap $((${cap:-0} / 1000000)) W" >&2
S=$(mktemp)
( timeout 300 python bench.py --workload $WL --no-cpu-baseline --no-live-pmc --quick-verify --steps 400 --warmup 5 2>/dev/null | tail -1 > $S.line ) &
BP=$!
sleep 0.2
while kill -0 $BP 2>/dev/null; do
  f=$(cat $H/freq1_input 2>/dev/null); p=$(cat $H/power1_average 2>/dev/null || cat $H/power1_input 2>/dev/null); t=$(cat $H/temp2_input 2>/dev/null)
  echo "$(date +%s.%N) ${f:-0} ${p:-0} ${t:-0}" >> $S
  sleep 0.05
done
wait $BP
python - "$S" "$S.line" "$ser" "$WL" <<'P'
import sys, json
rows = [l.split() for l in open(sys.argv[1])]
line = open(sys.argv[2]).read().strip()
d = json.loads(line) if line else {}
ms = d.get("ms_per_step") or d.get("ms_per_msm") or d.get("ms_per_batch") or d.get("ms_per_transform")
f = sorted(float(r[1]) / 1e6 for r in rows); p = sorted(float(r[2]) / 1e6 for r in rows); t = sorted(float(r[3]) / 1e3 for r in rows)
# the loaded phase = samples above 60 % of the top power reading
hot = [r for r in rows if float(r[2]) / 1e6 > 0.6 * p[-1]]
hf = sorted(float(r[1]) / 1e6 for r in hot); hp = sorted(float(r[2]) / 1e6 for r in hot)
med = lambda v: v[len(v) // 2] if v else float("nan")
print("serial %s %s %.3f ms/step (x sclk = %.0f) | under load (%d of %d samples): sclk MHz min/med/max %.0f/%.0f/%.0f  power W min/med/max %.0f/%.0f/%.0f  temp max %.0f C"
      % (sys.argv[3], sys.argv[4], ms or float("nan"), (ms or 0) * med(hf), len(hot), len(rows), hf[0] if hf else 0, med(hf), hf[-1] if hf else 0, hp[0] if hp else 0, med(hp), hp[-1] if hp else 0, t[-1] if t else 0))
P
rm -f $S $S.line
