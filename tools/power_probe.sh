#!/bin/bash
# tools/power_probe.sh: run each mode of tools/ubench/power_probe for a few seconds and sample clock / power meanwhile
root="$(cd "$(dirname "$0")/.." && pwd)"
smi() { rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed -e 's/.*sclk clock level: [0-9]*: (\([0-9]*\)Mhz).*/sclk \1/' -e 's/.*Power (W): \([0-9.]*\).*/power \1/' | tr '\n' ' '; echo; }
echo "idle: $(smi)"
for mode in ${MODES:-0 1 2 3 4 5 6}; do
  "$root/tools/ubench/power_probe" $mode 4 &
  pid=$!
  sleep 1.5
  for i in 1 2 3 4; do echo "  mode $mode: $(smi)"; sleep 0.5; done
  wait $pid
done
