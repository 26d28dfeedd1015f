#!/bin/bash
# Energy per byte by where it comes from (development aid): tools/ubench/power_probe mode 9 -- every CU streams
# global_load_dwordx4 -- over windows of 16 MB ... 4 GB, clock and package power sampled beside it.  A window up to ~200 MB lives in
# the Infinity Cache (profiles/r5_mall.txt: 7.5-7.9 TB/s at ~700 W = 21 pJ/B above the clocked floor), 4 GB comes from HBM (4.7 TB/s
# at ~990 W = 96 pJ/B).  Usage: bash tools/power_mall.sh   (from the repository root, after building tools/ubench/power_probe)
smi() { rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed -e 's/.*sclk clock level: [0-9]*: (\([0-9]*\)Mhz).*/sclk \1/' -e 's/.*Power (W): \([0-9.]*\).*/power \1/' | tr '\n' ' '; echo; }
for mb in 16 64 128 192 512 4096; do
  ./tools/ubench/power_probe 9 4 $mb &
  pid=$!
  sleep 1.8
  for i in 1 2 3; do echo "  span $mb MB: $(smi)"; sleep 0.5; done
  wait $pid
done
