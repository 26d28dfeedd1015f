smi() { rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed -e 's/.*sclk clock level: [0-9]*: (\([0-9]*\)Mhz).*/sclk \1/' -e 's/.*Power (W): \([0-9.]*\).*/power \1/' | tr '\n' ' '; echo; }
for mb in 16 64 128 192 512 4096; do
  ./tools/ubench/power_probe 9 4 $mb &
  pid=$!
  sleep 1.8
  for i in 1 2 3; do echo "  span $mb MB: $(smi)"; sleep 0.5; done
  wait $pid
done
