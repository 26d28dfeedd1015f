#!/bin/bash
# PMC passes over the HardNet bench (development aid).  Usage: tools/pmc_hn.sh <outdir> "<counters>" ...
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/$1"; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out/pass$i" -- python3 "$root/tools/bench_hardnet.py" 16384 1 > /dev/null 2>"$out/pass$i.err"
  i=$((i+1))
done
