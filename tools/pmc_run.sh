#!/bin/bash
# PMC passes over a short bench run (development aid).  Usage: [PREC=fp16|fp32] tools/pmc_run.sh <outdir> "<counters>" ...
# One rocprofv3 --pmc pass per counter list (with --kernel-trace only), 16 images per launch at 1088x1920 (one micro-batch).
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/$1"; shift
prec="${PREC:-fp16}"
mkdir -p "$out"
IMAGES=${PMC_IMAGES:-16}
# the shape the counters belong to, recorded next to them (tools/pmc_json.py copies it into the profile, bench.py checks it)
echo "{\"images_per_launch\": $IMAGES, \"hp\": 1088, \"wp\": 1920}" > "$out/shape.json"
cd /tmp && export TMPDIR=/tmp
export BALF_FP16_CHECK=0      # (the one-off split-f16 range check of a new checkpoint would add three tiny dispatches per kernel to the averages)
i=0
for ctrs in "$@"; do
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out/pass$i" -- python3 "$root/bench.py" --steps 1 --warmup 1 --batch-per-gpu $IMAGES --cpu-images 0 --other-steps 0 --other-configs 0 --no-single-rank-collective --sustained-seconds 0 --host-fed-steps 0 --precision "$prec" ${PMC_EXTRA_ARGS:-} > /dev/null 2>"$out/pass$i.err"
  i=$((i+1))
done
python3 "$root/tools/pmc_summary.py" "$out" > "$out/summary.txt"
