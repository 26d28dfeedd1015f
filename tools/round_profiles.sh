#!/bin/bash
# Everything profiles/rN_* is made from, in one GPU call: tools/round_profiles.sh <outdir under gpurun_out> [pmc]
#   bench.json            python bench.py --steps 20 --warmup 3 (the driver's form)
#   kernel_stats.csv      rocprofv3 --kernel-trace --stats of a short bench run (same command line as rounds 3-5)
#   demo_bench_*.json     tools/bench_demo.py at 1080p and VGA;  greedy_bench.json  tools/bench_greedy.py
#   pmc/ (+ <out>/pmc.json with the `pmc` argument)   tools/pmc_profile.sh
root="$(cd "$(dirname "$0")/.." && pwd)"
out=$1
mkdir -p "$root/$out"
cd "$root"
timeout -k 10 900 python bench.py --steps 20 --warmup 3 > "$out/bench.json" 2> "$out/bench.err" || { echo "bench failed"; tail -5 "$out/bench.err"; exit 1; }
echo "bench done"
(cd /tmp && export TMPDIR=/tmp && BALF_FP16_CHECK=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out/kstats" -- python3 "$root/bench.py" --steps 3 --warmup 1 --cpu-images 0 --other-configs 0 --other-steps 0 --no-single-rank-collective --sustained-seconds 0 --host-fed-steps 0 > "$root/$out/kstats_bench.json" 2> "$root/$out/kstats.err") || { echo "kernel stats failed"; exit 1; }
find "$out/kstats" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
rm -rf "$out/kstats"
echo "kernel stats done"
timeout -k 10 300 python tools/bench_demo.py 32 1080 1920 5 > "$out/demo_bench_1080p.json" 2> "$out/demo.err" && \
timeout -k 10 300 python tools/bench_demo.py 32 480 640 10 > "$out/demo_bench_vga.json" 2>> "$out/demo.err" && \
timeout -k 10 200 python tools/bench_greedy.py > "$out/greedy_bench.json" 2>> "$out/demo.err" || { echo "demo bench failed"; tail -5 "$out/demo.err"; exit 1; }
echo "demo done"
if [ "$2" = "pmc" ]; then
  timeout -k 10 900 bash tools/pmc_profile.sh "$out/pmc" "$out/pmc.json" > "$out/pmc.log" 2>&1 || { echo "pmc failed"; tail -5 "$out/pmc.log"; exit 1; }
  find "$out/pmc" -name "*.csv" -delete; find "$out/pmc" -name "*.db" -delete
  echo "pmc done"
fi
