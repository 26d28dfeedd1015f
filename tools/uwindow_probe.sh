#!/bin/bash
# Round 6: the bound of "u' through the Infinity Cache" (VERDICT r5 item 1a).  tools/build_variant.sh uw64 -DBALF_ABLATE_UWINDOW=64
# builds a library whose stage-1/2 grid kernels store u' into, and whose block kernels load it from, a 64 MiB window (wrong results,
# same instruction streams, same access pattern inside the window).  Per-kernel times of main vs the windows, then HBM bytes
# (FETCH_SIZE / WRITE_SIZE) and L2 hit counters of both: tools/uwindow_probe.sh <outdir under gpurun_out>
root="$(cd "$(dirname "$0")/.." && pwd)"
out=$1
mkdir -p "$root/$out"
cd "$root"
bash tools/kern16_variants.sh "$out" uw64 uw32
PMC_IMAGES=16 tools/pmc_run.sh "$out/pmc_main" "FETCH_SIZE" "WRITE_SIZE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"
BALF_HIP_LIB="$root/balf_amd/libbalf_hip_uw64.so" PMC_IMAGES=16 PMC_EXTRA_ARGS=--allow-diagnostic-build tools/pmc_run.sh "$out/pmc_uw64" "FETCH_SIZE" "WRITE_SIZE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"
