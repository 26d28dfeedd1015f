#!/bin/bash
# A/B of PMC counters between the main library and a variant (development aid): tools/pmc_ab.sh <outdir> <variant> "<counters>" ...
root="$(cd "$(dirname "$0")/.." && pwd)"
out=$1; var=$2; shift 2
"$root/tools/pmc_run.sh" "$out/main" "$@"
BALF_HIP_LIB="$root/balf_amd/libbalf_hip_$var.so" "$root/tools/pmc_run.sh" "$out/$var" "$@"
