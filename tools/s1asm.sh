#!/bin/bash
# compile detector_f16.hip to assembly and show the stage-1 kernels' figures + their vector-memory skeleton (development aid)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}
cd "$ROOT/balf_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wall -Wno-unused-function "$@" -S --cuda-device-only detector_f16.hip -o $OUT/d16.s 2>&1 | grep -E "error|warning: [^a]" -A5 | head -30
"$ROOT/tools/kstats.sh" $OUT/d16.s "stage1"
awk '/stage1_kernel16ILi0ELb0EEEvNS_9StageArgsE:/,/s_endpgm/' $OUT/d16.s > $OUT/g.s; awk '/stage1_kernel16ILi1ELb0EEEvNS_9StageArgsE:/,/s_endpgm/' $OUT/d16.s > $OUT/b.s
for f in g b; do echo "== $f"; grep -n "vmcnt\|global_load\|scratch_\|Loop Header\|s_endpgm" $OUT/$f.s | awk -F: '$1>450' | tr '\n' ';' | cut -c1-2200; echo; done
