#!/bin/bash
# compile detector_f16.hip to assembly and show the stage-1 kernels' figures + their vector-memory skeleton (development aid)
cd /root/repo/balf_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wall -Wno-unused-function "$@" -S --cuda-device-only detector_f16.hip -o /tmp/d16.s 2>&1 | grep -E "error|warning: [^a]" -A5 | head -30
/root/repo/tools/kstats.sh /tmp/d16.s "stage1"
awk '/stage1_kernel16ILi0ELb0EEEvNS_9StageArgsE:/,/s_endpgm/' /tmp/d16.s > /tmp/g.s; awk '/stage1_kernel16ILi1ELb0EEEvNS_9StageArgsE:/,/s_endpgm/' /tmp/d16.s > /tmp/b.s
for f in g b; do echo "== $f"; grep -n "vmcnt\|global_load\|scratch_\|Loop Header\|s_endpgm" /tmp/$f.s | awk -F: '$1>450' | tr '\n' ';' | cut -c1-2200; echo; done
