#!/bin/bash
# The HOST side of libbalf_hip.so under AddressSanitizer + UndefinedBehaviorSanitizer:
# weight packer, workspace planners, argument checks, state-tensor table, status plumbing.  Device code is compiled as usual
# (-fno-gpu-sanitize: GPU sanitizers are not available on this pool and this library never runs a kernel); the result is a
# separately named library, balf_amd/libbalf_hip_asan.so, that only the CPU job of tests/test_asan_host.py loads (with the
# sanitizer runtime preloaded).  This script and that test stay in the build container (.gpurunignore).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Wall -Wno-unused-function -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -fno-sanitize-recover=undefined"
mkdir -p obj_asan
pids=()
for f in *.hip; do
  $HIPCC $FLAGS -c "$f" -o "obj_asan/${f%.hip}.o" &
  pids+=($!)
done
fail=0
for p in "${pids[@]}"; do wait "$p" || fail=1; done
[ $fail = 0 ] || { echo "build_asan.sh: a translation unit failed to compile" >&2; exit 1; }
$HIPCC --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan obj_asan/*.o -o ../libbalf_hip_asan.so
echo "built $(cd .. && pwd)/libbalf_hip_asan.so (host code instrumented; preload $($HIPCC -print-file-name=libclang_rt.asan-x86_64.so))"
