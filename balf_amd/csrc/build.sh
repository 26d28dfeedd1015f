#!/bin/bash
# Build libbalf_hip.so for gfx950 with plain hipcc (no hipify, no torch extension machinery).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
OBJ=obj
OUT=../libbalf_hip.so
if [ "${BALF_ASAN:-0}" = "1" ]; then
  # SURVEY section 5 / VERDICT r4 item 6: the HOST side of the library (weight packer, planners, argument checks, status
  # plumbing) under AddressSanitizer + UndefinedBehaviorSanitizer.  Device code is compiled as usual (-fno-gpu-sanitize: GPU
  # sanitizers are not available on this pool); the result is a separately named library that only the CPU job of
  # tests/test_asan_host.py loads (LD_PRELOAD of the sanitizer runtime, no GPU call is made there).
  FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Wall -Wno-unused-function -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -fno-sanitize-recover=undefined"
  OBJ=obj_asan
  OUT=../libbalf_hip_asan.so
fi
mkdir -p $OBJ
pids=()
for f in *.hip; do
  o=$OBJ/${f%.hip}.o
  stale=0
  [ -f "$o" ] || stale=1
  [ "$f" -nt "$o" ] && stale=1
  for h in *.h ../../include/balf_hip.h; do [ "$h" -nt "$o" ] && stale=1; done   # every TU may include any header here
  if [ $stale = 1 ]; then
    $HIPCC $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
done
fail=0
for p in "${pids[@]:-}"; do [ -n "$p" ] && { wait "$p" || fail=1; }; done
[ $fail = 0 ] || { echo "build.sh: a translation unit failed to compile" >&2; exit 1; }
if [ "${BALF_ASAN:-0}" = "1" ]; then
  $HIPCC --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan $OBJ/*.o -o $OUT
  echo "built $(cd .. && pwd)/libbalf_hip_asan.so (host code instrumented; preload $($HIPCC -print-file-name=libclang_rt.asan-x86_64.so))"
  exit 0
fi
$HIPCC --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -o $OUT
# the hand-counted waits of stage1_f16.h are checked against the built code (tests/test_build_invariants.py); the compiler that
# produced it is recorded next to the library so that a toolchain change is visible
$HIPCC --version | head -2 > ../libbalf_hip.toolchain.txt
echo "built $(cd .. && pwd)/libbalf_hip.so"
