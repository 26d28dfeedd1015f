#!/bin/bash
# Build libbalf_hip.so for gfx950 with plain hipcc (no hipify, no torch extension machinery).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p obj
pids=()
for f in *.hip; do
  o=obj/${f%.hip}.o
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
$HIPCC --offload-arch=gfx950 -shared -fPIC obj/*.o -o ../libbalf_hip.so
# the hand-counted waits of stage1_f16.h are checked against the built code (tests/test_build_invariants.py); the compiler that
# produced it is recorded next to the library so that a toolchain change is visible
$HIPCC --version | head -2 > ../libbalf_hip.toolchain.txt
echo "built $(cd .. && pwd)/libbalf_hip.so"
