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
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ common.h -nt "$o" ] || [ ../../include/balf_hip.h -nt "$o" ] || \
     { [ -f "${f%.hip}.h" ] && [ "${f%.hip}.h" -nt "$o" ]; } || { [ -f layout.h ] && [ layout.h -nt "$o" ]; } || [ prof.h -nt "$o" ] || [ det_common.h -nt "$o" ] || [ split16.h -nt "$o" ]; then
    $HIPCC $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC obj/*.o -o ../libbalf_hip.so
echo "built $(cd .. && pwd)/libbalf_hip.so"
