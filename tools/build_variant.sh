#!/bin/bash
# Build a tuning variant of the library: tools/build_variant.sh <name> <extra -D flags...>
set -euo pipefail
name=$1; shift
cd "$(dirname "$0")/../balf_amd/csrc"
mkdir -p obj_$name
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DBALF_ALLOW_DIAGNOSTIC_BUILD=1 "$@" -c "$f" -o obj_$name/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC obj_$name/*.o -o ../libbalf_hip_$name.so
rm -rf obj_$name
echo built libbalf_hip_$name.so
