#!/bin/bash
# tools/kern16.py for the main library and each named variant (development aid): tools/kern16_variants.sh <outdir> [names...]
out=$1; shift
mkdir -p $out
export BALF_FP16_CHECK=0
timeout -k 10 200 python tools/kern16.py 2>&1 | grep total | sed -e "s/^/main: /" | tee $out/variants.txt
for v in "$@"; do
  BALF_HIP_LIB=$PWD/balf_amd/libbalf_hip_$v.so timeout -k 10 200 python tools/kern16.py 2>&1 | grep total | sed -e "s/^/$v: /" | tee -a $out/variants.txt
done
