#!/bin/bash
# bench_kernels.py for the main library and each named variant: tools/run_variants.sh <outdir> <precision> [names...]
out=$1; prec=$2; shift 2
mkdir -p $out
python tools/bench_kernels.py $prec 2>&1 | grep -v amdgpu.ids | sed -e "s/^/main: /" | tee $out/variants.txt
for v in "$@"; do
  BALF_HIP_LIB=$PWD/balf_amd/libbalf_hip_$v.so python tools/bench_kernels.py $prec 2>&1 | grep -v amdgpu.ids | sed -e "s/^/$v: /" | tee -a $out/variants.txt
done
python tools/bench_kernels.py $prec 2>&1 | grep -v amdgpu.ids | sed -e "s/^/main: /" | tee -a $out/variants.txt
