BALF_HIP_LIB=$PWD/balf_amd/libbalf_hip_dropwlo.so python tools/check_f16.py 2>&1 | grep -v amdgpu.ids | tail -2
python tools/stamps_cs.py 2>&1 | grep -v amdgpu.ids | grep "^C=" | cut -c1-600
bash tools/run_variants.sh gpurun_out/csv2 fp16 dropwlo nw2a nw2b 2>&1 | cut -c1-330
