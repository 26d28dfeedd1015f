BALF_HIP_LIB=$PWD/balf_amd/libbalf_hip_keepx0.so python tools/check_f16.py 2>&1 | grep -v amdgpu.ids | tail -1
bash tools/run_variants.sh gpurun_out/csv2 fp16 keepx0 2>&1 | cut -c1-200
