python tools/check_f16.py 2>&1 | grep -v amdgpu.ids | tail -2
timeout 1200 python -m pytest tests/test_forward_gpu.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -3
bash tools/run_variants.sh gpurun_out/csv2 fp16 nofuse 2>&1 | cut -c1-330
