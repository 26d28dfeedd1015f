python tools/check_f16.py 2>&1 | grep -v amdgpu.ids
python tools/bench_kernels.py fp16 2>&1 | grep -v amdgpu.ids
timeout 1200 python -m pytest tests/test_forward_gpu.py tests/test_nms_gpu.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -4
python tools/bench_kernels.py fp16 2>&1 | grep -v amdgpu.ids
