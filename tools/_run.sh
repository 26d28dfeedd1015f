python tools/check_f16.py 2>&1 | grep -v amdgpu.ids | tail -2
python tools/bench_kernels.py fp16 2>&1 | grep -v amdgpu.ids | cut -c1-330
timeout 1200 python -m pytest tests/test_forward_gpu.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -2
python tools/soak.py 60 16 1088 1920 2>&1 | tail -1
python tools/bench_kernels.py fp16 2>&1 | grep -v amdgpu.ids | cut -c1-330
