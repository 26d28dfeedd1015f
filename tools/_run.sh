mkdir -p gpurun_out/r2e
python tools/check_f16.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2e/check.txt
grep -q "equal: False" gpurun_out/r2e/check.txt && { echo "NONDETERMINISTIC - stop"; exit 1; }
python tools/bench_kernels.py fp16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2e/bk.txt
timeout 900 python -m pytest tests/test_forward_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r2e/pytest_fwd.txt
python tools/bench_kernels.py fp16 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2e/bk.txt
