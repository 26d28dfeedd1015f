python tools/check_f16.py 2>&1 | grep -v amdgpu.ids | tee /tmp/chk.txt
grep -q "equal: False\|bad px: [1-9]" /tmp/chk.txt && { echo "WRONG - stop"; exit 1; }
tools/run_variants.sh gpurun_out/r2n fp16 ring | cut -c1-330
timeout 1200 python -m pytest tests/test_forward_gpu.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -3
python tools/soak.py 60 16 1088 1920 2>&1 | tail -1
