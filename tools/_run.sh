set -x
mkdir -p gpurun_out/r2f
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/r2f/pytest_gpu.txt
python bench.py > gpurun_out/r2f/bench.json 2> gpurun_out/r2f/bench.err; tail -3 gpurun_out/r2f/bench.err; python -c "
import json;d=json.load(open('gpurun_out/r2f/bench.json'));print({k:d[k] for k in ('value','ms_per_step','index_match','other_configs','config')});print(d['roofline']);print(d.get('forward_hbm'))"
