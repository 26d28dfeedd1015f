set -x
mkdir -p gpurun_out/r2r
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r2r/pytest_gpu.txt
python bench.py > gpurun_out/r2r/bench.json 2> gpurun_out/r2r/bench.err
python tools/soak.py 100 16 1088 1920 2>&1 | tail -1
python tools/soak.py 200 4 512 640 2>&1 | tail -1
python tools/soak.py 200 2 128 192 2>&1 | tail -1
python tools/soak.py 100 3 704 1216 2>&1 | tail -1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2r/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['index_match'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['bound'], d['roofline']['frac'])
print({k:round(v,2) for k,v in d['kernels_ms_per_step'].items()})
print([ (c['workload'][:12], round(c['images_per_s'])) for c in d['other_configs']])
PY
