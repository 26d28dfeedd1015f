set -x
mkdir -p gpurun_out/r2m
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r2m/pytest_gpu.txt
python bench.py > gpurun_out/r2m/bench.json 2> gpurun_out/r2m/bench.err
python bench.py --other-configs 0 --cpu-images 0 > gpurun_out/r2m/bench2.json 2>> gpurun_out/r2m/bench.err
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2m/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-images 0 --other-configs 0 --no-single-rank-collective > $GRAFT_REPO_ROOT/gpurun_out/r2m/stats_bench.json 2>/dev/null
cd $GRAFT_REPO_ROOT
tools/pmc_profile.sh gpurun_out/r2m/pmc profiles/r2_pmc.json > /dev/null 2>&1
python tools/soak.py 100 16 1088 1920 2>&1 | tail -1
python tools/soak.py 200 4 512 640 2>&1 | tail -1
python - <<'PY'
import json
for f in ('gpurun_out/r2m/bench.json','gpurun_out/r2m/bench2.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['index_match'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['bound'], d['roofline']['frac'])
    print({k:round(v,2) for k,v in d['kernels_ms_per_step'].items()})
PY
