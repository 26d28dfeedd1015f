timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python tools/soak.py 40 16 1088 1920 2>&1 | tail -1
python tools/soak.py 100 4 512 640 2>&1 | tail -1
python tools/soak.py 100 3 704 1216 2>&1 | tail -1
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_r2_v4.json; cat gpurun_out/bench_r2_v4.json | cut -c1-600
