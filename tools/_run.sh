timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python tools/soak.py 30 16 1088 1920 2>&1 | tail -1
python tools/soak.py 60 3 704 1216 2>&1 | tail -1
