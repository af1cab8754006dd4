timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python tools/bench_configs.py 96 128 2>&1 | tail -1
