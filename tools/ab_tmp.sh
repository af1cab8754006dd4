timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for w in host device; do MPCX_PATTERN=$w timeout 300 python bench.py --setup-only 2>&1 | grep "pattern"; done
