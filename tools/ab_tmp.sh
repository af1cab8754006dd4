timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for v in "" ; do
  echo "== $v"; env $v timeout 300 python bench.py --steps 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['timings_ms'])"
done
