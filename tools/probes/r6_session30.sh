#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r6s30
timeout 1500 python -m pytest tests/test_hex.py -q -x -m gpu > gpurun_out/r6s30/tests.txt 2>&1
tail -3 gpurun_out/r6s30/tests.txt
for g in 1 0; do
  echo "== hex MPCX_BOX_GRID=$g"
  MPCX_BOX_GRID=$g timeout 600 python bench.py --cell hex --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-sub-records --cpu-allcores 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['ms_per_step'], r['value'], r.get('roofline',{}).get('launch_ms'), r.get('roofline',{}).get('kernel'))
"
done 2>&1 | tee gpurun_out/r6s30/bench.txt
