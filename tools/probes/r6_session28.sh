#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r6s28
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "tensor_grid or cluster" > gpurun_out/r6s28/tests.txt 2>&1
tail -2 gpurun_out/r6s28/tests.txt
for mode in "MPCX_VCUBE_ROWS=2048" "MPCX_VCUBE_ROWS=1024" "MPCX_VCUBE_ROWS=512" "MPCX_VCUBE_ROWS=4096"; do
  echo "== $mode"
  env $mode timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-sub-records --cpu-allcores 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['ms_per_step'], r['value'], r.get('roofline',{}).get('launch_ms'), r.get('roofline',{}).get('kernel'))
"
  env $mode timeout 300 python tools/probes/vector_only.py 256 10 2>/dev/null | tail -1
done 2>&1 | tee gpurun_out/r6s28/bench.txt
