#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r6s25
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -m gpu -k "tensor_grid or cluster or fullsize" > gpurun_out/r6s25/tests.txt 2>&1
tail -3 gpurun_out/r6s25/tests.txt
for mode in "MPCX_GRID_STAGE=1" "MPCX_GRID_STAGE=0" "MPCX_TENSOR_GRID=0"; do
  echo "== $mode"
  env $mode timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-sub-records --cpu-allcores 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['ms_per_step'], r['value'], r.get('roofline',{}).get('launch_ms'), r.get('roofline',{}).get('kernel'))
"
done 2>&1 | tee gpurun_out/r6s25/bench.txt
