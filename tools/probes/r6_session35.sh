#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r6s35
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_p2_fullsize.py tests/test_gpu_general_numbering.py -q -x -m gpu > gpurun_out/r6s35/tests.txt 2>&1
tail -8 gpurun_out/r6s35/tests.txt | cut -c1-300
for g in 1 0; do
  echo "== config 5 MPCX_CELL_GRID=$g"
  MPCX_CELL_GRID=$g timeout 900 python bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-sub-records --cpu-allcores 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['ms_per_step'], r['value'], [(k['kernel'], round(k['launch_ms'],3)) for k in r.get('roofline_kernels',[])])
"
done 2>&1 | tee gpurun_out/r6s35/bench.txt
