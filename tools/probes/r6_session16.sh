#!/bin/bash
# round 6, session 16: config 2's right-hand side on the tensor grid of a box (19 evaluations per factor instead of 84 points)
cd /root/repo
mkdir -p gpurun_out/r6s16
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_general_numbering.py tests/test_gpu_cluster_plan.py \
   tests/test_gpu_independent.py tests/test_gpu_reference_style.py tests/test_gpu_ufcx_clusters.py tests/test_gpu_driver.py -q -x -m gpu > gpurun_out/r6s16/tests.txt 2>&1
tail -5 gpurun_out/r6s16/tests.txt
for g in 0 1; do
  echo "== MPCX_BOX_GRID=$g"
  MPCX_BOX_GRID=$g timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-sub-records --cpu-allcores 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['ms_per_step'], r['value'], r.get('roofline',{}).get('launch_ms'), r.get('roofline',{}).get('kernel'))
"
done 2>&1 | tee gpurun_out/r6s16/bench.txt
