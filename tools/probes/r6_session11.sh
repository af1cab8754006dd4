#!/bin/bash
# round 6, session 11: lean write-out of the CSR-valued node-block kernel (scalar row bounds, one LDS read per row and lane,
# masks from LDS) against the per-entry-branch version of session 9
cd /root/repo
mkdir -p gpurun_out/r6s11
timeout 1200 python -m pytest tests/test_stokes.py tests/test_gpu_corun.py tests/test_gpu_parity.py -q -x -m gpu > gpurun_out/r6s11/tests.txt 2>&1
tail -3 gpurun_out/r6s11/tests.txt
for th in 512 768 1024; do
  echo "== threads=$th"
  MPCX_BLOCK_SCALAR=0 MPCX_NODEBLOCK_CSR_THREADS=$th timeout 600 python bench.py --config 3 --steps 5 --warmup 2 \
     --no-cpu-baseline --no-traffic --no-sub-records --cpu-allcores 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['ms_per_step'], r.get('roofline',{}).get('launch_ms'), r.get('roofline',{}).get('kernel'))
"
done 2>&1 | tee gpurun_out/r6s11/sweep.txt
