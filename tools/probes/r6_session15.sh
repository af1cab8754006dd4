#!/bin/bash
# round 6, session 15: mask loads issued before the entity loop and consumed after it; blocks of 8192 slots by default
cd /root/repo
mkdir -p gpurun_out/r6s15
timeout 1200 python -m pytest tests/test_stokes.py tests/test_gpu_corun.py tests/test_gpu_parity.py -q -x -m gpu > gpurun_out/r6s15/tests.txt 2>&1
tail -3 gpurun_out/r6s15/tests.txt
for nnz in 8192 9216 7168; do for th in 1024 512; do
  echo "== slots=$nnz threads=$th"
  MPCX_NODEBLOCK_CSR_MAX_SLOTS=$nnz MPCX_BLOCK_SCALAR=0 MPCX_NODEBLOCK_CSR_THREADS=$th timeout 600 python bench.py --config 3 --steps 5 --warmup 2 \
     --no-cpu-baseline --no-traffic --no-sub-records --cpu-allcores 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['ms_per_step'], r.get('roofline',{}).get('launch_ms'), r.get('roofline',{}).get('kernel'))
"
done; done 2>&1 | tee gpurun_out/r6s15/sweep.txt
echo "== block-scalar default"
timeout 600 python bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-sub-records --cpu-allcores 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['ms_per_step'], r.get('roofline',{}).get('launch_ms'), r.get('roofline',{}).get('kernel'))
" | tee -a gpurun_out/r6s15/sweep.txt
