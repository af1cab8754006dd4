#!/bin/bash
# round 6, session 14: CSR-valued node-block kernel, block size (LDS) x threads: two workgroups per CU with the staged masks?
cd /root/repo
mkdir -p gpurun_out/r6s14
for nnz in 9216 8192 6144 4608; do for th in 512 1024; do
  echo "== max_nnz=$nnz threads=$th"
  MPCX_ROWBLOCK_MAX_NNZ=$nnz MPCX_BLOCK_SCALAR=0 MPCX_NODEBLOCK_CSR_THREADS=$th timeout 600 python bench.py --config 3 --steps 5 --warmup 2 \
     --no-cpu-baseline --no-traffic --no-sub-records --cpu-allcores 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['ms_per_step'], r.get('roofline',{}).get('launch_ms'), r.get('roofline',{}).get('kernel'))
"
done; done 2>&1 | tee gpurun_out/r6s14/sweep.txt
