#!/bin/bash
# round 6, session 12: CSR-valued node-block kernel taken apart: no write-out (2), no entity loop (3), both (0)
cd /root/repo
mkdir -p gpurun_out/r6s12
for mode in 0 2 3; do for th in 1024; do
  echo "== mode=$mode threads=$th"
  MPCX_NODEBLOCK_NARROW_STORES=$mode MPCX_BLOCK_SCALAR=0 MPCX_NODEBLOCK_CSR_THREADS=$th timeout 600 python bench.py --config 3 --steps 5 --warmup 2 \
     --no-cpu-baseline --no-traffic --no-sub-records --cpu-allcores 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['ms_per_step'], r.get('roofline',{}).get('launch_ms'), r.get('roofline',{}).get('kernel'))
"
done; done 2>&1 | tee gpurun_out/r6s12/sweep.txt
