#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r6s36
for rows in 8192 4096 2048; do for th in 1024 512; do
  echo "== owner rows $rows threads $th"
  MPCX_VECTOR_OWNER_ROWS=$rows MPCX_VECTOR_BLOCK_ROWS_P2=$rows MPCX_CELL_GRID_THREADS=$th timeout 900 python bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-sub-records --cpu-allcores 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['ms_per_step'], r['value'], [(k['kernel'], round(k['launch_ms'],3)) for k in r.get('roofline_kernels',[])])
"
done; done 2>&1 | tee gpurun_out/r6s36/bench.txt
