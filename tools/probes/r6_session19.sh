#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r6s19
for rep in 1 2; do for pf in 0 1; do for th in 256 128; do
  echo "== prefetch=$pf threads=$th rep=$rep"
  MPCX_VCUBE_THREADS=$th MPCX_VCUBE_PREFETCH=$pf timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-sub-records --cpu-allcores 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['ms_per_step'], r['value'], r.get('roofline',{}).get('launch_ms'), r.get('roofline',{}).get('kernel'))
"
done; done; done 2>&1 | tee gpurun_out/r6s19/bench.txt
