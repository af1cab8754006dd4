#!/bin/bash
# round 6, session 9: 16-byte stores in the CSR-valued node-block write-out, against the 8-byte loop, at 512 and 1024 threads
cd /root/repo
mkdir -p gpurun_out/r6s9
timeout 900 python -m pytest tests/test_stokes.py -q -x -m gpu > gpurun_out/r6s9/stokes.txt 2>&1
tail -3 gpurun_out/r6s9/stokes.txt
for narrow in 0 1; do for th in 512 768 1024; do
  echo "== narrow=$narrow threads=$th"
  MPCX_NODEBLOCK_NARROW_STORES=$narrow MPCX_BLOCK_SCALAR=0 MPCX_NODEBLOCK_CSR_THREADS=$th timeout 600 python bench.py --config 3 --steps 5 --warmup 2 \
     --no-cpu-baseline --no-traffic --no-sub-records --cpu-allcores 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['ms_per_step'], r.get('roofline',{}).get('kernel_ms'), r.get('roofline',{}).get('kernel'))
"
done; done 2>&1 | tee gpurun_out/r6s9/sweep.txt
