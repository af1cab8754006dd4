#!/bin/bash
# config 3's momentum right-hand side (vector_ownblock_kernel, P2 vector space): rows per block x threads
for rows in 256 512 1024 2048; do for th in 256 512; do
  MPCX_VECTOR_BLOCK_ROWS=$((rows*3)) MPCX_VECTOR_THREADS=$th python bench.py --config 3 --steps 10 --warmup 2 --no-traffic --no-cpu-baseline --no-sub-records 2>/dev/null \
   | python -c "import json,sys;d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]);print('rows/3 $rows threads $th', [(k['kernel'],round(k['launch_ms'],3)) for k in d['roofline_kernels'] if 'vector' in k['kernel']], round(d['ms_per_step'],3))"
done; done
