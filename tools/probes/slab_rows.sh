#!/bin/bash
# a 1/8-size problem (what one rank holds at 8 GPUs): owner rows per block of the cluster vector kernel, rows of the matrix blocks
for n in 128; do for vr in 2048 1024 512; do for mr in 512 256; do
  MPCX_VCUBE_ROWS=$vr MPCX_CUBE_MAX_ROWS=$mr python bench.py --size $n --steps 30 --warmup 5 --no-traffic --no-cpu-baseline --no-sub-records --no-shuffled-record 2>/dev/null \
   | python -c "import json,sys;d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]);print('n $n vrows $vr mrows $mr', [(k['kernel'].split('_kernel')[0],round(k['launch_ms'],3)) for k in d['roofline_kernels']], 'step', round(d['ms_per_step'],3))"
done; done; done
