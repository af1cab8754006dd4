#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r6s32
timeout 1500 python -m pytest tests/test_gpu_driver.py -q -x -m gpu > gpurun_out/r6s32/tests.txt 2>&1
tail -3 gpurun_out/r6s32/tests.txt
timeout 900 python tools/driver_config2.py 256 10 > gpurun_out/r6s32/driver_config2.json 2> gpurun_out/r6s32/driver_config2.log
tail -2 gpurun_out/r6s32/driver_config2.log; cat gpurun_out/r6s32/driver_config2.json | cut -c1-400
MPCX_DRIVER_NO_GRID=1 timeout 900 python tools/driver_config2.py 256 10 2>&1 | tail -2 | cut -c1-300
echo "== shuffled sub-record"
timeout 900 python bench.py --numbering shuffled --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-sub-records --cpu-allcores 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['ms_per_step'], r['value'], [ (k['kernel'],round(k['launch_ms'],3)) for k in r.get('roofline_kernels',[])])
"
