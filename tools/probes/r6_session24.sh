#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r6s24
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r6s24/prof -o t -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-sub-records --cpu-allcores 0 > /root/repo/gpurun_out/r6s24/bench.txt 2>&1
cd /root/repo
python tools/rocprof_timeline.py gpurun_out/r6s24/prof/t_results.db 60 2 vector_cube_grid > gpurun_out/r6s24/timeline.txt 2>&1
tail -45 gpurun_out/r6s24/timeline.txt | cut -c1-150
