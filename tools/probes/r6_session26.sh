#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r6s26
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r6s26/prof -o t -- python /root/repo/tools/probes/vector_only.py 256 10 > /root/repo/gpurun_out/r6s26/run.txt 2>&1
cd /root/repo
tail -2 gpurun_out/r6s26/run.txt
python tools/rocprof_timeline.py gpurun_out/r6s26/prof/t_results.db 14 2 > gpurun_out/r6s26/timeline.txt 2>&1
tail -14 gpurun_out/r6s26/timeline.txt | cut -c1-140
