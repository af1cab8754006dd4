#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r6s22
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -m gpu -k "tensor_grid or cluster or fullsize" > gpurun_out/r6s22/tests.txt 2>&1
tail -3 gpurun_out/r6s22/tests.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r6s22/prof -o t -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-sub-records --cpu-allcores 0 > /root/repo/gpurun_out/r6s22/bench.txt 2>&1
cd /root/repo
tail -1 gpurun_out/r6s22/bench.txt | cut -c1-300
f=$(ls gpurun_out/r6s22/prof/*/*kernel_stats.csv 2>/dev/null | head -1); head -14 "$f" | cut -c1-220 | tee gpurun_out/r6s22/kernel_stats_head.txt
