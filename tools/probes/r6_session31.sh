#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r6s31
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cluster_plan.py tests/test_gpu_general_numbering.py tests/test_gpu_ufcx_clusters.py tests/test_gpu_driver.py -q -x -m gpu > gpurun_out/r6s31/tests.txt 2>&1
tail -12 gpurun_out/r6s31/tests.txt | cut -c1-300
