#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r6s17
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "tensor_grid or cluster_vector" > gpurun_out/r6s17/tests.txt 2>&1
tail -5 gpurun_out/r6s17/tests.txt
