#!/bin/bash
# round 6, session 10: streaming block expand (16-byte stores, 4 rows per wave) against the row-at-a-time kernel
cd /root/repo
mkdir -p gpurun_out/r6s10
{
MPCX_EXPAND_WIDE=0 timeout 600 python tools/probes/expand_probe.py 128 2>&1 | tail -2
MPCX_EXPAND_WIDE=1 MPCX_EXPAND_CHECK=1 timeout 600 python tools/probes/expand_probe.py 128 2>&1 | tail -3
} | tee gpurun_out/r6s10/expand.txt
