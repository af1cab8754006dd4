#!/bin/bash
# round 6: the whole GPU suite the way the driver runs it (serial), with durations
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6s6; mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 2700 python -m pytest tests -q -m gpu --durations=25 > $OUT/gpu_suite.txt 2>&1 ) 2> $OUT/gpu_suite.time
tail -40 $OUT/gpu_suite.txt | cut -c1-200; cat $OUT/gpu_suite.time
