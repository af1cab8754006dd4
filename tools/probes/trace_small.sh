#!/bin/bash
# kernel trace of a short default bench: average duration of every kernel of the step
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_small
rm -rf $OUT; mkdir -p $OUT
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-traffic --no-cpu-baseline --no-sub-records --no-shuffled-record "$@" > $OUT/bench.json 2>/dev/null )
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(ls $OUT/trace/*results.db $OUT/trace/*/*results.db 2>/dev/null | head -1) | cut -c1-150 | head -24
rm -rf $OUT/trace
python -c "import json;d=json.load(open('$OUT/bench.json'));print('step', d['ms_per_step'])"
