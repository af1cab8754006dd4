#!/bin/bash
# Round-6 evidence, third session (after the per-cell tensor-grid tables): the driver's own command again, and the trace,
# timeline and counters of config 5.  Outputs under gpurun_out/r06_final_c/.
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_final_c
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 2400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log ) 2> $OUT/bench_default.time
echo "default rc $? $(tail -3 $OUT/bench_default.time | tr '\n' ' ')"
T=$OUT/trace_c5
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$T -o t -- python $GRAFT_REPO_ROOT/bench.py --config 5 --no-cpu-baseline --no-sub-records --no-traffic --cpu-allcores 0 --steps 4 --warmup 3 > /dev/null 2>&1 )
DB=$(ls $T/*results.db $T/*/*results.db 2>/dev/null | head -1)
python tools/rocprof_summary.py $DB | cut -c1-170 | head -16 > $OUT/kernel_trace_config5.txt
python tools/rocprof_timeline.py $DB 30 20 matrix_pairs | cut -c1-190 > $OUT/timeline_config5.txt
rm -rf $T
python tools/collect_pmc.py $OUT/pmc_c5 246 5 > /dev/null 2>&1
rm -f $OUT/pmc_c5/*.db $OUT/pmc_c5/*/*.db
ls -la $OUT | head
