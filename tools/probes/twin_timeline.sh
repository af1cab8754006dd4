# which launches of a step on the locality twin overlap (config 2, nodes and cells shuffled)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/twin_trace
cd /tmp && timeout 900 rocprofv3 --kernel-trace -d $OUT -o t -- python $GRAFT_REPO_ROOT/bench.py --numbering shuffled --no-cpu-baseline --no-sub-records --no-config-records --no-traffic --steps 6 --warmup 4 $EXTRA > $OUT.log 2>&1
tail -1 $OUT.log | cut -c1-300
cd $GRAFT_REPO_ROOT && python tools/rocprof_timeline.py $(ls $OUT/*results.db $OUT/*/*results.db 2>/dev/null | head -1) 100000 20 | cut -c1-160 > gpurun_out/twin_timeline.txt
python tools/rocprof_summary.py $(ls $OUT/*results.db $OUT/*/*results.db 2>/dev/null | head -1) | cut -c1-150 | head -12
rm -rf $OUT
