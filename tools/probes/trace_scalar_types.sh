export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/sc_trace -o t -- python $GRAFT_REPO_ROOT/tools/bench_scalar_types.py 128 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocprof_summary.py $(ls gpurun_out/sc_trace/*results.db gpurun_out/sc_trace/*/*results.db 2>/dev/null | head -1) 2>/dev/null | cut -c1-260 | head -16
rm -rf gpurun_out/sc_trace
