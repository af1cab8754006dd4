#!/bin/bash
# Round-end evidence on the GPU box (run through gpurun): GPU tests, bench line, rocprofv3 kernel
# trace of the same bench command, PMC passes (tools/collect_pmc.py).  Outputs under gpurun_out/;
# copy the summaries into profiles/.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "$1" != "trace-only" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/final_tests.txt
  timeout 400 python bench.py 2>gpurun_out/final_bench.log | tail -1 > gpurun_out/final_bench.json
fi
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/final_trace
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/final_trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/final_trace.log 2>&1
DB=$(find /tmp/final_trace -name '*.db' | head -1)
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $DB > $GRAFT_REPO_ROOT/gpurun_out/final_kernel_trace.txt 2>&1
if [ "$1" != "trace-only" ]; then
  timeout 1500 python $GRAFT_REPO_ROOT/tools/collect_pmc.py $GRAFT_REPO_ROOT/gpurun_out/pmc_final 256 > $GRAFT_REPO_ROOT/gpurun_out/final_pmc.log 2>&1
  rm -f $GRAFT_REPO_ROOT/gpurun_out/pmc_final/*.db
fi
