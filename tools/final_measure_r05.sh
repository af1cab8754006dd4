#!/bin/bash
# Round-5 evidence on the GPU box: the driver's own command (python bench.py: config 2 with the config 3 / 4 / 5 sub-records,
# the imported-text / generic / shuffled sub-records, in-run PMC, CPU baselines), the rocprofv3 kernel trace of the same
# command, config 2 with imported kernels and on hexahedra as full records, PMC counter sets of the config-2 kernels (built-in
# and imported).  Outputs under gpurun_out/r05_final/; the summaries to be judged are copied to profiles/ afterwards.
set -u
OUT=gpurun_out/r05_final
mkdir -p $OUT
( time timeout 1700 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log ) 2> $OUT/bench_default.time
echo "default rc $? $(tail -3 $OUT/bench_default.time | tr '\n' ' ')"
export TMPDIR=/tmp
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --no-traffic --no-cpu-baseline --no-sub-records > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json 2> /dev/null )
python tools/rocprof_summary.py $(ls $OUT/trace/*results.db $OUT/trace/*/*results.db 2>/dev/null | head -1) > $OUT/kernel_trace_config2.txt
rm -rf $OUT/trace
timeout 900 python bench.py --ufcx generated --no-cpu-baseline > $OUT/bench_config2_ufcx.json 2> $OUT/bench_config2_ufcx.log
echo "ufcx rc $?"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --ufcx generated --no-traffic --no-cpu-baseline --no-sub-records > /dev/null 2>&1 )
python tools/rocprof_summary.py $(ls $OUT/trace/*results.db $OUT/trace/*/*results.db 2>/dev/null | head -1) > $OUT/kernel_trace_config2_ufcx.txt
rm -rf $OUT/trace
timeout 900 python bench.py --cell hex --no-cpu-baseline > $OUT/bench_config2_hex.json 2> $OUT/bench_config2_hex.log
echo "hex rc $?"
python tools/collect_pmc.py $OUT/pmc_c2 256 2 > /dev/null 2>&1
rm -f $OUT/pmc_c2/*.db $OUT/pmc_c2/*/*.db
python tools/collect_pmc.py $OUT/pmc_c2_ufcx 256 2 --ufcx generated > /dev/null 2>&1
rm -f $OUT/pmc_c2_ufcx/*.db $OUT/pmc_c2_ufcx/*/*.db
ls -la $OUT $OUT/pmc_c2 | head -40
