#!/bin/bash
# Round-6 evidence, second session (after the tensor-grid right-hand side and the CSR-valued write-out): the driver's own
# command, its kernel trace, the timeline of the config-2 step, the imported-text and hexahedron variants, PMC of configs 2, 3.
# Outputs under gpurun_out/r06_final_b/; summaries copied to profiles/ afterwards.
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_final_b
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 2400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log ) 2> $OUT/bench_default.time
echo "default rc $? $(tail -3 $OUT/bench_default.time | tr '\n' ' ')"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --no-traffic --no-cpu-baseline --no-sub-records > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json 2> /dev/null )
DB=$(ls $OUT/trace/*results.db $OUT/trace/*/*results.db 2>/dev/null | head -1)
python tools/rocprof_summary.py $DB > $OUT/kernel_trace_config2.txt
python tools/rocprof_timeline.py $DB 40 2 vector_cube_grid | cut -c1-190 > $OUT/timeline_config2.txt
rm -rf $OUT/trace
T=$OUT/trace_c3
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$T -o t -- python $GRAFT_REPO_ROOT/bench.py --config 3 --no-cpu-baseline --no-sub-records --no-traffic --cpu-allcores 0 --steps 4 --warmup 3 > /dev/null 2>&1 )
DB=$(ls $T/*results.db $T/*/*results.db 2>/dev/null | head -1)
python tools/rocprof_summary.py $DB | cut -c1-170 | head -16 > $OUT/kernel_trace_config3.txt
rm -rf $T
timeout 900 python bench.py --ufcx generated --no-cpu-baseline > $OUT/bench_config2_ufcx.json 2> $OUT/bench_config2_ufcx.log
echo "ufcx rc $?"
timeout 900 python bench.py --cell hex --no-cpu-baseline > $OUT/bench_config2_hex.json 2> $OUT/bench_config2_hex.log
echo "hex rc $?"
timeout 600 tools/probes/store_pattern.bin > $OUT/store_pattern.txt 2>&1
python tools/collect_pmc.py $OUT/pmc_c2 256 2 > /dev/null 2>&1
rm -f $OUT/pmc_c2/*.db $OUT/pmc_c2/*/*.db
python tools/collect_pmc.py $OUT/pmc_c3 128 3 > /dev/null 2>&1
rm -f $OUT/pmc_c3/*.db $OUT/pmc_c3/*/*.db
ls -la $OUT | head -40
