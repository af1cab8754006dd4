#!/bin/bash
# Round-6 evidence on the GPU box, one session: the driver's own command (python bench.py: config 2 with the config 3 / 4 / 5
# sub-records -- each with its imported-text, CSR-valued and graph-replay figures --, the imported-text / generic / shuffled
# sub-records, in-run PMC, CPU baselines), the rocprofv3 kernel trace of the same command, kernel TIMELINES of the config-5 and
# config-3 steps (do the matrix and the vector kernel overlap?), package power / clock while they run alone and together, the
# co-run sweeps, PMC counter sets of configs 2 / 3 / 5.  Outputs under gpurun_out/r06_final/; the summaries to be judged are
# copied to profiles/ afterwards.
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_final
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 2400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log ) 2> $OUT/bench_default.time
echo "default rc $? $(tail -3 $OUT/bench_default.time | tr '\n' ' ')"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --no-traffic --no-cpu-baseline --no-sub-records > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json 2> /dev/null )
python tools/rocprof_summary.py $(ls $OUT/trace/*results.db $OUT/trace/*/*results.db 2>/dev/null | head -1) > $OUT/kernel_trace_config2.txt
rm -rf $OUT/trace
# timelines: the steps of configs 5 and 3 as the two library streams run them
for C in 5 3; do
  T=$OUT/trace_c$C
  ( cd /tmp && MPCX_CORUN=0 timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$T -o t -- python $GRAFT_REPO_ROOT/bench.py --config $C --no-cpu-baseline --no-sub-records --no-traffic --cpu-allcores 0 --steps 4 --warmup 3 > /dev/null 2>&1 )
  DB=$(ls $T/*results.db $T/*/*results.db 2>/dev/null | head -1)
  python tools/rocprof_summary.py $DB | cut -c1-170 | head -16 > $OUT/kernel_trace_config$C.txt
  python tools/rocprof_timeline.py $DB 44 20 matrix_pairs | cut -c1-190 > $OUT/timeline_config$C.txt
  rm -rf $T
done
timeout 900 python tools/probes/power_probe.py --config 5 > $OUT/power_config5.txt 2>&1
for C in 5 3; do timeout 900 python tools/probes/corun_probe.py --config $C --arms off,w2_f60,w2_f100,w3_f60,w3_f100,w1_f100 > $OUT/corun_config$C.txt 2>&1; done
timeout 900 python tools/probes/corun_probe.py --config 2 --generic --arms off,w2_f60,w2_f100,w3_f60,w1_f100 > $OUT/corun_config2_generic.txt 2>&1
timeout 900 python bench.py --ufcx generated --no-cpu-baseline > $OUT/bench_config2_ufcx.json 2> $OUT/bench_config2_ufcx.log
echo "ufcx rc $?"
timeout 900 python bench.py --cell hex --no-cpu-baseline > $OUT/bench_config2_hex.json 2> $OUT/bench_config2_hex.log
echo "hex rc $?"
python tools/collect_pmc.py $OUT/pmc_c2 256 2 > /dev/null 2>&1
rm -f $OUT/pmc_c2/*.db $OUT/pmc_c2/*/*.db
python tools/collect_pmc.py $OUT/pmc_c3 128 3 > /dev/null 2>&1
rm -f $OUT/pmc_c3/*.db $OUT/pmc_c3/*/*.db
python tools/collect_pmc.py $OUT/pmc_c5 246 5 > /dev/null 2>&1
rm -f $OUT/pmc_c5/*.db $OUT/pmc_c5/*/*.db
ls -la $OUT | head -50
