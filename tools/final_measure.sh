#!/bin/bash
# Round evidence on the GPU box: driver-style bench lines for every config (+ the hexahedron variant of config 2), the
# rocprofv3 kernel trace of the default bench command, PMC counters of every config's kernels.
# Outputs under gpurun_out/r04_final/; the summaries to be judged are copied to profiles/ afterwards.
set -u
OUT=gpurun_out/r04_final
mkdir -p $OUT
for c in 2 3 4 5; do
  timeout 1200 python bench.py --config $c --steps 20 --warmup 3 > $OUT/bench_c$c.json 2> $OUT/bench_c$c.log
  echo "config $c rc $?"
done
timeout 1200 python bench.py --cell hex --steps 20 --warmup 3 > $OUT/bench_hex.json 2> $OUT/bench_hex.log
echo "hex rc $?"
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-traffic --no-cpu-baseline --no-sub-records > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json 2> /dev/null )
python tools/rocprof_summary.py $(ls $OUT/trace/*results.db $OUT/trace/*/*results.db 2>/dev/null | head -1) > $OUT/kernel_trace.txt
rm -rf $OUT/trace
for c in 2 3 4 5; do
  case $c in 2) n=256;; 3) n=128;; 4) n=56;; 5) n=246;; esac
  python tools/collect_pmc.py $OUT/pmc_c$c $n $c > /dev/null 2>&1
  rm -f $OUT/pmc_c$c/*.db $OUT/pmc_c$c/*/*.db
done
python tools/collect_pmc.py $OUT/pmc_hex 256 2 --cell hex > /dev/null 2>&1
rm -f $OUT/pmc_hex/*.db $OUT/pmc_hex/*/*.db
ls -la $OUT
