#!/bin/bash
OUT=gpurun_out/c5_sweep; mkdir -p $OUT
run() { name=$1; shift
  env "$@" python bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $OUT/$name.json 2> $OUT/$name.log
  python tools/show_bench.py $OUT/$name.json | grep -vE "roofline|timings|generic|one_shot" | tr '\n' ' '; echo " [$name]"; }
run t640 MPCX_ROWBLOCK_THREADS=640
run t576 MPCX_ROWBLOCK_THREADS=576
run t704 MPCX_ROWBLOCK_THREADS=704
run t768 MPCX_ROWBLOCK_THREADS=768
