#!/bin/bash
OUT=gpurun_out/c3_sweep; mkdir -p $OUT
run() { name=$1; shift
  env "$@" python bench.py --config 3 --steps 10 --warmup 2 --no-cpu-baseline --no-traffic > $OUT/$name.json 2> $OUT/$name.log
  python tools/show_bench.py $OUT/$name.json | grep -vE "roofline|one_shot|timings" | tr '\n' ' '; echo " [$name]"; }
run base
run t512 MPCX_ROWBLOCK_THREADS=512
run t768 MPCX_ROWBLOCK_THREADS=768
run t256 MPCX_ROWBLOCK_THREADS=256
