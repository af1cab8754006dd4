#!/bin/bash
OUT=gpurun_out/vt_sweep; mkdir -p $OUT
run() { c=$1; name=$2; shift; shift
  env "$@" python bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline --no-traffic > $OUT/$name.json 2> $OUT/$name.log
  python tools/show_bench.py $OUT/$name.json | grep -vE "roofline|one_shot|timings" | tr '\n' ' '; echo " [$name]"; }
run 3 c3_auto
run 3 c3_v512 MPCX_VECTOR_THREADS=512
run 3 c3_v1024 MPCX_VECTOR_THREADS=1024
run 5 c5_auto
run 5 c5_v512 MPCX_VECTOR_THREADS=512
run 5 c5_v1024 MPCX_VECTOR_THREADS=1024
run 5 c5_v768 MPCX_VECTOR_THREADS=768
