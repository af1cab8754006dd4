#!/bin/bash
OUT=gpurun_out/c3_sweep; mkdir -p $OUT
run() { name=$1; shift
  env "$@" python bench.py --config 3 --steps 10 --warmup 2 --no-cpu-baseline --no-traffic > $OUT/$name.json 2> $OUT/$name.log
  python tools/show_bench.py $OUT/$name.json | grep -E "ms/step|b0" | tr '\n' ' '; echo " [$name]"; }
run base
run own4096 MPCX_VECTOR_OWNER_ROWS=4096
run own2048 MPCX_VECTOR_OWNER_ROWS=2048
run own3072 MPCX_VECTOR_OWNER_ROWS=3072
run noowner MPCX_VECTOR_OWNER=0
run own4096_t256 MPCX_VECTOR_OWNER_ROWS=4096 MPCX_VECTOR_THREADS=256
