#!/bin/bash
OUT=gpurun_out/c5_sweep; mkdir -p $OUT
run() { name=$1; shift
  env "$@" python bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $OUT/$name.json 2> $OUT/$name.log
  python tools/show_bench.py $OUT/$name.json | grep -vE "roofline|timings|generic" | tr '\n' ' '; echo " [$name]"; }
run nnz7680 MPCX_ROWBLOCK_MAX_NNZ=7680
run nnz6912 MPCX_ROWBLOCK_MAX_NNZ=6912
run nnz6144 MPCX_ROWBLOCK_MAX_NNZ=6144
run nnz4608_t256 MPCX_ROWBLOCK_MAX_NNZ=4608 MPCX_ROWBLOCK_THREADS=256
run nnz18432_t1024 MPCX_ROWBLOCK_MAX_NNZ=18432 MPCX_ROWBLOCK_MAX_ROWS=1024 MPCX_ROWBLOCK_THREADS=1024
