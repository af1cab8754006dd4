#!/bin/bash
OUT=gpurun_out/c2_sweep; mkdir -p $OUT
run() { name=$1; shift
  env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic > $OUT/$name.json 2> $OUT/$name.log
  python tools/show_bench.py $OUT/$name.json | grep -vE "roofline|one_shot|timings|generic" | tr '\n' ' '; echo " [$name]"; }
run base
run t768 MPCX_CUBE_AFFINE_THREADS=768
run t1024 MPCX_CUBE_AFFINE_THREADS=1024
run t640 MPCX_CUBE_AFFINE_THREADS=640
run r256_t512 MPCX_CUBE_MAX_ROWS=256 MPCX_CUBE_MAX_NNZ=4608
run r256_t384 MPCX_CUBE_MAX_ROWS=256 MPCX_CUBE_MAX_NNZ=4608 MPCX_CUBE_AFFINE_THREADS=384
run r1024_t1024 MPCX_CUBE_MAX_ROWS=1024 MPCX_CUBE_MAX_NNZ=18432 MPCX_CUBE_AFFINE_THREADS=1024
