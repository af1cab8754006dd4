#!/bin/bash
OUT=gpurun_out/el_sweep; mkdir -p $OUT
run() { name=$1; shift
  env "$@" python bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline --no-traffic > $OUT/$name.json 2> $OUT/$name.log
  python tools/show_bench.py $OUT/$name.json | grep -E "ms/step|elasticity|rowpair" | tr '\n' ' '; echo " [$name]"; }
run t512 MPCX_CUBE_EL_THREADS=512
run t256 MPCX_CUBE_EL_THREADS=256
run t768 MPCX_CUBE_EL_THREADS=768
run nnz6144_t512 MPCX_CUBE_MAX_NNZ=6144 MPCX_CUBE_EL_THREADS=512
run nnz6144_t256 MPCX_CUBE_MAX_NNZ=6144 MPCX_CUBE_EL_THREADS=256
run nnz4608_t256 MPCX_CUBE_MAX_NNZ=4608 MPCX_CUBE_EL_THREADS=256
run nnz4608_t512 MPCX_CUBE_MAX_NNZ=4608 MPCX_CUBE_EL_THREADS=512
