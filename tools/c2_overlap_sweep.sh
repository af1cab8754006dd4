#!/bin/bash
# do the matrix and the vector call of a config-2 step overlap on their two streams?  LDS shares, stream priorities
OUT=gpurun_out/c2_overlap; mkdir -p $OUT
run() { name=$1; shift
  env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-traffic > $OUT/$name.json 2> $OUT/$name.log
  python tools/show_bench.py $OUT/$name.json | grep -vE "roofline|one_shot|generic|kernel" | tr '\n' ' '; echo " [$name]"; }
R256="MPCX_CUBE_MAX_ROWS=256 MPCX_CUBE_MAX_NNZ=4608"
run r256 $R256
run r256_v45 $R256 MPCX_VCUBE_LDS_FLOOR=45000
run r256_v45_vhigh $R256 MPCX_VCUBE_LDS_FLOOR=45000 MPCX_VECTOR_STREAM_PRIORITY=-1
run r256_v45_mhigh $R256 MPCX_VCUBE_LDS_FLOOR=45000 MPCX_MATRIX_STREAM_PRIORITY=-1
run r256_v60 $R256 MPCX_VCUBE_LDS_FLOOR=60000
run r256_v60_vhigh $R256 MPCX_VCUBE_LDS_FLOOR=60000 MPCX_VECTOR_STREAM_PRIORITY=-1
run r256_mhigh $R256 MPCX_MATRIX_STREAM_PRIORITY=-1
run r128_v45_vhigh MPCX_CUBE_MAX_ROWS=128 MPCX_CUBE_MAX_NNZ=2304 MPCX_VCUBE_LDS_FLOOR=45000 MPCX_VECTOR_STREAM_PRIORITY=-1
