#!/bin/bash
# co-residency of the (lean, memory-bound) cluster matrix kernel and the (VALU-bound) cluster vector kernel on two streams:
# LDS floors shape how many workgroups of each kernel a CU takes
OUT=gpurun_out/c2_overlap; mkdir -p $OUT
run() { name=$1; shift
  env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-traffic > $OUT/$name.json 2> $OUT/$name.log
  python tools/show_bench.py $OUT/$name.json | grep -vE "roofline|one_shot|generic" | tr '\n' ' '; echo " [$name]"; }
run base
run mfloor90k MPCX_CUBE_LDS_FLOOR=90000
run mfloor110k MPCX_CUBE_LDS_FLOOR=110000
run mfloor90k_v40k MPCX_CUBE_LDS_FLOOR=90000 MPCX_VCUBE_LDS_FLOOR=35000
run mfloor90k_t256 MPCX_CUBE_LDS_FLOOR=90000 MPCX_CUBE_AFFINE_THREADS=256
run mfloor90k_t1024 MPCX_CUBE_LDS_FLOOR=90000 MPCX_CUBE_AFFINE_THREADS=1024
run nostreams MPCX_ASYNC_STREAMS=0
