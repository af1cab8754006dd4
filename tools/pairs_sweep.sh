#!/bin/bash
# matrix_pairs_kernel (round 4): block size / threads / context mode at config 5 (P2 Poisson 246^3), config 3, config 4
OUT=gpurun_out/pairs_sweep; mkdir -p $OUT
run() { cfg=$1; name=$2; shift 2
  env "$@" timeout 900 python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-sub-records > $OUT/c${cfg}_$name.json 2> $OUT/c${cfg}_$name.log
  python tools/show_bench.py $OUT/c${cfg}_$name.json | grep -vE "roofline|timings|generic|one_shot" | tr '\n' ' '; echo " [c$cfg $name]"; }
for cfg in "$@"; do
case $cfg in
5)
run 5 p4608 MPCX_FORCE_KERNEL=matrix=pairs
run 5 p4608_nodict MPCX_FORCE_KERNEL=matrix=pairs MPCX_PAIRS_DICT=0
run 5 p4608_nostage_nodict MPCX_FORCE_KERNEL=matrix=pairs MPCX_PAIRS_DICT=0 MPCX_PAIRS_STAGE=0
run 5 p3584 MPCX_FORCE_KERNEL=matrix=pairs MPCX_PAIRS_MAX_NNZ=3584
run 5 p3584_nodict MPCX_FORCE_KERNEL=matrix=pairs MPCX_PAIRS_MAX_NNZ=3584 MPCX_PAIRS_DICT=0
run 5 p2304 MPCX_FORCE_KERNEL=matrix=pairs MPCX_PAIRS_MAX_NNZ=2304
run 5 p2304_nodict MPCX_FORCE_KERNEL=matrix=pairs MPCX_PAIRS_MAX_NNZ=2304 MPCX_PAIRS_DICT=0
run 5 p6144 MPCX_FORCE_KERNEL=matrix=pairs MPCX_PAIRS_MAX_NNZ=6144
run 5 p9216 MPCX_FORCE_KERNEL=matrix=pairs MPCX_PAIRS_MAX_NNZ=9216
run 5 p4608_rc MPCX_FORCE_KERNEL=matrix=pairs MPCX_PAIRS_CONTEXT=recompute
;;
3)
run 3 base
run 3 p4608 MPCX_FORCE_KERNEL=matrix=pairs
run 3 p2304 MPCX_FORCE_KERNEL=matrix=pairs MPCX_PAIRS_MAX_NNZ=2304
run 3 p1152 MPCX_FORCE_KERNEL=matrix=pairs MPCX_PAIRS_MAX_NNZ=1152
run 3 p2304_nodict MPCX_FORCE_KERNEL=matrix=pairs MPCX_PAIRS_MAX_NNZ=2304 MPCX_PAIRS_DICT=0
;;
4)
run 4 base
run 4 p4608 MPCX_FORCE_KERNEL=matrix=pairs
run 4 p2304 MPCX_FORCE_KERNEL=matrix=pairs MPCX_PAIRS_MAX_NNZ=2304
run 4 p1152 MPCX_FORCE_KERNEL=matrix=pairs MPCX_PAIRS_MAX_NNZ=1152
run 4 p2304_nodict MPCX_FORCE_KERNEL=matrix=pairs MPCX_PAIRS_MAX_NNZ=2304 MPCX_PAIRS_DICT=0
;;
esac
done
