#!/bin/bash
# block-shape sweep of the hexahedron matrix kernel (rows / nnz per row block, threads per workgroup)
OUT=gpurun_out/hex_sweep; mkdir -p $OUT
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --cell hex --steps 10 --warmup 2 --no-cpu-baseline --no-traffic > $OUT/$name.json 2> $OUT/$name.log
  python tools/show_bench.py $OUT/$name.json | grep -E "ms/step|hex_kernel" | tr '\n' ' '; echo " [$name]"
}
export MPCX_HEX_ONLY_AFFINE=1
run ao_base
run ao_r256_t512 MPCX_HEX_THREADS=512
run ao_r256_t384 MPCX_HEX_THREADS=384
run ao_r128_t128 MPCX_HEX_MAX_ROWS=128 MPCX_HEX_MAX_NNZ=4608 MPCX_HEX_THREADS=128
run ao_r128_t256 MPCX_HEX_MAX_ROWS=128 MPCX_HEX_MAX_NNZ=4608 MPCX_HEX_THREADS=256
run ao_r512_t512 MPCX_HEX_MAX_ROWS=512 MPCX_HEX_MAX_NNZ=14336 MPCX_HEX_THREADS=512
run ao_r64_t128 MPCX_HEX_MAX_ROWS=64 MPCX_HEX_MAX_NNZ=2304 MPCX_HEX_THREADS=128
