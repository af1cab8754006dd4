#!/bin/bash
# round 6: the driver's command once more on the final tree (+ config 4 alone, CSR-valued thread sweep)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_final; mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 2400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log ) 2> $OUT/bench_default.time
echo "default rc $? $(tail -3 $OUT/bench_default.time | tr '\n' ' ')"
python tools/design_table.py $OUT/bench_default.json | cut -c1-400
for T in 640 896; do
  MPCX_BLOCK_SCALAR=0 MPCX_NODEBLOCK_CSR_THREADS=$T timeout 600 python bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-sub-records --cpu-allcores 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('csr threads $T', round(d['ms_per_step'],2), [(k['kernel'], round(k['launch_ms'],2)) for k in d['roofline_kernels']][:1])"
done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "affine or vector_kernel_variants" 2>&1 | tail -2
