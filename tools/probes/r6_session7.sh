#!/bin/bash
# round 6, GPU session 7: pair records for imported text: parity + timing
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6s7; mkdir -p $OUT
export TMPDIR=/tmp
for T in tests/test_ufcx_generated.py tests/test_stokes.py tests/test_gpu_driver.py tests/test_ffcx_layout.py tests/test_gpu_graph.py; do
  ( time timeout 900 python -m pytest $T -x -q -m gpu > $OUT/$(basename $T .py).log 2>&1 ) 2> $OUT/$(basename $T .py).time
  echo "$T: $(tail -1 $OUT/$(basename $T .py).log) $(grep real $OUT/$(basename $T .py).time)"
done
run() { C=$1; name=$2; shift; shift
  env "$@" timeout 1500 python bench.py --config $C --ufcx generated --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-sub-records --cpu-allcores 0 > $OUT/$name.json 2> $OUT/$name.log
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    print("$name: step", round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["timings_ms"].items()}, [k["kernel"] for k in d["roofline_kernels"]])
except Exception as e:
    print("$name failed", e); print(open("$OUT/$name.log").read()[-600:])
PY
}
for C in 5 3 4; do
  run $C c${C}_pairs
  run $C c${C}_pairs_t512 MPCX_UFCX_RB_THREADS=512
  run $C c${C}_pairs_t128 MPCX_UFCX_RB_THREADS=128
  run $C c${C}_rowblock MPCX_UFCX_PAIRS=0
done
