#!/bin/bash
# round 6, GPU session 4: row-wise imported kernels (P2 / Taylor-Hood / elasticity text): parity + timing
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6s4; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_stokes.py tests/test_ufcx_generated.py tests/test_ffcx_layout.py tests/test_ufcx_import.py tests/test_element_sweep.py tests/test_hex.py -x -q -m gpu -n 2 > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
run() { C=$1; name=$2; shift; shift
  env "$@" timeout 1500 python bench.py --config $C --ufcx generated --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-sub-records --cpu-allcores 0 > $OUT/$name.json 2> $OUT/$name.log
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    print("$name: step", round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["timings_ms"].items()}, "first", round(d["one_shot"]["first_call_s"],2))
except Exception as e:
    print("$name failed", e); print(open("$OUT/$name.log").read()[-600:])
PY
}
for C in 5 3 4; do
  run $C c${C}_rowwise
  run $C c${C}_rowwise_t256 MPCX_UFCX_RB_THREADS=256
  run $C c${C}_rowwise_t512 MPCX_UFCX_RB_THREADS=512
  run $C c${C}_whole MPCX_UFCX_ROWWISE=0
done
