#!/bin/bash
# round 6, GPU session 5: new tests, imported text after the per-node copies, affine source kernel, graph replay, config-4 timeline
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6s5; mkdir -p $OUT
export TMPDIR=/tmp
for T in tests/test_gpu_driver.py tests/test_ffcx_layout.py tests/test_stokes.py tests/test_gpu_corun.py tests/test_ufcx_generated.py; do
  ( time timeout 900 python -m pytest $T -x -q -m gpu -n 2 > $OUT/$(basename $T .py).log 2>&1 ) 2> $OUT/$(basename $T .py).time
  echo "$T: $(tail -1 $OUT/$(basename $T .py).log) $(grep real $OUT/$(basename $T .py).time)"
done
run() { C=$1; name=$2; shift; shift
  env "$@" timeout 1500 python bench.py --config $C --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --cpu-allcores 0 > $OUT/$name.json 2> $OUT/$name.log
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    print("$name: step", round(d["ms_per_step"],3), "graph", d.get("ms_per_step_graph"), "csr", d.get("ms_per_step_csr_valued"), {k: round(v,3) for k,v in d["timings_ms"].items()})
    u=d.get("roofline_ufcx_text") or {}
    print("    text:", u.get("ms_per_step"), u.get("timings_ms"), u.get("error"))
except Exception as e:
    print("$name failed", e); print(open("$OUT/$name.log").read()[-600:])
PY
}
run 3 c3
run 3 c3_general_b0 MPCX_AFFINE_OWNBLOCK=0
run 4 c4
run 5 c5
T=$OUT/trace_c4
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$T -o t -- python $GRAFT_REPO_ROOT/bench.py --config 4 --no-cpu-baseline --no-sub-records --no-traffic --cpu-allcores 0 --steps 4 --warmup 3 > $GRAFT_REPO_ROOT/$T.log 2>&1)
DB=$(ls $T/*results.db $T/*/*results.db 2>/dev/null | head -1)
python tools/rocprof_timeline.py $DB 60 3 matrix_cube_elasticity | cut -c1-170 > $OUT/timeline_c4.txt
tail -40 $OUT/timeline_c4.txt
rm -rf $T
