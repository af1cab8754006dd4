#!/bin/bash
# round 6, GPU session 3: FFCx-layout files on the GPU, imported text on configs 3/4/5, kernel split of config 3, timelines
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6s3; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_ffcx_layout.py tests/test_stokes.py tests/test_dof_transformations.py tests/test_gpu_ufcx_clusters.py tests/test_ufcx_generated.py -x -q -m gpu -n 2 > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
for C in 3 4 5; do
  timeout 1500 python bench.py --config $C --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --cpu-allcores 0 > $OUT/c$C.json 2> $OUT/c$C.log
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/c$C.json").read().strip().splitlines()[-1])
    print("config $C: step", round(d["ms_per_step"],3), "csr-valued", d.get("ms_per_step_csr_valued"), [(k["kernel"], round(k["launch_ms"],3)) for k in d.get("roofline_kernels",[])])
    print("   ufcx text:", json.dumps(d.get("roofline_ufcx_text"))[:900])
except Exception as e:
    print("config $C failed", e)
PY
done
# kernel trace of config 3 (which kernel of the b0 call takes the time?) and timelines of configs 3 / 5
for C in 3 5; do
  T=$OUT/trace_c$C
  (cd /tmp && MPCX_CORUN=0 timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$T -o t -- python $GRAFT_REPO_ROOT/bench.py --config $C --no-cpu-baseline --no-sub-records --no-traffic --cpu-allcores 0 --steps 4 --warmup 3 > $GRAFT_REPO_ROOT/$T.log 2>&1)
  DB=$(ls $T/*results.db $T/*/*results.db 2>/dev/null | head -1)
  python tools/rocprof_summary.py $DB | cut -c1-160 | head -14 > $OUT/summary_c$C.txt
  python tools/rocprof_timeline.py $DB 40 20 matrix_pairs | cut -c1-170 > $OUT/timeline_c$C.txt
  cat $OUT/summary_c$C.txt; tail -34 $OUT/timeline_c$C.txt
  rm -rf $T
done
