#!/bin/bash
# round 6, GPU session 2: vertex-moment sources, CSR-valued node-block occupancy variants, power / clock while co-running, timelines
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6s2; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_corun.py tests/test_stokes.py tests/test_gpu_ufcx_clusters.py "tests/test_gpu_parity.py::test_small_cases_match_oracle" "tests/test_gpu_parity.py::test_small_cases_vector_kernel_variants" tests/test_gpu_independent.py -x -q -m gpu > $OUT/parity.log 2>&1; tail -5 $OUT/parity.log
c3() { name=$1; shift
  env "$@" timeout 600 python bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-sub-records --cpu-allcores 0 > $OUT/$name.json 2> $OUT/$name.log
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    print("$name: step", round(d["ms_per_step"],3), [(k["kernel"], round(k["launch_ms"],3)) for k in d.get("roofline_kernels",[])])
except Exception as e:
    print("$name failed", e)
PY
}
c3 c3_csr_v0 MPCX_BLOCK_SCALAR=0
c3 c3_csr_v6 MPCX_BLOCK_SCALAR=0 MPCX_NODEBLOCK_CSR_VARIANT=6
c3 c3_csr_v8 MPCX_BLOCK_SCALAR=0 MPCX_NODEBLOCK_CSR_VARIANT=8
c3 c3_default
c3 c3_rule_walked MPCX_VERTEX_SOURCE=0
timeout 900 python tools/probes/power_probe.py --config 5 > $OUT/power_c5.log 2>&1; grep -E "^ARM|^#" $OUT/power_c5.log | cut -c1-700
# kernel timelines of the config-5 step: the two library streams left alone, and with the first part of the matrix launch capped
for arm in off corun; do
  if [ $arm = corun ]; then EXTRA="MPCX_CORUN=1 MPCX_CORUN_FRAC=0.999 MPCX_CORUN_MATRIX_WGS=3 MPCX_VECTOR_OWNER_ROWS=3072"; else EXTRA="MPCX_CORUN=0"; fi
  T=$OUT/trace_$arm
  (cd /tmp && env $EXTRA timeout 900 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$T -o t -- python $GRAFT_REPO_ROOT/bench.py --config 5 --no-cpu-baseline --no-sub-records --no-traffic --cpu-allcores 0 --steps 4 --warmup 3 > $GRAFT_REPO_ROOT/$T.log 2>&1)
  DB=$(ls $T/*results.db $T/*/*results.db 2>/dev/null | head -1)
  python tools/rocprof_timeline.py $DB 60 100 | cut -c1-170 > $OUT/timeline_c5_$arm.txt
  tail -24 $OUT/timeline_c5_$arm.txt
  rm -rf $T
done
