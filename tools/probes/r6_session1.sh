#!/bin/bash
# round 6, GPU session 1: sanity of the ABI change, co-run sweeps (configs 5, 3, generic 2), CSR-valued a00 thread sweep
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6s1; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_corun.py tests/test_stokes.py "tests/test_gpu_parity.py::test_small_cases_match_oracle" -x -q -m gpu > $OUT/parity.log 2>&1; tail -5 $OUT/parity.log
timeout 1200 python tools/probes/corun_probe.py --config 5 > $OUT/corun_c5.log 2>&1; grep -E "^ARM|^#" $OUT/corun_c5.log | cut -c1-400
timeout 900 python tools/probes/corun_probe.py --config 3 --arms off,w2_f60,w2_f100,w1_f60,w1_f100,w3_f60,w2_f60_v > $OUT/corun_c3.log 2>&1; grep -E "^ARM|^#" $OUT/corun_c3.log | cut -c1-400
timeout 900 python tools/probes/corun_probe.py --config 2 --generic --arms off,w2_f60,w2_f100,w1_f60,w1_f100,w3_f60,w2_f60_v > $OUT/corun_c2g.log 2>&1; grep -E "^ARM|^#" $OUT/corun_c2g.log | cut -c1-400
for T in 512 768 1024; do
  MPCX_CORUN=0 MPCX_BLOCK_SCALAR=0 MPCX_NODEBLOCK_CSR_THREADS=$T timeout 600 python bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-sub-records --cpu-allcores 0 > $OUT/c3_csr_t$T.json 2> $OUT/c3_csr_t$T.log
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/c3_csr_t$T.json").read().strip().splitlines()[-1])
    print("csr-valued threads $T: step", round(d["ms_per_step"],3), [(k["kernel"], round(k["launch_ms"],3)) for k in d.get("roofline_kernels",[])])
except Exception as e:
    print("csr-valued threads $T failed", e)
PY
done
