#!/bin/bash
# counters of the tensor-grid cluster kernel
cd /root/repo
mkdir -p gpurun_out/r6s27
timeout 1500 python tools/collect_pmc.py gpurun_out/r6s27/pmc 256 2 > gpurun_out/r6s27/collect.txt 2>&1
tail -3 gpurun_out/r6s27/collect.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6s27/pmc/pmc_summary.json'))
for name,k in d.items():
    if 'vector_cube_grid' in name or 'matrix_cube_affine_kernel<true>' in name:
        print(name[:100])
        for f in sorted(k): print('   ',f,k[f])
PY
