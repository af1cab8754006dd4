for v in "" "MPCX_VCUBE_ROWS=512" "MPCX_VCUBE_ROWS=256" "MPCX_VCUBE_ROWS=512 MPCX_VECTOR_STREAM_PRIORITY=-1 MPCX_MATRIX_STREAM_PRIORITY=0" "MPCX_VCUBE_ROWS=1024"; do
  echo "== $v"
  env $v python bench.py --no-cpu-baseline --no-traffic --no-sub-records 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), [(k['kernel'], round(k['launch_ms'],3)) for k in d['roofline_kernels']])"
done
