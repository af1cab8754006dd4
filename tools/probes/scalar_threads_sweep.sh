for t in 256 512 1024; do echo "== MPCX_SCALAR_THREADS=$t"; MPCX_SCALAR_THREADS=$t python tools/bench_scalar_types.py 128 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d['types'].items():
    if 'rowblock' in v: print(k, v['rowblock'])"; done
