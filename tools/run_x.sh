run() { echo "== $*"; env "$@" python bench.py --config $C --no-traffic --no-cpu-baseline --steps 10 --warmup 3 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], [(k['kernel'][:40], round(k['launch_ms'],3), round(k['hbm_frac'],3)) for k in d.get('roofline_kernels',[])][:6])
    elif 'Error' in l or 'error' in l: print(l.rstrip())
"; }
C=5 run MPCX_X=1
C=4 run MPCX_X=1
C=3 run MPCX_X=1
MPCX_NO_CUBE=1 python bench.py --config 2 --no-traffic --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('c2 nocube', d['value'], d['ms_per_step'], d['timings_ms'])
"
