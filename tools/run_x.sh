run() { echo "== $*"; env "$@" python bench.py --config $C --no-traffic --no-cpu-baseline --steps 10 --warmup 3 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], [(k['kernel'][:40], round(k['launch_ms'],3), round(k['hbm_frac'],3)) for k in d.get('roofline_kernels',[])][:6], d['one_shot']['first_call_s'], d['one_shot']['plan_bytes'])
    elif 'Error' in l or 'error' in l: print(l.rstrip())
"; }
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
C=5 run MPCX_X=1
C=5 run MPCX_ROWBLOCK_THREADS=1024
C=5 run MPCX_ROWBLOCK_THREADS=256
