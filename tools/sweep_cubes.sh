run() { env "$@" python bench.py --steps 8 --warmup 2 --no-traffic --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print('$*', '%.3g'%d['value'], '%.2f'%d['ms_per_step'], {k:round(v,3) for k,v in d['timings_ms'].items()}, [(k['kernel'],round(k['launch_ms'],3)) for k in d['roofline_kernels']])"; }
run A=1
run MPCX_PLAN_KERNEL_BIG=1
run MPCX_MPC_PLAN=host
run MPCX_MPC_PLAN=host MPCX_PLAN_KERNEL_BIG=1
