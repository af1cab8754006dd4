#!/bin/bash
# N > 1 path of bench.py on a ONE-GPU box: two / four ranks share cuda:0, transport gloo with host staging
# (RCCL refuses two ranks on one device).  Everything but the RCCL transport itself is exercised.
export MPCX_DIST_BACKEND=gloo
run() { echo "== $*"; timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 1000)) bench.py --gpus $1 --steps 4 --warmup 1 --no-traffic --no-cpu-baseline "${@:2}" 2>/tmp/err.log | python -c "import json,sys;d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]);print(d['n_gpus'], d['scaling'], '%.3g'%d['value'], '%.2f ms'%d['ms_per_step'], d['config']['parallelism'], d['config']['dofs_global'])" || tail -5 /tmp/err.log; }
run 2 --config 2 --size 128
run 4 --config 2 --size 96 --scaling strong
run 2 --config 2 --size 96 --scaling weak
run 2 --config 4 --size 24
run 2 --config 5 --size 48
run 2 --config 5 --size 40 --scaling weak
