# the hand-back pass of the reordered twin (permute_values_kernel) as a bounded-grid kernel: step time of the shuffled config 2
for w in 1024 2048 4096 1000000; do
  echo "== MPCX_PERMUTE_WGS=$w"
  MPCX_PERMUTE_WGS=$w python bench.py --no-cpu-baseline --no-traffic --no-config-records 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline_spatial']; print(round(d['ms_per_step'],3), {k:s[k] for k in ('ms_per_step','ms_per_step_lazy_handback','set_up_and_first_step_s','set_up_split_s','timings_ms')})"
done
