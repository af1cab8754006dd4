#!/bin/bash
# numbering-tile sweep: tools/tile_sweep.sh CONFIG "tx ty tz" ...
OUT=gpurun_out/tile_sweep; mkdir -p $OUT
c=$1; shift
for t in "$@"; do
  name=c${c}_$(echo $t | tr ' ' 'x')
  python bench.py --config $c --tile $t --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $OUT/$name.json 2> $OUT/$name.log
  python tools/show_bench.py $OUT/$name.json | grep -vE "roofline|generic" | tr '\n' ' '; echo
done
