#!/bin/bash
# one slab of the 8-way cut (rank 1: lower ghost plane): own rows per block of the cluster vector kernel
export TMPDIR=/tmp
for vr in 512 1024 2048; do
OUT=$GRAFT_REPO_ROOT/gpurun_out/slab_rows_$vr; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && MPCX_VCUBE_ROWS=$vr rocprofv3 --kernel-trace --stats -d $OUT/t -o t -- python $GRAFT_REPO_ROOT/tools/slab_trace.py 8 ${1:-1} > /dev/null 2>&1 )
echo "rows $vr: $(python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(ls $OUT/t/*results.db $OUT/t/*/*results.db 2>/dev/null | head -1) | grep vector_cube_own | awk '{print $(NF-12), $(NF-11), $(NF-10)}')"
rm -rf $OUT
done
