#!/bin/bash
# kernel trace of one slab of the 8-way cut of config 2 (rank given)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/slab_trace
rm -rf $OUT; mkdir -p $OUT
for r in "$@"; do
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/trace$r -o t -- python $GRAFT_REPO_ROOT/tools/slab_trace.py 8 $r > /dev/null 2>&1 )
echo "== rank $r"
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(ls $OUT/trace$r/*results.db $OUT/trace$r/*/*results.db 2>/dev/null | head -1) | cut -c1-170 | grep -v "^#" | awk '$0 ~ / 1[0-9] | 2[0-9] | 3[0-9] | 4[0-9] |calls/' | head -24
rm -rf $OUT/trace$r
done
