python tools/probes/b0_probe.py 128 2>&1 | tail -2
for r in 1536 3072 12288; do echo "MPCX_VECTOR_BLOCK_ROWS=$r"; MPCX_VECTOR_BLOCK_ROWS=$r python tools/probes/b0_probe.py 128 2>&1 | tail -2; done
echo "force rowblock"; MPCX_FORCE_KERNEL=vector=rowblock python tools/probes/b0_probe.py 128 2>&1 | tail -2
echo "force hash"; MPCX_FORCE_KERNEL=vector=hash python tools/probes/b0_probe.py 128 2>&1 | tail -2
echo "owner rows 2048"; MPCX_VECTOR_OWNER_ROWS=2048 python tools/probes/b0_probe.py 128 2>&1 | tail -2
