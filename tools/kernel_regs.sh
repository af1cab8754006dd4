#!/bin/bash
# VGPR / scratch use of the kernels of one translation unit: tools/kernel_regs.sh mpcx_cubes.hip [name filter]
cd /root/repo/dolfinx_mpc_amd/csrc || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -I../../include -I. --offload-arch=gfx950 -munsafe-fp-atomics -c "$1" -o /tmp/kr.o --save-temps=obj 2>&1 | grep -v "warning\|^ \|generated" | head
base=$(basename "$1" .hip)
grep -E "^\s+\.(vgpr_count|sgpr_count|private_segment_fixed_size|name):" /tmp/${base}-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - - | grep "${2:-.}" | sed -E 's/\s+/ /g'
