#!/usr/bin/env python
"""Run the config-2 hot kernels a few times, for rocprofv3 (--kernel-trace / --pmc):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o name -- python tools/profile_kernels.py [N] [reps]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import dolfinx_mpc_amd as dm  # noqa: E402
from dolfinx_mpc_amd.la import MPCMatrix, create_vector  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mesh, V, bc, mpc, a, L = bench.build_problem(N, (8, 8, 8))
rowptr, cols = dm.create_sparsity_pattern(a, mpc)
A = MPCMatrix(rowptr, cols, V.num_dofs)
b = create_vector(V)
for _ in range(reps):
    dm.assemble_matrix(a, mpc, bcs=[bc], A=A, algorithm=os.environ.get("MPCX_MATRIX_ALG", "rowblock"))
    dm.assemble_vector(L, mpc, b=b)
    dm.apply_lifting(b, [a], [[bc]], mpc)
torch.cuda.synchronize()
print("done", float(A.vals.sum()), float(b.array.sum()))
