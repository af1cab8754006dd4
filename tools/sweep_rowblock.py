#!/usr/bin/env python
"""Tuning sweep of the row-block matrix kernel on the config-2 workload:
threads per workgroup x rows per block (x tile of the numbering).
    python tools/sweep_rowblock.py [N]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import dolfinx_mpc_amd as dm  # noqa: E402
from dolfinx_mpc_amd import _native  # noqa: E402
from dolfinx_mpc_amd.la import MPCMatrix  # noqa: E402

am = sys.modules["dolfinx_mpc_amd.assemble_matrix"]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tiles = [(8, 8, 8)] if len(sys.argv) < 3 else [tuple(int(c) for c in t.split("x")) for t in sys.argv[2].split(",")]
for tile in tiles:
    mesh, V, bc, mpc, a, L = bench.build_problem(N, tile)
    rowptr, cols = dm.create_sparsity_pattern(a, mpc)
    A = MPCMatrix(rowptr, cols, V.num_dofs)
    nc = mesh.num_cells
    alg_bytes = 4 * 4 * nc + 4 * 4 * nc + 24 * mesh.num_nodes + 8 * cols.size + 2 * V.num_dofs
    for max_rows, max_nnz in ((128, 2304), (256, 4608), (512, 9216)):
        am.ROWBLOCK_MAX_ROWS, am.ROWBLOCK_MAX_NNZ = max_rows, max_nnz
        am.ROWBLOCK_LIGHT_MAX_ROWS, am.ROWBLOCK_LIGHT_MAX_NNZ = max_rows, max_nnz
        for threads in (256, 384, 512, 768):
            os.environ["MPCX_ROWBLOCK_THREADS"] = str(threads)
            margs, keep = am.matrix_args(a, 0, A, mpc, mpc, [bc], 2, store_mode=1, with_mpc_kernel=False)
            info = am._rowblock_plan(A, a, 0, V)[2]
            lib = _native.lib()
            for _ in range(2):
                _native.check(lib.mpcx_assemble_matrix(C.byref(margs)), "x")
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
            for s, e in ev:
                s.record()
                _native.check(lib.mpcx_assemble_matrix(C.byref(margs)), "x")
                e.record()
            torch.cuda.synchronize()
            t = float(np.mean([s.elapsed_time(e) for s, e in ev]))
            print(f"tile={tile} rows={max_rows:4d} nnz={max_nnz:5d} threads={threads:4d}: {t:7.3f} ms  "
                  f"{alg_bytes / t / 1e6:7.1f} GB/s  blocks={info['num_blocks']} redundancy={info['num_ents'] / nc:.3f}"
                  , flush=True)
    del A
    torch.cuda.empty_cache()
