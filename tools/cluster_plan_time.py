"""Time of mpcx_cluster_plan_create (the cluster set-up of MPCX_ALG_CUBE through the C ABI alone) at config-2 size, next to
the torch-driven builder of dolfinx_mpc_amd/assemble_matrix.py.   python tools/cluster_plan_time.py [N]"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import dolfinx_mpc_amd as dm  # noqa: E402
from dolfinx_mpc_amd import _device as D  # noqa: E402
from dolfinx_mpc_amd import _native  # noqa: E402
from dolfinx_mpc_amd import assemble_matrix as am_mod  # noqa: E402,F401

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
args = argparse.Namespace(n=N, no_tile=False, tile=[8, 8, 8], cell="tet", scaling="strong", ufcx=None, numbering="tiled")
w = bench.poisson_workload(args, 0, 1, 1)
V, a, mpc = w.V, w.blocks[0][1], w.vectors[0][2]
A = dm.create_matrix(a, mpc)
torch.cuda.synchronize()
t0 = time.perf_counter()
dm.assemble_matrix(a, mpc, bcs=w.bcs, A=A)
torch.cuda.synchronize()
t_first = time.perf_counter() - t0
L = _native.lib()
md = D.mesh_device(V.mesh)
_, bc = D.bc_markers(V, w.bcs, a._device)
_, t = mpc._device()
hints = np.ascontiguousarray(V.dof_tile_offsets.astype(np.int32))
rowptr_h = np.ascontiguousarray(A.rowptr.astype(np.int64))
import importlib  # noqa: E402

am = importlib.import_module("dolfinx_mpc_amd.assemble_matrix")
times = []
for _ in range(3):
    h = C.c_void_p()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = L.mpcx_cluster_plan_create(V.mesh.num_owned_cells, md["x_dofmap"].data_ptr(), V.mesh.num_nodes, md["x"].data_ptr(), A.shape[0],
                                    A.d_rowptr.data_ptr(), rowptr_h.ctypes.data, A.d_cols.data_ptr(), D.ptr(bc), t["is_slave"].data_ptr(),
                                    am.CUBE_MAX_ROWS, am.CUBE_MAX_NNZ, hints.ctypes.data, hints.size, D.stream_ptr(), C.byref(h))
    _native.check(rc, "mpcx_cluster_plan_create")
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
    info = (L.mpcx_cluster_plan_num_parts(h), L.mpcx_cluster_plan_num_clusters(h), L.mpcx_cluster_plan_num_slots(h))
    L.mpcx_cluster_plan_destroy(h)
print({"N": N, "first_assemble_matrix_s (clusters + plan through torch + first launch)": round(t_first, 3),
       "mpcx_cluster_plan_create_s": [round(v, 3) for v in times], "parts_clusters_slots": info})
