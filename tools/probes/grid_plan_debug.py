import ctypes as C, os, sys, importlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import dolfinx_mpc_amd as dm
from dolfinx_mpc_amd import _device as D, _native
from dolfinx_mpc_amd.la import create_vector
from dolfinx_mpc_amd.workloads import case_cube_periodic
from test_gpu_cluster_plan import _dev_array
from test_gpu_parity import product_mpc
N = int(sys.argv[1])
av = importlib.import_module("dolfinx_mpc_amd.assemble_vector")
case = case_cube_periodic(N, 1, 0.0, reorder=(8, 8, 8))
mpc = product_mpc(case)
args, keep = av.vector_args(case.L, 0, create_vector(case.V), mpc, 0)
L = _native.lib()
h = C.c_void_p()
rc = L.mpcx_grid_plan_create(args.cube_verts, int(args.n_cubes), args.x, C.byref(args.plan), D.stream_ptr(), C.byref(h))
torch.cuda.synchronize()
print("rc", rc, "python staged", bool(args.grid_block_rows), int(args.grid_block_rows_max), "C rows", L.mpcx_grid_plan_block_rows(h) if h else None,
      "blocks", int(args.plan.num_blocks), "max_rows", int(args.plan.max_rows))
nc = int(args.n_cubes)
ns = [L.mpcx_grid_plan_num_intervals(h, d) for d in range(3)]
print("ns", ns, [int(args.grid_n[d]) for d in range(3)])
a2 = _native.VectorArgs.from_buffer_copy(args)
_native.check(L.mpcx_grid_plan_fill(h, C.byref(a2)), "fill")
print("iv equal", np.array_equal(_dev_array(a2.grid_iv, 2 * sum(ns), np.float64), _dev_array(args.grid_iv, 2 * sum(ns), np.float64)))
gi, ri = _dev_array(a2.grid_idx, 4 * nc, np.int32).reshape(nc, 4), _dev_array(args.grid_idx, 4 * nc, np.int32).reshape(nc, 4)
print("idx equal", np.array_equal(gi[:, :3], ri[:, :3]), gi.max(axis=0), ri.max(axis=0))
if bool(args.grid_block_rows) and bool(a2.grid_block_rows):
    nb = int(args.plan.num_blocks)
    print("rows equal", np.array_equal(_dev_array(a2.grid_block_rows, 128 * nb, np.int32), _dev_array(args.grid_block_rows, 128 * nb, np.int32)))
b2 = create_vector(case.V)
a2.b = b2.array.data_ptr()
_native.check(L.mpcx_assemble_vector(C.byref(a2)), "assemble")
torch.cuda.synchronize()
got = dm.assemble_vector(case.L, mpc).numpy()
print("b diff", abs(b2.numpy() - got).max(), abs(got).max())
