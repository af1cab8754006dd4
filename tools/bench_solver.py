"""SpMV / CG timing on the config-2 system (periodic Poisson, P1, N^3 cube): the caller of the
assembly path.  Prints one JSON line.

    python tools/bench_solver.py [N] [rtol]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from dolfinx_mpc_amd import _device as D  # noqa: E402
from dolfinx_mpc_amd import _native  # noqa: E402
from dolfinx_mpc_amd.la import Vector  # noqa: E402
from dolfinx_mpc_amd.problem import LinearProblem, cg, spmv  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rtol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-8
import argparse  # noqa: E402

args = argparse.Namespace(n=N, no_tile=False, tile=[8, 8, 8], cell="tet", scaling="strong", ufcx=None, numbering="tiled")
w = bench.poisson_workload(args, 0, 1, 1)
V, a, L, mpc = w.V, w.blocks[0][1], w.vectors[0][1], w.vectors[0][2]
prob = LinearProblem(a, L, mpc, w.bcs, solver_options={"rtol": rtol, "max_it": 20000, "check_every": 50})
A, b = prob.assemble()
torch.cuda.synchronize()
n, nnz = A.shape[0], A.nnz

# SpMV
x = Vector(n)
x.array.normal_()
y = Vector(n)
for _ in range(3):
    spmv(A, x, y)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for s, e in ev:
    s.record()
    spmv(A, x, y)
    e.record()
torch.cuda.synchronize()
t_spmv = sum(s.elapsed_time(e) for s, e in ev) / len(ev)
spmv_bytes = 12 * nnz + 20 * n

# one CG iteration (steady state)
Lib = _native.lib()
work = torch.empty((5, n), dtype=torch.float64, device=A.vals.device)
scal = torch.zeros(8, dtype=torch.float64, device=A.vals.device)
xs = Vector(n)
st = D.stream_ptr()
Lib.mpcx_inverse_diagonal(n, A.d_rowptr.data_ptr(), A.d_cols.data_ptr(), A.vals.data_ptr(), work[0].data_ptr(), st)
Lib.mpcx_cg_start(n, work[0].data_ptr(), b.array.data_ptr(), xs.array.data_ptr(), work[1].data_ptr(),
                  work[2].data_ptr(), work[3].data_ptr(), scal.data_ptr(), st)
def steps(k0, k1):
    for k in range(k0, k1):
        Lib.mpcx_cg_step(n, A.d_rowptr.data_ptr(), A.d_cols.data_ptr(), A.vals.data_ptr(), work[0].data_ptr(),
                         xs.array.data_ptr(), work[1].data_ptr(), work[2].data_ptr(), work[3].data_ptr(),
                         work[4].data_ptr(), scal.data_ptr(), k, st)
steps(0, 10)
torch.cuda.synchronize()
t0 = time.perf_counter()
steps(10, 110)
torch.cuda.synchronize()
t_iter = (time.perf_counter() - t0) / 100 * 1e3
iter_bytes = spmv_bytes + 56 * n + 24 * n

# full solve
torch.cuda.synchronize()
t0 = time.perf_counter()
xsol, info = cg(A, b, rtol=rtol, max_it=20000, check_every=50)
torch.cuda.synchronize()
t_solve = time.perf_counter() - t0
print(json.dumps({
    "workload": f"periodic Poisson P1 {N}^3: n = {n}, nnz = {nnz}",
    "spmv": {"ms": t_spmv, "algorithmic_bytes": spmv_bytes, "GB/s": spmv_bytes / t_spmv / 1e6,
             "frac_of_8TB/s": spmv_bytes / t_spmv / 1e6 / 8000},
    "cg_iteration": {"ms": t_iter, "algorithmic_bytes": iter_bytes, "GB/s": iter_bytes / t_iter / 1e6,
                     "frac_of_8TB/s": iter_bytes / t_iter / 1e6 / 8000},
    "solve": {"rtol": rtol, "seconds": t_solve, **info},
}))
