"""Solve of the config-2 system (periodic Poisson, P1, N^3) on the device: CG preconditioned with the smoothed-aggregation
V-cycle (dolfinx_mpc_amd/amg.py) against the fused Jacobi-CG kernels.  One JSON line.
    python tools/bench_multigrid.py [N] [rtol] [degree]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from dolfinx_mpc_amd.problem import LinearProblem, cg, multigrid_cg  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rtol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-8
degree = int(sys.argv[3]) if len(sys.argv) > 3 else 1
import argparse  # noqa: E402

args = argparse.Namespace(n=N, no_tile=False, tile=[8, 8, 8], cell="tet", scaling="strong", ufcx=None, numbering="tiled")
w = bench.poisson_workload(args, 0, 1, degree)
V, a, L, mpc = w.V, w.blocks[0][1], w.vectors[0][1], w.vectors[0][2]
prob = LinearProblem(a, L, mpc, w.bcs)
A, b = prob.assemble()
torch.cuda.synchronize()
out = {"N": N, "degree": degree, "dofs": int(A.shape[0]), "nnz": int(A.nnz), "rtol": rtol}
x, info = multigrid_cg(A, b, V, rtol=rtol)
out["gamg"] = info
torch.cuda.synchronize()
t0 = time.perf_counter()
xj, infoj = cg(A, b, rtol=rtol, max_it=20000, check_every=50)
torch.cuda.synchronize()
infoj["solve_s"] = time.perf_counter() - t0
out["jacobi"] = infoj
out["solution_difference"] = float((x.array - xj.array).abs().max() / xj.array.abs().max())
print(json.dumps(out))
