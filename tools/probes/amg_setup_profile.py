"""where the smoothed-aggregation set-up spends its time (config 2 at N, torch profiler by op)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
from dolfinx_mpc_amd.amg import SmoothedAggregation  # noqa: E402
from dolfinx_mpc_amd.problem import LinearProblem  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
args = argparse.Namespace(n=N, no_tile=False, tile=[8, 8, 8], cell="tet", scaling="strong", ufcx=None, numbering="tiled")
w = bench.poisson_workload(args, 0, 1, 1)
V, a, L, mpc = w.V, w.blocks[0][1], w.vectors[0][1], w.vectors[0][2]
prob = LinearProblem(a, L, mpc, w.bcs)
A, b = prob.assemble()
X = V.tabulate_dof_coordinates()
for _ in range(2):
    mg = SmoothedAggregation(A.d_rowptr, A.d_cols, A.vals, X, bs=1)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    mg = SmoothedAggregation(A.d_rowptr, A.d_cols, A.vals, X, bs=1)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))
