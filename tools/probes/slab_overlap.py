"""one slab (rank given) of the 8-way cut: wall time per plain step, repeated -- does the matrix / vector overlap hold?"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import dolfinx_mpc_amd as dm
from dolfinx_mpc_amd.la import create_vector
rank = int(sys.argv[1]) if len(sys.argv) > 1 else 1
args = argparse.Namespace(n=256, no_tile=False, tile=[8, 8, 8], scaling="strong", numbering="tiled", cell="tet", ufcx=None)
w = bench.poisson_workload(args, rank, 8, 1)
label, f, (m0, m1) = w.blocks[0]
lv, fv, mv = w.vectors[0]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # matrices created before: the measured one takes stream slot `skip` mod 3
for _ in range(skip):
    D0 = dm.create_matrix(f, m0, m1)
    dm.assemble_matrix(f, (m0, m1), bcs=w.bcs, A=D0)
A = dm.create_matrix(f, m0, m1)
b = create_vector(mv.function_space)
def step():
    dm.assemble_matrix(f, (m0, m1), bcs=w.bcs, A=A)
    dm.assemble_vector(fv, mv, b=b)
for _ in range(5):
    step()
torch.cuda.synchronize()
out = []
for rep in range(8):
    t0 = time.perf_counter()
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    out.append(round((time.perf_counter() - t0) / 30 * 1e6))
print("rank", rank, "slot", getattr(A, "_side_slot", None), "us per step", out, "GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"))
