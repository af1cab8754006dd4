"""config 4 (two-body contact elasticity): one slab of the `world`-way cut alone on this GPU, wall time per plain step"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import dolfinx_mpc_amd as dm
from dolfinx_mpc_amd.la import create_vector
world, rank = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 56
args = argparse.Namespace(n=n, no_tile=False, tile=[8, 8, 8], scaling="strong")
w = bench.contact_workload(args, rank, world)
label, f, (m0, m1) = w.blocks[0]
lv, fv, mv = w.vectors[0]
A = dm.create_matrix(f, m0, m1)
b = create_vector(mv.function_space)
def step():
    dm.assemble_matrix(f, (m0, m1), bcs=w.bcs, A=A)
    dm.assemble_vector(fv, mv, b=b)
    dm.apply_lifting(b, [f], [w.bcs], mv)
for _ in range(5):
    step()
torch.cuda.synchronize()
tm = bench.hip_time(lambda: dm.assemble_matrix(f, (m0, m1), bcs=w.bcs, A=A), 5)
tv = bench.hip_time(lambda: dm.assemble_vector(fv, mv, b=b), 5)
out = []
for rep in range(4):
    t0 = time.perf_counter()
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    out.append(round((time.perf_counter() - t0) / 30 * 1e6))
print("world", world, "rank", rank, "dofs", mv.function_space.num_dofs, "slaves", int(mv.slaves.size), "matrix ms %.3f vector ms %.3f" % (tm, tv), "us per step", out)
