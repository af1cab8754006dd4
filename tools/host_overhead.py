"""Host-side cost of one assemble_matrix + assemble_vector call pair (tiny mesh: the GPU work is negligible)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import dolfinx_mpc_amd as dm
from problems import case_cube_periodic, product_mpc
case = case_cube_periodic(int(sys.argv[1]) if len(sys.argv) > 1 else 12, 1, 0.0, reorder=(4, 4, 4))
mpc = product_mpc(case)
A = dm.assemble_matrix(case.a, mpc, bcs=case.bcs, algorithm="rowblock")
b = dm.assemble_vector(case.L, mpc)
def step():
    dm.assemble_matrix(case.a, mpc, bcs=case.bcs, A=A, algorithm="rowblock")
    dm.assemble_vector(case.L, mpc, b=b)
for _ in range(20): step()
torch.cuda.synchronize()
n = 500
t = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t) / n * 1e6:.1f} us per step (matrix + vector)")
if len(sys.argv) > 2:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(n): step()
    pr.disable(); pstats.Stats(pr).sort_stats("cumtime").print_stats(18)
