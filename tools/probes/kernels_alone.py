"""config 2's matrix call and vector call, each alone, N times (for rocprofv3 --kernel-trace: the per-kernel durations that
bench.py's HIP-event figures must agree with):  python tools/probes/kernels_alone.py [n=256] [reps=10]"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import dolfinx_mpc_amd as dm
from dolfinx_mpc_amd.la import create_vector, wait_assembly

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ap = types.SimpleNamespace(n=n, no_tile=False, tile=(8, 8, 8), ufcx=None, numbering="tiled", cell="tet", config=2, alg=None)
w = bench.poisson_workload(ap, 0, 1, 1)
label, f, m = w.vectors[0]
la, fa, (m0, m1) = w.blocks[0]
b = create_vector(m.function_space)
A = dm.create_matrix(fa, m0, m1)
for _ in range(3):
    dm.assemble_matrix(fa, (m0, m1), bcs=w.bcs, A=A)
    wait_assembly()
    torch.cuda.synchronize()
    dm.assemble_vector(f, m, b=b)
    wait_assembly()
    torch.cuda.synchronize()
for _ in range(reps):
    dm.assemble_matrix(fa, (m0, m1), bcs=w.bcs, A=A)
    wait_assembly()
    torch.cuda.synchronize()
for _ in range(reps):
    dm.assemble_vector(f, m, b=b)
    wait_assembly()
    torch.cuda.synchronize()
print("done")
