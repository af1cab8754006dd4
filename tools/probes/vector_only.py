"""config 2's vector assembly alone, N times (for rocprofv3 --kernel-trace): python tools/probes/vector_only.py [n=256] [reps=10]"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import dolfinx_mpc_amd as dm
from dolfinx_mpc_amd.la import create_vector

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ap = types.SimpleNamespace(n=n, no_tile=False, tile=(8, 8, 8), ufcx=None, numbering="tiled", cell="tetrahedron", config=2, alg=None)
w = bench.poisson_workload(ap, 0, 1, 1)
label, f, m = w.vectors[0]
b = create_vector(m.function_space)
for _ in range(3):
    dm.assemble_vector(f, m, b=b)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
from dolfinx_mpc_amd.la import wait_assembly
ev[0].record()
for i in range(reps):
    dm.assemble_vector(f, m, b=b)
    wait_assembly()
    ev[i + 1].record()
torch.cuda.synchronize()
print("vector call ms:", min(ev[i].elapsed_time(ev[i + 1]) for i in range(reps)))
