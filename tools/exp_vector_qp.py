"""Vector-kernel time against the number of quadrature points (per-cell overhead vs per-point cost)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import dolfinx_mpc_amd as dm
from dolfinx_mpc_amd import fem
from dolfinx_mpc_amd.la import create_vector

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mesh, V, bc, mpc, a, L = bench.build_problem(N, (8, 8, 8), 0, 1)
b = create_vector(V)
for fn, name in ((fem.FN_BENCH_PERIODIC, "bench f"), (fem.FN_ONE, "f=1")):
    for deg in (1, 2, 5):
        Lq = fem.form_source(V, fn, quadrature_degree=deg)
        nq = Lq.integrals[0].kernel.qwts.size
        for _ in range(3):
            dm.assemble_vector(Lq, mpc, b=b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            dm.assemble_vector(Lq, mpc, b=b)
        torch.cuda.synchronize()
        print(f"{name:8s} degree {deg} nq {nq:3d}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms", flush=True)
