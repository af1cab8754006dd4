"""P2 vector-kernel time against the number of quadrature points."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dolfinx_mpc_amd as dm
from dolfinx_mpc_amd import fem
from dolfinx_mpc_amd.mesh import create_unit_cube
from dolfinx_mpc_amd.la import create_vector

N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
mesh = create_unit_cube(N, N, N, reorder=(8, 8, 8))
V = fem.functionspace(mesh, ("Lagrange", 2))
mpc = dm.MultiPointConstraint(V)
mpc.finalize()
b = create_vector(V)
for fn, name in ((fem.FN_BENCH_PERIODIC, "bench f"), (fem.FN_ONE, "f=1")):
    for deg in (1, 2, 4, 6):
        Lq = fem.form_source(V, fn, quadrature_degree=deg)
        nq = Lq.integrals[0].kernel.qwts.size
        for alg in ("atomic", "rowblock"):
            for _ in range(2):
                dm.assemble_vector(Lq, mpc, b=b, algorithm=alg)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                dm.assemble_vector(Lq, mpc, b=b, algorithm=alg)
            torch.cuda.synchronize()
            print(f"{name:8s} degree {deg} nq {nq:3d} {alg:8s}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms", flush=True)
