"""Assembly times of the other configurations' element types on one GPU (documentation, not the
bench line): periodic Poisson P2, P1 elasticity (bs = 3) with the slip-type constraint of the tests.

    python tools/bench_configs.py [N_p2] [N_elasticity]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import dolfinx_mpc_amd as dm  # noqa: E402
from dolfinx_mpc_amd import fem  # noqa: E402
from dolfinx_mpc_amd.mesh import create_unit_cube  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def periodic(N, degree, bs):
    mesh = create_unit_cube(N, N, N, reorder=(8, 8, 8))
    V = fem.functionspace(mesh, ("Lagrange", degree)) if bs == 1 else fem.functionspace(mesh, ("Lagrange", degree), (bs,))
    walls = fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[1], 0) | np.isclose(x[1], 1) | np.isclose(x[2], 0) | np.isclose(x[2], 1))
    bc = fem.dirichletbc(0.0 if bs == 1 else np.zeros(bs), walls, V)
    mpc = dm.MultiPointConstraint(V)

    def rel(x):
        out = x.copy()
        out[0] = 1 - x[0]
        return out

    mpc.create_periodic_constraint_geometrical(V, lambda x: np.isclose(x[0], 1), rel, [bc])
    mpc.finalize()
    return mesh, V, bc, mpc


out = {}
N2 = int(sys.argv[1]) if len(sys.argv) > 1 else 96
mesh, V, bc, mpc = periodic(N2, 2, 1)
a, L = fem.form_stiffness(V), fem.form_source(V, fem.FN_BENCH_PERIODIC)
t0 = time.time()
A = dm.create_matrix(a, mpc)
t_pat = time.time() - t0
b = dm.assemble_vector(L, mpc)
for alg in ("rowblock", "atomic"):
    tm = timed(lambda: dm.assemble_matrix(a, mpc, bcs=[bc], A=A, algorithm=alg))
    out[f"P2 periodic Poisson N={N2} matrix {alg} ms"] = tm
tv = timed(lambda: dm.assemble_vector(L, mpc, b=b))
out[f"P2 periodic Poisson N={N2}"] = {"cells": mesh.num_cells, "dofs": V.num_dofs, "nnz": int(A.nnz), "pattern_s": t_pat,
                                       "vector_ms": tv}
del A, b, a, L, mpc, V, mesh

N3 = int(sys.argv[2]) if len(sys.argv) > 2 else 128
mesh, V, bc, mpc = periodic(N3, 1, 3)
a = fem.form_elasticity(V, 1.0, 0.5)
L = fem.form_source(V, fem.FN_CONSTANT_VEC, constant=np.array([1.0, 0.1, -0.2, 0.3]))
t0 = time.time()
A = dm.create_matrix(a, mpc)
t_pat = time.time() - t0
b = dm.assemble_vector(L, mpc)
for alg in ("rowblock", "atomic"):
    tm = timed(lambda: dm.assemble_matrix(a, mpc, bcs=[bc], A=A, algorithm=alg))
    out[f"P1^3 periodic elasticity N={N3} matrix {alg} ms"] = tm
tv = timed(lambda: dm.assemble_vector(L, mpc, b=b))
out[f"P1^3 periodic elasticity N={N3}"] = {"cells": mesh.num_cells, "dofs": V.num_dofs, "nnz": int(A.nnz), "pattern_s": t_pat,
                                            "vector_ms": tv}
print(json.dumps(out))
