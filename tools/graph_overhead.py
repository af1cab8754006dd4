"""Host cost of a steady-state step: plain calls vs one HIP-graph replay (dolfinx_mpc_amd/graph.py), on a tiny mesh (the
GPU work is negligible) and on a mesh of the size one rank holds at 8 GPUs of config 2 (256 x 256 x 32 cubes)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import dolfinx_mpc_amd as dm  # noqa: E402
from dolfinx_mpc_amd import fem  # noqa: E402
from dolfinx_mpc_amd.graph import CapturedStep  # noqa: E402
from dolfinx_mpc_amd.mesh import create_box  # noqa: E402


def problem(n):
    mesh = create_box((0, 0, 0), (1, 1, n[2] / n[0]), n, "tetrahedron", (8, 8, 8))
    V = fem.functionspace(mesh, ("Lagrange", 1))
    zmax = n[2] / n[0]
    bc = fem.dirichletbc(0.0, fem.locate_dofs_geometrical(
        V, lambda x: np.isclose(x[1], 0) | np.isclose(x[1], 1) | np.isclose(x[2], 0) | np.isclose(x[2], zmax)), V)
    mpc = dm.MultiPointConstraint(V)

    def rel(x):
        o = x.copy()
        o[0] = 1 - x[0]
        return o

    mpc.create_periodic_constraint_geometrical(V, lambda x: np.isclose(x[0], 1), rel, [bc])
    mpc.finalize()
    a, L = fem.form_stiffness(V), fem.form_source(V, fem.FN_BENCH_PERIODIC)
    A = dm.assemble_matrix(a, mpc, bcs=[bc])
    b = dm.assemble_vector(L, mpc)

    def step():
        dm.assemble_matrix(a, mpc, bcs=[bc], A=A)
        dm.assemble_vector(L, mpc, b=b)

    return step, A, b, V


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    t_host = time.perf_counter() - t  # the loop has returned: everything is enqueued
    torch.cuda.synchronize()
    return t_host / n * 1e6, (time.perf_counter() - t) / n * 1e6


for n, reps in (((8, 8, 8), 400), ((256, 256, 32), 200)):
    step, A, b, V = problem(n)
    ref_vals, ref_b = None, None
    step()
    torch.cuda.synchronize()
    ref_vals, ref_b = A.vals.clone(), b.array.clone()
    h0, w0 = timed(step, reps)
    g = CapturedStep(step)
    A.vals.zero_()
    b.array.zero_()
    g.replay()
    torch.cuda.synchronize()
    same = bool((A.vals - ref_vals).abs().max() <= 1e-12 * ref_vals.abs().max()) and bool((b.array - ref_b).abs().max() <= 1e-12 * ref_b.abs().max())
    h1, w1 = timed(g.replay, reps)
    print(f"{n}: {V.num_dofs} dofs  plain calls: host {h0:.1f} us / step, wall {w0:.1f} us;  graph replay: host {h1:.1f} us, wall {w1:.1f} us;  "
          f"replay reproduces the step: {same}", flush=True)
