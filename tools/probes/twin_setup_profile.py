#!/usr/bin/env python
"""Where the set-up time of a mesh without locality goes (33 s against 6.6 s tiled at config 2): the
harness's own work (shuffling the mesh, building space / bc / constraint on it) apart from the library's (the Morton twin:
orders, renumbered mesh, twin space, twin constraint, twin pattern, hand-back index, plans of the first call)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    import torch

    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import MultiPointConstraint, fem, locality
    from dolfinx_mpc_amd.la import create_vector
    from dolfinx_mpc_amd.mesh import create_box, renumber

    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    T = {}

    def tick(name, t0):
        torch.cuda.synchronize()
        T[name] = round(time.time() - t0, 3)
        print(f"{name}: {T[name]} s", flush=True)
        return time.time()

    t = time.time()
    mesh0 = create_box((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), (N, N, N), "tetrahedron", None)
    t = tick("harness: create_box", t)
    rng = np.random.default_rng(0)
    mesh = renumber(mesh0, rng.permutation(mesh0.num_nodes), rng.permutation(mesh0.num_cells))
    del mesh0
    t = tick("harness: shuffle (renumber)", t)
    V = fem.functionspace(mesh, ("Lagrange", 1))
    bc = fem.dirichletbc(0.0, fem.locate_dofs_geometrical(
        V, lambda x: np.isclose(x[1], 0) | np.isclose(x[1], 1) | np.isclose(x[2], 0) | np.isclose(x[2], 1)), V)
    t = tick("harness: space + bc", t)
    mpc = MultiPointConstraint(V)

    def rel(x):
        o = x.copy()
        o[0] = 1 - x[0]
        return o

    mpc.create_periodic_constraint_geometrical(V, lambda x: np.isclose(x[0], 1), rel, [bc])
    mpc.finalize()
    t = tick("harness: constraint", t)
    fa, fl = fem.form_stiffness(V), fem.form_source(V, fem.FN_BENCH_PERIODIC)
    A = dm.create_matrix(fa, mpc)
    b = create_vector(V)
    t = tick("library: create_matrix (caller numbering)", t)
    t_lib0 = time.time()
    t0 = time.time()
    perm, cell_order = locality._morton_orders(mesh)
    t0 = tick("library/twin: morton orders", t0)
    tw = locality.twin_of(mesh)
    t0 = tick("library/twin: Twin() (orders again + renumbered mesh)", t0)
    tw.space(V)
    t0 = tick("library/twin: twin space", t0)
    tw.mpc(mpc)
    t0 = tick("library/twin: twin constraint", t0)
    tw.form(fa), tw.form(fl)
    t0 = tick("library/twin: twin forms", t0)
    tw.matrix(A, fa, mpc, mpc)
    t0 = tick("library/twin: twin matrix (pattern + hand-back index)", t0)
    dm.assemble_matrix(fa, mpc, bcs=[bc], A=A)
    t0 = tick("library: first assemble_matrix (plans)", t0)
    dm.assemble_vector(fl, mpc, b=b)
    t0 = tick("library: first assemble_vector (plans)", t0)
    print("library total after create_matrix:", round(time.time() - t_lib0, 2), "s")


if __name__ == "__main__":
    main()
