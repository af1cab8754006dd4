"""Periodic Poisson on the unit cube, the problem of the reference's python/benchmarks/bench_periodic.py:35-160 and
python/demos/demo_periodic_geometrical.py, with ``dolfinx_mpc_amd`` where the reference has ``dolfinx_mpc``:

    -div(grad u) = f  in (0, 1)^3,   u = 0 on y, z in {0, 1},   u(1, y, z) = u(0, y, z)

assemble (HIP kernels behind the C ABI) -> lifting -> set_bc -> solve (CG + smoothed-aggregation V-cycle, the
preconditioner family of the benchmark's BoomerAMG / GAMG) -> backsubstitution.

    python examples/demo_periodic_poisson.py [N] [degree]
"""
import sys
import time

import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a checkout
import dolfinx_mpc_amd  # noqa: E402
from dolfinx_mpc_amd import fem  # noqa: E402
from dolfinx_mpc_amd.mesh import create_unit_cube  # noqa: E402


def main(N: int = 32, degree: int = 1, verbose: bool = True):
    mesh = create_unit_cube(N, N, N, reorder=(8, 8, 8))
    V = fem.functionspace(mesh, ("Lagrange", degree))

    # Dirichlet walls (bench_periodic.py:52-60)
    def dirichletboundary(x):
        return np.logical_or(np.logical_or(np.isclose(x[1], 0), np.isclose(x[1], 1)),
                             np.logical_or(np.isclose(x[2], 0), np.isclose(x[2], 1)))

    bc = fem.dirichletbc(0.0, fem.locate_dofs_geometrical(V, dirichletboundary), V)
    bcs = [bc]

    # x = 1 is tied to x = 0 (bench_periodic.py:62-76)
    def periodic_boundary(x):
        return np.isclose(x[0], 1)

    def periodic_relation(x):
        out_x = np.copy(x)
        out_x[0] = 1 - x[0]
        return out_x

    mpc = dolfinx_mpc_amd.MultiPointConstraint(V)
    mpc.create_periodic_constraint_geometrical(V, periodic_boundary, periodic_relation, bcs)
    mpc.finalize()

    a = fem.form_stiffness(V)  # inner(grad(u), grad(v)) * dx
    L = fem.form_source(V, fem.FN_BENCH_PERIODIC)  # f = x sin(5 pi y) + exp(-((x - 0.5)^2 + (y - 0.5)^2) / 0.02)

    t0 = time.perf_counter()
    problem = dolfinx_mpc_amd.LinearProblem(a, L, mpc, bcs=bcs, solver_options={"pc_type": "gamg", "rtol": 1e-8})
    uh = problem.solve()
    elapsed = time.perf_counter() - t0

    # what the reference's demos check (demo_periodic_geometrical.py: compare_mpc_lhs / compare_mpc_rhs on small meshes)
    if V.num_dofs <= 40000:
        plain = dolfinx_mpc_amd.MultiPointConstraint(V)
        plain.finalize()
        A_org = dolfinx_mpc_amd.assemble_matrix(a, plain, bcs=bcs)
        dolfinx_mpc_amd.utils.compare_mpc_lhs(A_org, problem.A, mpc)
    u = uh.x.array
    off, m = mpc.masters.offsets, mpc.masters.array
    periodic_gap = max((abs(u[s] - u[m[off[s]]]) for s in mpc.slaves), default=0.0)
    info = dict(problem.info, dofs=V.num_dofs, slaves=int(mpc.slaves.size), seconds=elapsed, periodic_gap=float(periodic_gap),
                u_max=float(abs(u).max()))
    if verbose:
        print({k: (round(v, 6) if isinstance(v, float) else v) for k, v in info.items()})
    return info


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
