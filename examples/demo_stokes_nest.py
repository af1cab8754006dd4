"""Stokes flow in a channel with periodic in- and outflow, nest assembly and a field-split solve: the shape of the
reference's python/demos/demo_stokes_nest.py:150-270 and python/tests/test_stokes_channelflow.py, with ``dolfinx_mpc_amd``
where the reference has ``dolfinx_mpc``.  Taylor-Hood P2 / P1 on tetrahedra, no-slip walls y in {0, 1}, velocity AND
pressure periodic in x and z (two constraints on the rectangular blocks), body force (1, 0, 0): the exact solution is
the Poiseuille profile u = (y (1 - y) / 2, 0, 0).

    python examples/demo_stokes_nest.py [n]
"""
import sys

import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a checkout
import dolfinx_mpc_amd  # noqa: E402
from dolfinx_mpc_amd import fem  # noqa: E402
from dolfinx_mpc_amd.mesh import create_unit_cube  # noqa: E402


def main(n: int = 8, verbose: bool = True):
    mesh = create_unit_cube(n, n, n, reorder=(4, 4, 4))
    V = fem.functionspace(mesh, ("Lagrange", 2, (3,)))
    Q = fem.functionspace(mesh, ("Lagrange", 1))

    walls = fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[1], 0) | np.isclose(x[1], 1))
    bc = fem.dirichletbc(np.zeros(3), walls, V)

    def periodic_boundary(x):
        return np.isclose(x[0], 1) | np.isclose(x[2], 1)

    def periodic_map(x):
        out = x.copy()
        out[0][np.isclose(x[0], 1)] -= 1
        out[2][np.isclose(x[2], 1)] -= 1
        return out

    mpc_u = dolfinx_mpc_amd.MultiPointConstraint(V)
    mpc_u.create_periodic_constraint_geometrical(V, periodic_boundary, periodic_map, [bc])
    mpc_u.finalize()
    mpc_p = dolfinx_mpc_amd.MultiPointConstraint(Q)
    mpc_p.create_periodic_constraint_geometrical(Q, periodic_boundary, periodic_map, [])
    mpc_p.finalize()

    # a = [[inner(grad u, grad v) dx, -inner(p, div v) dx], [-inner(div u, q) dx, None]],  L = [inner(f, v) dx, 0]
    a = [[fem.form_stiffness(V), fem.form_div_test(V, Q, constant=-1.0)], [fem.form_div_trial(Q, V, constant=-1.0), None]]
    L = [fem.form_source(V, fem.FN_CONSTANT_VEC, constant=np.array([1.0, 1.0, 0.0, 0.0])), None]
    P = [[None, None], [None, fem.form_mass(Q)]]  # demo_stokes_nest.py:226-228: the pressure mass matrix preconditions p

    problem = dolfinx_mpc_amd.LinearProblem(a, L, [mpc_u, mpc_p], bcs=[bc], P=P,
                                            solver_options={"ksp_type": "minres", "pc_type": "fieldsplit", "rtol": 1e-10,
                                                            "fieldsplit_pc_types": ["gamg", "jacobi"]})
    uh, ph = problem.solve()
    x = V.tabulate_dof_coordinates()
    exact = np.zeros((x.shape[0], 3))
    exact[:, 0] = 0.5 * x[:, 1] * (1.0 - x[:, 1])
    info = dict(problem.info, dofs=V.num_dofs + Q.num_dofs, velocity_error=float(abs(uh.x.array - exact.reshape(-1)).max()),
                pressure_ptp=float(np.ptp(ph.x.array)))
    if verbose:
        print({k: (round(v, 12) if isinstance(v, float) else v) for k, v in info.items()})
    return info


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
