"""SpMV / CG / LinearProblem on the device against scipy on the same
assembled system.  Tolerances: SpMV 1e-13 relative to |A||x| (different summation order);
the CG solution to 1e-8 relative (iterative solve to rtol 1e-12 of a system whose condition
number is O(h^-2))."""

import numpy as np
import pytest

from problems import case_cube_elasticity_slip, case_cube_periodic, product_mpc

pytestmark = pytest.mark.gpu


def _host_backsubstitute(mpc, x):
    x = x.copy()
    off, m = mpc.masters.offsets, mpc.masters.array
    c = mpc.coefficients()[0]
    for s in mpc.slaves:
        x[s] = sum(c[q] * x[m[q]] for q in range(off[s], off[s + 1]))
    return x


def test_spmv_matches_scipy():
    import torch

    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd.la import Vector
    from dolfinx_mpc_amd.problem import spmv

    case = case_cube_periodic(9, 2, 0.0)  # P2: rows of very different lengths
    mpc = product_mpc(case)
    A = dm.assemble_matrix(case.a, mpc, bcs=case.bcs)
    rng = np.random.default_rng(3)
    xh = rng.standard_normal(A.shape[1])
    x = Vector(A.shape[1])
    x.array.copy_(torch.from_numpy(xh))
    y = spmv(A, x).numpy()
    As = A.to_scipy()
    ref = As @ xh
    bound = 1e-13 * (abs(As) @ abs(xh)).max()
    assert abs(y - ref).max() <= bound


def _contact():
    from problems import case_contact_two_body

    return case_contact_two_body(4, 8, 0.0, reorder=(4, 4, 4))


@pytest.mark.parametrize("make", [lambda: case_cube_periodic(10, 1, 0.3), lambda: case_cube_periodic(5, 2, 0.0),
                                  lambda: case_cube_elasticity_slip(4), _contact],
                         ids=["p1-periodic", "p2-periodic", "elasticity-slip", "contact"])
@pytest.mark.parametrize("pc", ["jacobi", "gamg"])
def test_linear_problem_matches_direct_solve(make, pc):
    import scipy.sparse.linalg as spla

    from dolfinx_mpc_amd.problem import LinearProblem

    case = make()
    mpc = product_mpc(case)
    prob = LinearProblem(case.a, case.L, mpc, case.bcs, solver_options={"rtol": 1e-13, "max_it": 20000, "pc_type": pc})
    u = prob.solve()
    assert prob.info["converged"] and prob.info["residual_norm"] <= 1e-13 * prob.info["b_norm"]
    A, b = prob.A.to_scipy(), prob.b.numpy()
    x = spla.spsolve(A.tocsc(), b)
    x[mpc.slaves] = 0.0
    x = _host_backsubstitute(mpc, x)
    got = u.x.array
    assert abs(got - x).max() <= 1e-8 * max(1.0, abs(x).max())
    # the constraint holds on the returned function: u[slave] = sum c u[master]
    assert abs(got - _host_backsubstitute(mpc, got)).max() <= 1e-14 * max(1.0, abs(got).max())
    # Dirichlet values are met
    for bc in case.bcs:
        vals = np.zeros_like(got)
        bc.set(vals, None, 1.0)
        dofs = bc.dof_indices()[0]
        assert abs(got[dofs] - vals[dofs]).max() <= 1e-12


@pytest.mark.parametrize("degree,N", [(1, 40), (2, 20)], ids=["p1-40", "p2-20"])
def test_multigrid_iteration_counts(degree, N):
    """the smoothed-aggregation V-cycle makes the CG iteration count (nearly) independent of the mesh width: periodic
    Poisson at two resolutions, a dozen iterations where Jacobi-CG needs hundreds"""
    from dolfinx_mpc_amd.problem import LinearProblem

    its = {}
    for n in (N // 2, N):
        case = case_cube_periodic(n, degree, 0.0, reorder=(4, 4, 4))
        mpc = product_mpc(case)
        prob = LinearProblem(case.a, case.L, mpc, case.bcs, solver_options={"rtol": 1e-10, "pc_type": "gamg"})
        prob.solve()
        its[n] = prob.info["iterations"]
        assert prob.info["converged"] and len(prob.info["levels"]) >= 2, prob.info
    jac = LinearProblem(case.a, case.L, mpc, case.bcs, solver_options={"rtol": 1e-10, "check_every": 10})
    jac.solve()
    assert its[N] <= 40 and its[N] <= its[N // 2] + 8, its
    assert jac.info["iterations"] >= 3 * its[N], (jac.info["iterations"], its)
