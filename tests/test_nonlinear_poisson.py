"""The reference's known-answer test for a NONLINEAR form (python/tests/test_nonlinear_assembly.py:23-114; SURVEY 8c
lists its convergence-rate criterion among the pins): quasi-linear Poisson
    F(u; v) = inner((1 + u^2) grad(u), grad(v)) dx - inner(f, v) dx = 0,   u = 0 on the boundary,
on unit squares of 4, 8, 10 cells per side with the manufactured solution u = sin(pi x) sin(pi y), whose symmetry
(x, y) -> (y, x) is imposed as a multi point constraint on the dofs of the line x = 0.5 (except the centre).  Newton's
method from the non-zero initial guess x^2 y^2; the L2 error must converge with rate > p + 0.9 for P1, P2, P3.

CPU: the oracle assembles residual, lifting and Jacobian in the reference's call order (problem.py:88-152, 26-85) and the
Newton step is a direct solve.  GPU: ``dolfinx_mpc_amd.problem.NonlinearProblem`` (HIP assembly of imported kernels with a
coefficient that changes every iteration) with both of its linear solvers; the iterates must agree with the oracle's."""

import numpy as np
import pytest
import scipy.sparse.linalg as spla

from dolfinx_mpc_amd import elements, fem
from dolfinx_mpc_amd.mesh import create_unit_square
from dolfinx_mpc_amd.quadrature import make_quadrature

# f = -div((1 + u^2) grad u) for u = sin(pi x) sin(pi y), as C text (what FFCx would bake from the UFL expression)
_U = "(sin(M_PI * x[0]) * sin(M_PI * x[1]))"
F_EXPR = (f"(2.0 * M_PI * M_PI * {_U} * (1.0 + {_U} * {_U}) - 2.0 * {_U} * M_PI * M_PI * "
          "(cos(M_PI * x[0]) * cos(M_PI * x[0]) * sin(M_PI * x[1]) * sin(M_PI * x[1]) + "
          "sin(M_PI * x[0]) * sin(M_PI * x[0]) * cos(M_PI * x[1]) * cos(M_PI * x[1])))")


def _setup(N, p):
    mesh = create_unit_square(N, N)
    V = fem.functionspace(mesh, ("Lagrange", p))
    bdofs = fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[0], 0) | np.isclose(x[0], 1) | np.isclose(x[1], 0) | np.isclose(x[1], 1))
    bc = fem.dirichletbc(0.0, bdofs, V)

    def indicator(x):  # test_nonlinear_assembly.py:48-50
        eps = 1e-12
        return np.isclose(x[0], 0.5, atol=eps) & ((x[1] < 0.5 - eps) | (x[1] > 0.5 + eps))

    def relation(x):  # :52-57
        out = np.zeros_like(x)
        out[0], out[1], out[2] = x[1], x[0], x[2]
        return out

    u = fem.Function(V)
    F, J = fem.forms_nonlinear_poisson(V, u, F_EXPR.replace("M_PI", "3.14159265358979323846"))
    return mesh, V, bc, indicator, relation, u, F, J


def _l2_error(V, uh):
    """sqrt(int (u_h - u)^2 dx) with a degree-12 rule on every triangle"""
    mesh = V.mesh
    pts, wts = make_quadrature("triangle", 12)
    phi, _ = elements.tabulate("triangle", V.degree, pts)
    gphi, _ = elements.tabulate("triangle", 1, pts)
    xc = mesh.geometry.x[mesh.geometry.dofmap]  # (nc, 3, 3)
    xq = np.einsum("qv,cvk->cqk", gphi, xc)
    e1, e2 = xc[:, 1] - xc[:, 0], xc[:, 2] - xc[:, 0]
    det = np.abs(e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0])
    uq = np.einsum("qi,ci->cq", phi, uh[V.dofmap.list])
    ex = np.sin(np.pi * xq[:, :, 0]) * np.sin(np.pi * xq[:, :, 1])
    return float(np.sqrt(np.sum((uq - ex) ** 2 * wts[None, :] * det[:, None])))


def _raw_constraint(V, indicator, relation, bcs):
    import dolfinx_mpc_amd as dm

    captured = {}

    class _Probe(dm.MultiPointConstraint):
        def add_constraint(self, V, slaves, masters, coeffs, owners, offsets):
            captured["raw"] = (slaves, masters, coeffs, owners, offsets)

    _Probe(V).create_periodic_constraint_geometrical(V, indicator, relation, bcs)
    return captured["raw"]


def _oracle_newton(oracle, V, bc, raw, u, F, J, iterates=None):
    """Newton's method with the oracle in the reference's call order; returns the number of iterations"""
    om = oracle.OracleMPC.from_raw(V, *raw)
    assert om.slaves.size > 0
    u.interpolate(lambda x: x[0] ** 2 * x[1] ** 2)  # test_nonlinear_assembly.py:98
    x = u.x.array.copy()
    f0 = None
    for it in range(25):
        uu = x.copy()
        oracle.homogenize(om, uu)
        oracle.backsubstitution(om, uu)
        u.x.array[:] = uu
        b = oracle.assemble_vector(F, om)
        oracle.apply_lifting(b, [J], [[bc]], om, x0=[x], scale=-1.0)
        dofs = bc.dof_indices()[0]
        b[dofs] = -1.0 * (bc.values_at_dofs() - x[dofs])  # set_bc(F, bcs, x0=x, alpha=-1)
        fn = np.linalg.norm(b)
        if iterates is not None:
            iterates.append(uu.copy())
        f0 = fn if f0 is None else f0
        if fn < 1e-13 or (it > 0 and fn <= 1e-12 * f0):
            break
        A = oracle.assemble_matrix(J, om, bcs=[bc])
        x -= spla.splu(A.tocsc()).solve(b)
    uu = x.copy()
    oracle.homogenize(om, uu)
    oracle.backsubstitution(om, uu)
    u.x.array[:] = uu
    return it


@pytest.mark.parametrize("p", [1, 2, 3])
def test_oracle_nonlinear_poisson_rates(oracle, p):
    Ns = np.array([4, 8, 10])
    err = []
    for N in Ns:
        mesh, V, bc, ind, rel, u, F, J = _setup(int(N), p)
        raw = _raw_constraint(V, ind, rel, [bc])
        assert raw[0].size > 0 and raw[1].size >= raw[0].size  # slaves and masters exist (:84-88)
        its = _oracle_newton(oracle, V, bc, raw, u, F, J)
        assert its < 12, its
        err.append(_l2_error(V, u.x.array))
    h = 1.0 / Ns
    rates = np.log(np.array(err[:-1]) / np.array(err[1:])) / np.log(h[:-1] / h[1:])
    assert np.all(rates > p + 0.9), (rates, err)  # test_nonlinear_assembly.py:112-114


@pytest.mark.gpu
@pytest.mark.parametrize("linear", ["lu", "bicgstab"])
@pytest.mark.parametrize("p", [1, 2, 3])
def test_gpu_nonlinear_problem(oracle, p, linear):
    """``NonlinearProblem.solve`` on the HIP path: rates as above, and the converged solution equals the oracle's Newton
    limit to 1e-9 (both solve F = 0 to 1e-12 of the first residual)"""
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd.problem import NonlinearProblem

    Ns = np.array([4, 8, 10])
    err = []
    for N in Ns:
        mesh, V, bc, ind, rel, u, F, J = _setup(int(N), p)
        mpc = dm.MultiPointConstraint(V)
        mpc.create_periodic_constraint_geometrical(V, ind, rel, [bc])
        mpc.finalize()
        assert mpc.slaves.size > 0 and mpc.masters.array.size >= mpc.slaves.size
        opts = {"snes_rtol": 1e-12, "snes_atol": 1e-13, "snes_stol": 0.0, "snes_max_it": 25}
        opts.update({"ksp_type": "preonly", "pc_type": "lu"} if linear == "lu" else {"ksp_type": "bicgstab", "ksp_rtol": 1e-13})
        problem = NonlinearProblem(F, u, mpc, [bc], J=J, solver_options=opts)
        u.interpolate(lambda x: x[0] ** 2 * x[1] ** 2)  # a non-zero initial guess (:97-98)
        uh, reason, its = problem.solve()
        assert reason > 0 and its < 12, (reason, its, problem.info)
        got = uh.x.array.copy()
        err.append(_l2_error(V, got))
        if N == 8:
            raw = _raw_constraint(V, ind, rel, [bc])
            _oracle_newton(oracle, V, bc, raw, u, F, J)
            assert abs(got - u.x.array).max() < 1e-9
    h = 1.0 / Ns
    rates = np.log(np.array(err[:-1]) / np.array(err[1:])) / np.log(h[:-1] / h[1:])
    assert np.all(rates > p + 0.9), (rates, err)


@pytest.mark.gpu
@pytest.mark.parametrize("tensor_order", [0, 1, 2])
@pytest.mark.parametrize("poly_order", [1, 2, 3])
def test_gpu_homogenize(tensor_order, poly_order):
    """python/tests/test_nonlinear_assembly.py:117-166: u = 1 everywhere, ``mpc.homogenize(u)`` zeroes exactly the slaves
    (scalar, vector and tensor valued P1-P3 spaces; the device kernel through the Function call shape)"""
    import dolfinx_mpc_amd as dm

    mesh = create_unit_square(8, 8)
    shape = [(), (2,), (2, 2)][tensor_order]
    V = fem.functionspace(mesh, ("Lagrange", poly_order, shape)) if shape else fem.functionspace(mesh, ("Lagrange", poly_order))

    def rel(x):
        out = np.zeros(x.shape)
        out[0], out[1], out[2] = 1.0 - x[0], x[1], x[2]
        return out

    mpc = dm.MultiPointConstraint(V)
    mpc.create_periodic_constraint_geometrical(V, lambda x: np.isclose(x[0], 0.0), rel, [])
    mpc.finalize()
    assert mpc.slaves.size > 0
    u = fem.Function(V)
    u.x.array[:] = 1.0
    mpc.homogenize(u)
    expect = np.ones(V.num_dofs)
    expect[mpc.slaves] = 0.0
    assert np.array_equal(u.x.array, expect)
    mpc.backsubstitution(u)  # every master is 1
    assert np.array_equal(u.x.array, np.ones(V.num_dofs))
