"""UFCx import: element kernels given as C source with the UFCx signature -- the reference's own
seam, cpp/assemble_matrix.cpp:438-439 -- instead of the built-in operator ids.  The product compiles the text with
hipRTC into a __device__ function (include/mpcx.h mpcx_ufcx_compile); the oracle compiles THE SAME text with gcc
and calls it through the function pointer, exactly as the reference would.

CPU: the imported kernels are right (against the built-in operators and analytic integrals), and the hipRTC
compilation itself works without a device.  GPU (-m gpu): product == oracle for matrix, vector and lifting with
constraints and Dirichlet conditions, on cells and on exterior facets, square and rectangular."""

import os

import numpy as np
import pytest

from dolfinx_mpc_amd import fem
from dolfinx_mpc_amd.mesh import create_unit_cube
from problems import Case, _walls_yz, oracle_mpc, oracle_outputs, periodic_raw, product_outputs, stokes_slip_problem
from test_independent_checks import gauss_cube, u_fun, v_fun

HERE = os.path.dirname(os.path.abspath(__file__))


def _src(name):
    return open(os.path.join(HERE, "ufcx", name + ".c")).read()


def _laplace_case(N=4, reorder=None):
    mesh = create_unit_cube(N, N, N, reorder=reorder)
    V = fem.functionspace(mesh, ("Lagrange", 1))
    g = fem.Function(V)
    g.interpolate(lambda x: 0.3 + x[1] - 2.0 * x[2])
    bc = fem.dirichletbc(g, fem.locate_dofs_geometrical(V, _walls_yz), V)
    a = fem.form_ufcx([V, V], _src("laplace_p1_tet"), "tabulate_tensor_laplace_p1_tet")
    wh = fem.Function(V)
    wh.interpolate(lambda x: 1.0 + 0.5 * x[0] + x[2])
    L = fem.form_ufcx([V], _src("source_p1_tet"), "tabulate_tensor_source_p1_tet", coefficient=wh, constant=fem.Constant(0.7))
    return Case("ufcx_laplace", V, a, L, [bc], periodic_raw(V, [bc])), wh


def test_imported_laplace_equals_builtin_operator(oracle):
    case, _ = _laplace_case()
    mpc = oracle_mpc(oracle, case)
    A_imp = oracle.assemble_matrix(case.a, mpc, bcs=case.bcs)
    A_ref = oracle.assemble_matrix(fem.form_stiffness(case.V), mpc, bcs=case.bcs)
    assert np.array_equal(A_imp.indptr, A_ref.indptr) and np.array_equal(A_imp.indices, A_ref.indices)
    assert abs(A_imp - A_ref).max() <= 1e-13 * abs(A_ref).max()


def test_imported_source_reproduces_analytic_integral(oracle):
    case, wh = _laplace_case()
    b = oracle.assemble_vector(case.L, oracle.OracleMPC.empty(case.V))
    X = case.V.tabulate_dof_coordinates().T
    u, _ = u_fun(1)
    exact = gauss_cube(lambda x: 0.7 * (1.0 + 0.5 * x[0] + x[2]) * (1.0 + 2.0 * x[0] - x[1] * x[2]) * u(x))
    assert u(X) @ b == pytest.approx(exact, rel=1e-12)


def _slip_forms(n=2):
    V, Q, bcs, raw_v, forms, L0 = stokes_slip_problem(3, n)
    wall = V.mesh.locate_exterior_facets(lambda x: np.isclose(x[1], 1.0) | np.isclose(x[0], 1.0) | np.isclose(x[2], 0.0))
    a01f = fem.form_ufcx([V, Q], _src("slip_facet_p2p1_tet"), "tabulate_tensor_slip_facet_p2p1_tet", "exterior_facet", wall)
    return V, Q, bcs, raw_v, a01f, wall


def test_imported_slip_facet_term_reproduces_analytic_integral(oracle):
    """v^T A01 p = int_Gamma p (n . v) ds over the faces x = 1 (n = e_x), y = 1 (n = e_y), z = 0 (n = -e_z)"""
    V, Q, bcs, raw_v, a01f, wall = _slip_forms(2)
    A = oracle.assemble_matrix(a01f, oracle.OracleMPC.empty(V), oracle.OracleMPC.empty(Q))
    (u, _), (v, _) = u_fun(2), v_fun(2)
    X = V.tabulate_dof_coordinates().T
    Vh = np.stack([u(X), v(X), u(X) - 2 * v(X)], axis=1).reshape(-1)
    p = lambda x: 0.2 + x[0] - x[1] + 3.0 * x[2]
    Ph = p(Q.tabulate_dof_coordinates().T)
    fx = lambda s: np.stack([np.ones_like(s[0]), s[0], s[1]])
    fy = lambda s: np.stack([s[0], np.ones_like(s[0]), s[1]])
    fz = lambda s: np.stack([s[0], s[1], np.zeros_like(s[0])])
    exact = (gauss_cube(lambda s: p(fx(s)) * u(fx(s)), dim=2) + gauss_cube(lambda s: p(fy(s)) * v(fy(s)), dim=2)
             - gauss_cube(lambda s: p(fz(s)) * (u(fz(s)) - 2 * v(fz(s))), dim=2))
    assert Vh @ (A @ Ph) == pytest.approx(exact, rel=1e-12)


def test_hiprtc_compiles_the_imported_kernels_without_a_device():
    """mpcx_ufcx_compile cross-compiles for gfx950 (no GPU needed): handle, non-empty code object, and a
    readable compiler log on broken source"""
    from dolfinx_mpc_amd import _native

    L = _native.lib()
    for name, fn, rank, shape in (("laplace_p1_tet", "tabulate_tensor_laplace_p1_tet", 2, (4, 1, 4, 1)),
                                  ("source_p1_tet", "tabulate_tensor_source_p1_tet", 1, (4, 1, 0, 0)),
                                  ("slip_facet_p2p1_tet", "tabulate_tensor_slip_facet_p2p1_tet", 2, (10, 3, 4, 1))):
        d = _native.UfcxDescT(_src(name).encode(), fn.encode(), rank, *shape, 4)
        h = L.mpcx_ufcx_compile(d)
        assert h, L.mpcx_last_error().decode()
        assert L.mpcx_ufcx_code_size(h) > 1000
        L.mpcx_ufcx_free(h)
    bad = _native.UfcxDescT(b"void broken(double* A) { A[0] = undefined_symbol; }", b"broken", 1, 4, 1, 0, 0, 4)
    assert not L.mpcx_ufcx_compile(bad)
    assert "undefined_symbol" in L.mpcx_last_error().decode()


# ------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("alg", ["atomic", "rowblock"])
@pytest.mark.parametrize("reorder", [None, (2, 2, 2)])
def test_gpu_imported_laplace_and_source_match_oracle(oracle, reorder, alg):
    case, _ = _laplace_case(5, reorder)
    ref = oracle_outputs(oracle, case)
    out = product_outputs(case, algorithm=alg)
    assert np.array_equal(out["A"].indptr, ref["A"].indptr) and np.array_equal(out["A"].indices, ref["A"].indices)
    assert abs(out["A"].data - ref["A"].data).max() <= 1e-12 * max(1.0, abs(ref["A"]).max())
    for k in ("b", "b_lifted"):
        assert abs(out[k] - ref[k]).max() <= 1e-12 * max(1.0, abs(ref[k]).max()), k


@pytest.mark.gpu
@pytest.mark.parametrize("alg", ["atomic", "rowblock"])
def test_gpu_imported_slip_facet_block_matches_oracle(oracle, alg):
    """rectangular block (P2^3 x P1) on exterior facets with the slip constraint on the rows and Dirichlet rows"""
    import dolfinx_mpc_amd as dm
    from problems import empty_raw

    V, Q, bcs, raw_v, a01f, wall = _slip_forms(2)
    mv, mq = oracle.OracleMPC.from_raw(V, *raw_v), oracle.OracleMPC.from_raw(Q, *empty_raw())
    ref = oracle.assemble_matrix(a01f, mv, mq, bcs=bcs)
    pv = dm.MultiPointConstraint(V)
    pv.add_constraint(V, *raw_v)
    pv.finalize()
    pq = dm.MultiPointConstraint(Q)
    pq.finalize()
    A = dm.assemble_matrix(a01f, (pv, pq), bcs=bcs, algorithm=alg).to_scipy()
    assert np.array_equal(A.indptr, ref.indptr) and np.array_equal(A.indices, ref.indices)
    assert abs(A.data - ref.data).max() <= 1e-12 * max(1.0, abs(ref).max())
    assert abs(ref).max() > 0.01
