"""Known-answer tests for the oracle's element kernels.

The reference's element tensors come from FFCx-generated code (absent here), so
absolute values are "parity unpinned" against the reference; they are pinned
against closed forms instead: textbook P1 matrices, exact
mass matrices, row sums, patch tests, rigid-body modes, facet measures, and the
generic engine vs the FFCx-like fast paths used by the CPU baseline.  CPU only.
"""

import math

import numpy as np
import pytest

from dolfinx_mpc_amd import fem
from dolfinx_mpc_amd.fem import KernelSpec
from dolfinx_mpc_amd.quadrature import make_quadrature

TET = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=float)
TRI = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=float)


def _affine(ref, seed):
    rng = np.random.default_rng(seed)
    F = np.eye(3) + 0.3 * rng.standard_normal((3, 3))
    if ref.shape[0] == 3:  # keep triangles in the z=0 plane
        F[2, :] = 0
        F[:, 2] = 0
        F[2, 2] = 1
    return ref @ F.T + rng.standard_normal(3) * np.array([1, 1, 0 if ref.shape[0] == 3 else 1]), F


def _spec(form, cell, degree, bs=1, qdeg=2, fn=0, cd=0, fq=0):
    name = {1: "triangle", 2: "tetrahedron"}[cell]
    q, w = make_quadrature(name, qdeg)
    fname = {1: "interval", 2: "triangle"}[cell]
    fqp, fqw = make_quadrature(fname, fq)
    return KernelSpec(form, cell, degree, bs, fn, cd, q, w, fqp, fqw)


def test_quadrature_exactness():
    def exact_tet(a, b, c):
        return math.factorial(a) * math.factorial(b) * math.factorial(c) / math.factorial(a + b + c + 3)

    def exact_tri(a, b):
        return math.factorial(a) * math.factorial(b) / math.factorial(a + b + 2)

    for deg in range(0, 8):
        p, w = make_quadrature("tetrahedron", deg)
        for a in range(deg + 1):
            for b in range(deg + 1 - a):
                for c in range(deg + 1 - a - b):
                    assert abs((w * p[:, 0] ** a * p[:, 1] ** b * p[:, 2] ** c).sum() - exact_tet(a, b, c)) < 1e-15
        p, w = make_quadrature("triangle", deg)
        for a in range(deg + 1):
            for b in range(deg + 1 - a):
                assert abs((w * p[:, 0] ** a * p[:, 1] ** b).sum() - exact_tri(a, b)) < 2e-15


def test_p1_tet_laplace_textbook(oracle):
    A = oracle.tabulate_one(_spec(fem.FORM_STIFFNESS, 2, 1, qdeg=0), TET)
    ref = np.array([[3, -1, -1, -1], [-1, 1, 0, 0], [-1, 0, 1, 0], [-1, 0, 0, 1]]) / 6.0
    assert np.allclose(A, ref, atol=1e-15)
    Af = oracle.tabulate_one(_spec(fem.FORM_STIFFNESS, 2, 1, qdeg=0), TET, which=1)
    assert np.allclose(Af, ref, atol=1e-15)


def test_p1_tri_laplace_textbook(oracle):
    A = oracle.tabulate_one(_spec(fem.FORM_STIFFNESS, 1, 1, qdeg=0), TRI)
    ref = np.array([[2, -1, -1], [-1, 1, 0], [-1, 0, 1]]) / 2.0
    assert np.allclose(A, ref, atol=1e-15)


@pytest.mark.parametrize("cell,degree", [(1, 1), (1, 2), (2, 1), (2, 2)])
def test_stiffness_symmetric_rows_sum_to_zero_and_formula(oracle, cell, degree):
    ref = TRI if cell == 1 else TET
    X, F = _affine(ref, 3 + cell + degree)
    A = oracle.tabulate_one(_spec(fem.FORM_STIFFNESS, cell, degree, qdeg=2 * (degree - 1)), X)
    assert np.allclose(A, A.T, atol=1e-13)
    assert np.allclose(A.sum(axis=1), 0, atol=1e-12)
    if degree == 1:
        # A = |det J| / d! * G G^T with G = grad(phi) = [-1..; I] J^-1
        tdim = 2 if cell == 1 else 3
        J = F[:tdim, :tdim]
        G = np.vstack([-np.ones((1, tdim)), np.eye(tdim)]) @ np.linalg.inv(J)
        assert np.allclose(A, abs(np.linalg.det(J)) / math.factorial(tdim) * G @ G.T, atol=1e-12)
    # energy of a linear field u = a.x equals |T| |a|^2
    tdim = 2 if cell == 1 else 3
    a = np.array([0.3, -1.2, 0.7])[:tdim]
    nodes = X if degree == 1 else None
    if degree == 1:
        u = nodes[:, :tdim] @ a
        vol = abs(np.linalg.det(F[:tdim, :tdim])) / math.factorial(tdim)
        assert u @ A @ u == pytest.approx(vol * a @ a, rel=1e-12)


@pytest.mark.parametrize("cell", [1, 2])
def test_p1_mass_matrix_closed_form(oracle, cell):
    ref = TRI if cell == 1 else TET
    X, F = _affine(ref, 11 + cell)
    tdim = 2 if cell == 1 else 3
    vol = abs(np.linalg.det(F[:tdim, :tdim])) / math.factorial(tdim)
    M = oracle.tabulate_one(_spec(fem.FORM_MASS, cell, 1, qdeg=2), X)
    n = tdim + 1
    # |T| / ((d+1)(d+2)) (1 + delta_ij)
    expect = vol / ((tdim + 1) * (tdim + 2)) * (np.ones((n, n)) + np.eye(n))
    assert np.allclose(M, expect, atol=1e-14)
    M2 = oracle.tabulate_one(_spec(fem.FORM_MASS, cell, 2, qdeg=4), X)
    assert M2.sum() == pytest.approx(vol, rel=1e-12)
    assert np.allclose(M2, M2.T)


@pytest.mark.parametrize("cell,degree", [(1, 1), (1, 2), (2, 1), (2, 2)])
def test_source_integrates_to_volume_and_poly(oracle, cell, degree):
    ref = TRI if cell == 1 else TET
    X, F = _affine(ref, 21 + cell)
    tdim = 2 if cell == 1 else 3
    vol = abs(np.linalg.det(F[:tdim, :tdim])) / math.factorial(tdim)
    b = oracle.tabulate_one(_spec(fem.FORM_SOURCE, cell, degree, qdeg=degree, fn=fem.FN_ONE), X)
    assert b.sum() == pytest.approx(vol, rel=1e-13)
    # polynomial f of degree 3: two rules that are both exact must agree
    b1 = oracle.tabulate_one(_spec(fem.FORM_SOURCE, cell, degree, qdeg=degree + 3, fn=fem.FN_POLY3), X)
    b2 = oracle.tabulate_one(_spec(fem.FORM_SOURCE, cell, degree, qdeg=degree + 5, fn=fem.FN_POLY3), X)
    assert np.allclose(b1, b2, rtol=1e-12, atol=1e-14)


def test_fast_paths_equal_generic(oracle):
    X, _ = _affine(TET, 5)
    k = _spec(fem.FORM_STIFFNESS, 2, 1, qdeg=0)
    assert np.allclose(oracle.tabulate_one(k, X, which=0), oracle.tabulate_one(k, X, which=1), rtol=1e-13, atol=1e-15)
    k = _spec(fem.FORM_SOURCE, 2, 1, qdeg=5, fn=fem.FN_BENCH_PERIODIC)
    g, f = oracle.tabulate_one(k, X, which=0), oracle.tabulate_one(k, X, which=2)
    assert np.allclose(g, f, rtol=1e-13, atol=1e-16)


def test_coefficient_and_constant_weighting(oracle):
    """w = packed coefficient dofs, c = constants (python/tests/test_mpc_pipeline.py:45-46)."""
    X, _ = _affine(TET, 9)
    k0 = _spec(fem.FORM_STIFFNESS, 2, 1, qdeg=0)
    k1 = _spec(fem.FORM_STIFFNESS, 2, 1, qdeg=1, cd=1)
    A0 = oracle.tabulate_one(k0, X)
    A1 = oracle.tabulate_one(k1, X, w=np.full(4, 2.5), c=np.array([1.5]))
    assert np.allclose(A1, 2.5 * 1.5 * A0, rtol=1e-13)
    # linear coefficient: mean value times A0 (gradients constant)
    w = np.array([1.0, 2.0, -1.0, 0.5])
    A2 = oracle.tabulate_one(k1, X, w=w)
    assert np.allclose(A2, w.mean() * A0, rtol=1e-13)


@pytest.mark.parametrize("cell,bs", [(1, 2), (2, 3)])
def test_elasticity_rigid_body_modes(oracle, cell, bs):
    ref = TRI if cell == 1 else TET
    X, _ = _affine(ref, 31)
    k = _spec(fem.FORM_ELASTICITY, cell, 1, bs=bs, qdeg=0)
    A = oracle.tabulate_one(k, X, c=np.array([0.7, 1.3]))
    assert np.allclose(A, A.T, atol=1e-12)
    n = X.shape[0]
    modes = []
    for a in range(bs):  # translations
        u = np.zeros((n, bs))
        u[:, a] = 1
        modes.append(u.ravel())
    if bs == 2:
        modes.append(np.stack([-X[:, 1], X[:, 0]], axis=1).ravel())
    else:
        for (i, j) in ((0, 1), (0, 2), (1, 2)):
            u = np.zeros((n, 3))
            u[:, i], u[:, j] = -X[:, j], X[:, i]
            modes.append(u.ravel())
    for m in modes:
        assert np.allclose(A @ m, 0, atol=1e-12)
    ev = np.linalg.eigvalsh(A)
    assert (abs(ev) < 1e-10).sum() == len(modes)


@pytest.mark.parametrize("cell", [1, 2])
def test_facet_kernels_measure(oracle, cell):
    ref = TRI if cell == 1 else TET
    X, _ = _affine(ref, 41)
    nf = 3 if cell == 1 else 4
    for lf in range(nf):
        verts = [v for v in range(nf) if v != lf]
        P = X[verts]
        if cell == 1:
            meas = np.linalg.norm(P[1] - P[0])
        else:
            meas = 0.5 * np.linalg.norm(np.cross(P[1] - P[0], P[2] - P[0]))
        k = _spec(fem.FORM_FACET_MASS, cell, 1, fq=2)
        M = oracle.tabulate_one(k, X, local_facet=lf)
        assert M.sum() == pytest.approx(meas, rel=1e-13)
        assert np.allclose(M[lf], 0) and np.allclose(M[:, lf], 0)  # the opposite vertex does not see the facet
        k = _spec(fem.FORM_FACET_SOURCE, cell, 2, fq=2, fn=fem.FN_ONE)
        b = oracle.tabulate_one(k, X, local_facet=lf)
        assert b.sum() == pytest.approx(meas, rel=1e-13)


def test_global_patch_tests(oracle):
    """A.1 = 0 before bcs; sum(b) = integral of f (SURVEY.md Appendix D known answers)."""
    from dolfinx_mpc_amd.mesh import create_unit_cube

    mesh = create_unit_cube(3, 4, 2)
    assert mesh.geometry.x[mesh.geometry.dofmap].shape == (3 * 4 * 2 * 6, 4, 3)
    for degree in (1, 2):
        V = fem.functionspace(mesh, ("Lagrange", degree))
        emp = oracle.OracleMPC.empty(V)
        A = oracle.assemble_matrix(fem.form_stiffness(V), emp)
        assert abs(A @ np.ones(V.num_dofs)).max() < 1e-12
        b = oracle.assemble_vector(fem.form_source(V, fem.FN_ONE), emp)
        assert b.sum() == pytest.approx(1.0, rel=1e-13)
        # quadratic u: a(u, 1) = 0 and a(u,u) = int |grad u|^2 for u = x^2 (P2 exact): 4/3
        if degree == 2:
            x = V.tabulate_dof_coordinates()
            u = x[:, 0] ** 2
            assert u @ (A @ u) == pytest.approx(4.0 / 3.0, rel=1e-12)
    # the Kuhn mesh has nnz = Nv + 2E with E = 3n(n+1)^2 + 3n^2(n+1) + n^3
    n = 5
    mesh = create_unit_cube(n, n, n)
    V = fem.functionspace(mesh, ("Lagrange", 1))
    emp = oracle.OracleMPC.empty(V)
    rowptr, cols = oracle.create_pattern(fem.form_stiffness(V), emp, emp)
    E = 3 * n * (n + 1) ** 2 + 3 * n * n * (n + 1) + n**3
    assert cols.size == (n + 1) ** 3 + 2 * E
