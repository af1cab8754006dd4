"""Cell types and degrees the reference's assembly tests sweep (python/tests/test_matrix_assembly.py:23-26, 61-64,
test_vector_assembly.py:22-24: degree 1-3 on triangles AND quadrilaterals; test_stokes_channelflow.py:21-22: Q2 on
hexahedra) beyond the built-in operators (P1 / P2 simplices, Q1 hexahedra): dolfinx_mpc_amd/elements.py defines the
elements and their dof numbering, codegen.generate_general writes their kernels as UFCx C text, which the product
compiles with hipRTC and the oracle with gcc.

CPU: the elements themselves (nodal basis, shared-edge dofs), ANALYTIC integrals that no shared table can fake (exact
energies / moments of polynomials on the unit square and cube), the reference's K^T A K identity through the oracle.
GPU (-m gpu): both algorithms and every dispatch entry against the oracle."""

import numpy as np
import pytest

from dolfinx_mpc_amd import elements as el
from dolfinx_mpc_amd import fem
from dolfinx_mpc_amd.mesh import create_unit_cube, create_unit_square
from problems import element_sweep_cases, oracle_mpc, oracle_outputs, product_outputs

CASES = element_sweep_cases()
IDS = [f"el{i}" for i in range(len(CASES))]
ELEMENTS = [("triangle", 1), ("triangle", 2), ("triangle", 3), ("quadrilateral", 1), ("quadrilateral", 2), ("quadrilateral", 3),
            ("tetrahedron", 1), ("tetrahedron", 2), ("tetrahedron", 3), ("hexahedron", 1), ("hexahedron", 2), ("hexahedron", 3),
            ("triangle", 4), ("quadrilateral", 4), ("tetrahedron", 4), ("hexahedron", 4)]


@pytest.mark.parametrize("cell,degree", ELEMENTS)
def test_nodal_basis_and_polynomial_reproduction(cell, degree):
    pts, ent, _ = el.reference_nodes(cell, degree)
    phi, _ = el.tabulate(cell, degree, pts)
    assert np.allclose(phi, np.eye(pts.shape[0]), atol=1e-11)  # phi_j(node_i) = delta_ij
    rng = np.random.default_rng(1)
    x = rng.random((7, el.tdim(cell))) / el.tdim(cell)
    ph, dp = el.tabulate(cell, degree, x)
    assert np.allclose(ph.sum(axis=1), 1.0) and np.allclose(dp.sum(axis=2), 0.0, atol=1e-10)
    # the space holds every monomial of total degree <= p: interpolation reproduces it and its gradient
    for ex in ([degree] + [0] * (el.tdim(cell) - 1), [1] * min(degree, el.tdim(cell)) + [0] * max(el.tdim(cell) - degree, 0)):
        if sum(ex) > degree and el.is_simplex(cell):
            continue
        f = lambda y: np.prod(y ** np.array(ex), axis=1)  # noqa: E731
        assert np.allclose(ph @ f(pts), f(x), atol=1e-11)
    # the order agrees with the built-in operators' for degree <= 2 on simplices
    if el.is_simplex(cell) and degree <= 2:
        from dolfinx_mpc_amd.quadrature import lagrange_basis

        assert np.allclose(ph, lagrange_basis(cell, degree, x), atol=1e-12)


def _hex_symmetry(perm, flips):
    """local vertex order of a hexahedron after a symmetry of the reference cube: new vertex with bits (b0, b1, b2) is the
    old vertex whose bit perm[k] is b_k ^ flips[k]"""
    out = []
    for v in range(8):
        b = [(v >> k) & 1 for k in range(3)]
        old = 0
        for k in range(3):
            old |= (b[k] ^ flips[k]) << perm[k]
        out.append(old)
    return out


def _scrambled(mesh, cell):
    """the same mesh with the local vertex order of its cells permuted cell by cell (all orientations a mesh file may hold)"""
    from dolfinx_mpc_amd.mesh import Mesh

    c = mesh.geometry.dofmap.copy()
    if cell == "triangle":
        c[1::2] = c[1::2][:, [1, 2, 0]]
        c[::3] = c[::3][:, [0, 2, 1]]
    elif cell == "tetrahedron":
        c[1::2] = c[1::2][:, [1, 2, 0, 3]]
        c[::3] = c[::3][:, [0, 3, 2, 1]]
        c[2::5] = c[2::5][:, [3, 0, 1, 2]]
    elif cell == "hexahedron":
        syms = [((1, 0, 2), (0, 0, 0)), ((2, 0, 1), (1, 0, 0)), ((0, 2, 1), (0, 1, 1)), ((1, 2, 0), (1, 1, 1)), ((0, 1, 2), (0, 0, 1)),
                ((2, 1, 0), (0, 1, 0)), ((0, 1, 2), (1, 1, 0))]
        for k, (perm, flips) in enumerate(syms):
            c[k + 1::len(syms) + 1] = c[k + 1::len(syms) + 1][:, _hex_symmetry(perm, flips)]
    else:
        return mesh
    return Mesh(mesh.geometry.x, c, cell)


@pytest.mark.parametrize("cell,degree", [("triangle", 3), ("quadrilateral", 2), ("quadrilateral", 3), ("hexahedron", 2), ("hexahedron", 3),
                                         ("tetrahedron", 3)])
def test_shared_dofs_agree_between_cells(cell, degree):
    """a dof on a shared edge / face is ONE dof: the coordinates the two cells assign to it agree, whatever the cells'
    local orientations (shuffled local vertex order on simplices AND hexahedra: all relative orientations of a shared
    quadrilateral face occur)"""
    mesh = create_unit_square(4, 3, cell) if el.tdim(cell) == 2 else create_unit_cube(2, 3, 2, cell)
    mesh = _scrambled(mesh, cell)
    V = fem.functionspace(mesh, ("Lagrange", degree))
    x = V.tabulate_dof_coordinates()
    gphi, _ = el.tabulate(cell, 1, el.reference_nodes(cell, degree)[0])
    xc = np.einsum("dv,cvk->cdk", gphi, mesh.geometry.x[mesh.geometry.dofmap])
    assert np.allclose(x[V.dofmap.list], xc, atol=1e-13)
    assert np.unique(np.round(x, 10), axis=0).shape[0] == V.num_dofs  # no two dofs at one point
    n1 = {2: (4 * degree + 1) * (3 * degree + 1), 3: (2 * degree + 1) * (3 * degree + 1) * (2 * degree + 1)}[el.tdim(cell)]
    assert V.num_dofs == n1


def _unconstrained(V, a=None, L=None):
    from problems import Case, empty_raw

    return Case("plain", V, a, L, [], empty_raw())


@pytest.mark.parametrize("cell,degree", [("triangle", 3), ("quadrilateral", 1), ("quadrilateral", 2), ("quadrilateral", 3), ("hexahedron", 2),
                                         ("hexahedron", 3), ("tetrahedron", 3)])
def test_analytic_integrals(oracle, cell, degree):
    """u^T A v = int grad u . grad v, u^T M v = int u v, u^T b = int f u for polynomials IN the space on the unit square /
    cube with u_i = u(x_i) at the space's own dof coordinates (also on a sheared mesh for the affine invariance)"""
    two = el.tdim(cell) == 2
    mesh = create_unit_square(3, 2, cell) if two else _scrambled(create_unit_cube(2, 2, 1, cell), cell)
    V = fem.functionspace(mesh, ("Lagrange", degree))
    x = V.tabulate_dof_coordinates()
    A = oracle_outputs(oracle, _unconstrained(V, fem.form_stiffness(V)))["A"]
    M = oracle_outputs(oracle, _unconstrained(V, fem.form_mass(V)))["A"]
    b = oracle_outputs(oracle, _unconstrained(V, None, fem.form_source(V, fem.FN_LINEAR)))["b"]
    p = degree
    if two:
        u = x[:, 0] ** p + 2.0 * x[:, 1]                      # grad u = (p x^(p-1), 2)
        v = x[:, 1] ** p - x[:, 0]                            # grad v = (-1, p y^(p-1))
        # int_0^1 int_0^1 (-p x^(p-1) + 2 p y^(p-1)) = -1 + 2 = 1
        assert abs(u @ (A @ v) - 1.0) < 1e-12
        # int u v = int (x^p + 2y)(y^p - x) = 1/((p+1)^2) - 1/(p+2) + 2/(p+2) - 1/2
        assert abs(u @ (M @ v) - (1.0 / (p + 1) ** 2 - 1.0 / (p + 2) + 2.0 / (p + 2) - 0.5)) < 1e-12
        # f = 1 + x - 2 y (FN_LINEAR, component 0): int f x^p... with u: int (1 + x - 2y)(x^p + 2y)
        want = (1.0 / (p + 1) + 1.0 / (p + 2) - 1.0 / (p + 1)) + (1.0 + 0.5 - 4.0 / 3.0)
        assert abs(u @ b - want) < 1e-12
    else:
        u = x[:, 0] ** p + x[:, 1] * x[:, 2]
        v = x[:, 2] ** p + x[:, 0]
        # grad u . grad v = p x^(p-1) + y p z^(p-1) -> 1 + 1/2
        assert abs(u @ (A @ v) - 1.5) < 1e-12
        assert abs(M.sum() - 1.0) < 1e-13
    assert abs(A @ np.ones(V.num_dofs)).max() < 1e-12  # constants in the kernel of the stiffness matrix


@pytest.mark.parametrize("make", CASES, ids=IDS)
def test_oracle_identities(oracle, make):
    """A_mpc[free, free] == K^T A K, b_mpc[free] == K^T b (python/src/dolfinx_mpc/utils/test.py:202-265)"""
    case = make()
    mpc = oracle_mpc(oracle, case)
    emp = oracle.OracleMPC.empty(case.V)
    out = oracle_outputs(oracle, case)
    A_org = oracle.assemble_matrix(case.a, emp, bcs=case.bcs, diagval=case.diagval)
    oracle.compare_mpc_lhs(A_org, out["A"], mpc, atol=5e3 * np.finfo(np.float64).resolution * max(1.0, abs(A_org).max()))
    oracle.compare_mpc_rhs(oracle.assemble_vector(case.L, emp), out["b"], mpc)


RTOL = 1e-12


def _check(case, ref, out, what):
    if "A" in ref:
        assert np.array_equal(out["A"].indptr, ref["A"].indptr) and np.array_equal(out["A"].indices, ref["A"].indices)
        assert abs(out["A"].data - ref["A"].data).max() <= RTOL * max(1.0, abs(ref["A"].data).max()), f"{case.name} A [{what}]"
    for k in ("b", "b_lifted"):
        if k in ref:
            assert abs(out[k] - ref[k]).max() <= RTOL * max(1.0, abs(ref[k]).max()), f"{case.name} {k} [{what}]"


@pytest.mark.gpu
@pytest.mark.parametrize("alg", ["atomic", "rowblock", None])
@pytest.mark.parametrize("make", CASES, ids=IDS)
def test_gpu_element_sweep(oracle, make, alg):
    from dolfinx_mpc_amd import _native

    case = make()
    if alg == "rowblock" and case.V.element_ndofs > 32:
        # Q3 hexahedra (64 nodes per cell): no row-block plan; asked for explicitly it says so, 'auto' (None) falls back
        with pytest.raises(_native.PlanNotRepresentable):
            product_outputs(case, algorithm=alg)
        return
    _check(case, oracle_outputs(oracle, case), product_outputs(case, algorithm=alg), alg)


@pytest.mark.gpu
@pytest.mark.parametrize("which,name", [("matrix", "ufcx_rowblock"), ("vector", "ufcx_ownblock"), ("vector", "ufcx_rowblock")])
@pytest.mark.parametrize("make", CASES, ids=IDS)
def test_gpu_element_sweep_dispatch_entries(oracle, make, which, name, monkeypatch):
    monkeypatch.setenv("MPCX_FORCE_KERNEL", f"{which}={name}")
    case = make()
    _check(case, oracle_outputs(oracle, case), product_outputs(case, algorithm=None), f"{which}={name}")
