"""Imported (UFCx) element kernels inside the CLUSTER kernels (ufcx_matrix_cube_*_kernel / ufcx_vector_cube_own_kernel,
csrc/mpcx_ufcx.cpp): the path BASELINE's north_star names -- FFCx-shaped ``tabulate_tensor`` text batched on the device
(cpp/assemble_matrix.cpp:438-439, 488-547) -- on the fast kernels, with nothing known about the text.

What is checked against the oracle (which calls the gcc-compiled text through the function pointer, cell by cell, with
the mesh's own vertex order):
  * the default dispatch takes the cluster entries for scalar P1 forms on tetrahedra and matches on matrix, vector, lifting;
  * a rule that is NOT symmetric under vertex permutations (so a call with permuted vertices is a different number):
    the clusters hand the function every cell exactly as the mesh lists it;
  * meshes whose cells list their vertices in another order: those clusters are left to the per-cell kernels
    (all cells permuted: none qualifies; half of them: both paths in one assembly);
  * unsymmetric element tensors (a convection-like term): no symmetry is assumed in the 46-pair accumulation;
  * coefficients and constants: the packed coefficients of the six cells of a cluster through ``cube_cells``;
  * constrained cells / Dirichlet rows and columns / wide records (master rows) through the same launches."""

import importlib

import numpy as np
import pytest

from problems import Case, _walls_yz, case_cube_periodic, oracle_outputs, periodic_raw, product_mpc, product_outputs

pytestmark = pytest.mark.gpu


def _unsymmetric_rule():
    """a positive 5-point rule on the reference tetrahedron without any vertex symmetry (weights sum to 1/6)"""
    pts = np.array([[0.11, 0.17, 0.23], [0.52, 0.13, 0.09], [0.08, 0.61, 0.14], [0.19, 0.07, 0.66], [0.31, 0.29, 0.27]])
    wts = np.array([0.21, 0.17, 0.26, 0.13, 0.23])
    return pts, wts / wts.sum() / 6.0


def _forms(V, rule=None, fexpr=None, coefficient=None, constant=None):
    from dolfinx_mpc_amd import fem
    from dolfinx_mpc_amd.codegen import BENCH_PERIODIC_F, generate
    from dolfinx_mpc_amd.quadrature import make_quadrature

    qa = rule or make_quadrature("tetrahedron", 0)
    ql = rule or make_quadrature("tetrahedron", 5)
    cd = 0 if coefficient is None else 1
    sa, na = generate("stiffness", "tetrahedron", 1, 1, qa, coefficient_degree=cd, use_constant=constant is not None)
    sl, nl = generate("source", "tetrahedron", 1, 1, ql, fexpr=fexpr or BENCH_PERIODIC_F, coefficient_degree=cd,
                      use_constant=constant is not None)
    a = fem.form_ufcx([V, V], sa, na, coefficient=coefficient, constant=constant)
    L = fem.form_ufcx([V], sl, nl, coefficient=coefficient, constant=constant)
    return a, L


def _taken(case):
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd.la import create_vector

    am = importlib.import_module("dolfinx_mpc_amd.assemble_matrix")
    av = importlib.import_module("dolfinx_mpc_amd.assemble_vector")
    mpc = product_mpc(case)
    A = dm.create_matrix(case.a, mpc)
    ma, _k = am.matrix_args(case.a, 0, A, mpc, mpc, case.bcs, 2)
    va, _k2 = av.vector_args(case.L, 0, create_vector(case.V), mpc, 0)
    return ma, va


def _check(oracle, case, what=""):
    ref = oracle_outputs(oracle, case)
    out = product_outputs(case)
    assert np.array_equal(out["A"].indptr, ref["A"].indptr) and np.array_equal(out["A"].indices, ref["A"].indices)
    assert abs(out["A"].data - ref["A"].data).max() <= 1e-12 * max(1.0, abs(ref["A"].data).max()), what + " A"
    for k in ("b", "b_lifted"):
        assert abs(out[k] - ref[k]).max() <= 1e-12 * max(1.0, abs(ref[k]).max()), f"{what} {k}"


def _periodic_case(mesh, name, **kw):
    from dolfinx_mpc_amd import fem

    V = fem.functionspace(mesh, ("Lagrange", 1))
    bc = fem.dirichletbc(0.3, fem.locate_dofs_geometrical(V, _walls_yz), V)
    coefficient = None
    if kw.pop("with_coefficient", False):
        coefficient = fem.Function(V)
        coefficient.interpolate(lambda x: 1.0 + 0.5 * x[0] + x[2] * x[1])
    a, L = _forms(V, coefficient=coefficient, **kw)
    return Case(name, V, a, L, [bc], periodic_raw(V, [bc]))


@pytest.mark.parametrize("n, reorder", [(6, (2, 2, 2)), (12, (4, 4, 4)), (5, None)])
def test_default_dispatch_takes_the_cluster_kernels(oracle, n, reorder):
    base = case_cube_periodic(n, 1, 0.3, reorder=reorder) if reorder else case_cube_periodic(n, 1, 0.3)
    case = _periodic_case(base.mesh, f"ufcx_clusters_{n}")
    ma, va = _taken(case)
    assert ma.kernel_name == "ufcx_cube" and ma.leftover is None
    assert va.kernel_name == "ufcx_cube_own" and va.leftover is None
    _check(oracle, case)


def test_rule_without_symmetry_sees_the_mesh_vertex_order(oracle):
    """with an unsymmetric rule and a non-polynomial integrand the value depends on the order the vertices are handed
    over: the oracle on a mesh with one cell's vertices permuted differs from the oracle on the original mesh by far
    more than the tolerance (so the test can see the difference), and the cluster kernels match the oracle"""
    from dolfinx_mpc_amd.mesh import Mesh

    base = case_cube_periodic(6, 1, 0.3, reorder=(2, 2, 2)).mesh
    rule = _unsymmetric_rule()
    case = _periodic_case(base, "unsym_rule", rule=rule)
    ma, va = _taken(case)
    assert ma.kernel_name == "ufcx_cube" and va.kernel_name == "ufcx_cube_own"
    _check(oracle, case, "unsymmetric rule")
    cells = base.geometry.dofmap.copy()
    cells[:, [1, 2]] = cells[:, [2, 1]]
    other = Mesh(base.geometry.x, cells, "tetrahedron")
    other.node_tile_offsets = base.node_tile_offsets
    b0 = oracle_outputs(oracle, case)["b"]
    b1 = oracle_outputs(oracle, _periodic_case(other, "unsym_rule_swapped", rule=rule))["b"]
    assert abs(b0 - b1).max() > 1e-6 * abs(b0).max()


@pytest.mark.parametrize("fraction", [1.0, 0.3])
def test_cells_listed_in_another_vertex_order_keep_the_per_cell_kernels(oracle, fraction):
    from dolfinx_mpc_amd.clusters import mesh_clusters_device, mesh_clusters_ordered_device
    from dolfinx_mpc_amd.mesh import Mesh

    base = case_cube_periodic(6, 1, 0.3, reorder=(2, 2, 2)).mesh
    rng = np.random.default_rng(5)
    cells = base.geometry.dofmap.copy()
    ncl = cells.shape[0] // 6
    for g in np.flatnonzero(rng.random(ncl) < fraction):  # (the generator lists the six cells of a cube consecutively)
        c = 6 * g + rng.integers(6)
        p = rng.permutation(4)
        while np.array_equal(p, np.arange(4)):
            p = rng.permutation(4)
        cells[c] = cells[c][p]
    mesh = Mesh(base.geometry.x, cells, "tetrahedron")
    mesh.node_tile_offsets = base.node_tile_offsets
    verts, left = mesh_clusters_device(mesh, mesh.num_cells)
    assert left.size == 0  # topologically every cell sits in a fan ...
    overts, oleft, ocells = mesh_clusters_ordered_device(mesh, mesh.num_cells)
    assert overts.shape[0] * 6 + oleft.size == mesh.num_cells
    if fraction == 1.0:
        assert overts.shape[0] == 0  # ... but none lists its cells the way the cluster kernels call the function
    else:
        assert 0 < overts.shape[0] < ncl
        # the qualifying clusters: cell t lists the cluster's vertices in table order
        T = np.array([[0, 1, 3, 7], [0, 1, 7, 5], [0, 5, 7, 4], [0, 3, 2, 7], [0, 6, 4, 7], [0, 2, 6, 7]])
        v, cc = overts.cpu().numpy(), ocells.cpu().numpy()
        assert all(np.array_equal(cells[cc[p, t]], v[p, T[t]]) for p in range(v.shape[0]) for t in range(6))
    case = _periodic_case(mesh, f"ufcx_permuted_{fraction}", rule=_unsymmetric_rule())
    ma, va = _taken(case)
    if fraction == 1.0:
        assert ma.kernel_name == "ufcx_rowblock" and va.kernel_name == "ufcx_ownblock"
    else:
        assert ma.kernel_name == "ufcx_cube" and ma.leftover is not None and va.kernel_name == "ufcx_cube_own"
    _check(oracle, case)


def test_unsymmetric_element_tensor(oracle):
    """a(u, v) = inner(grad u, grad v) + (beta . grad u) v: A_e is not symmetric -- hand-written UFCx text"""
    from dolfinx_mpc_amd import fem

    src = r"""
void tt_convection(double* restrict A, const double* restrict w, const double* restrict c, const double* restrict coordinate_dofs,
                   const int* restrict entity_local_index, const uint8_t* restrict quadrature_permutation, void* custom_data)
{
  const double* x = coordinate_dofs;
  double J[3][3];
  for (int r = 0; r < 3; ++r)
    for (int d = 0; d < 3; ++d)
      J[r][d] = x[3 * (d + 1) + r] - x[r];
  const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1], c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2], c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
  const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
  double K[3][3];
  K[0][0] = c00 / det; K[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det; K[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
  K[1][0] = c01 / det; K[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det; K[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
  K[2][0] = c02 / det; K[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det; K[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
  double g[4][3];
  for (int a = 0; a < 3; ++a)
  {
    g[0][a] = -(K[0][a] + K[1][a] + K[2][a]);
    g[1][a] = K[0][a]; g[2][a] = K[1][a]; g[3][a] = K[2][a];
  }
  const double beta[3] = {1.0, -2.0, 0.5};
  const double vol = fabs(det) / 6.0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
    {
      double s = 0.0, bj = 0.0;
      for (int a = 0; a < 3; ++a)
      {
        s += g[i][a] * g[j][a];
        bj += beta[a] * g[j][a];
      }
      A[4 * i + j] += vol * (s + 0.25 * bj);
    }
}
"""
    base = case_cube_periodic(6, 1, 0.3, reorder=(2, 2, 2))
    V = base.V
    a = fem.form_ufcx([V, V], src, "tt_convection")
    _a, L = _forms(V)
    case = Case("ufcx_convection", V, a, L, base.bcs, base.raw)
    ma, _va = _taken(case)
    assert ma.kernel_name == "ufcx_cube"
    ref = oracle_outputs(oracle, case)["A"]
    assert abs(ref - ref.T).max() > 1e-3  # the assembled matrix is visibly unsymmetric
    _check(oracle, case)


def test_coefficients_and_constants_through_the_clusters(oracle):
    from dolfinx_mpc_amd import fem

    base = case_cube_periodic(8, 1, 0.3, reorder=(4, 4, 4)).mesh
    case = _periodic_case(base, "ufcx_clusters_coefficient", with_coefficient=True, constant=fem.Constant(0.7))
    ma, va = _taken(case)
    assert ma.kernel_name == "ufcx_cube" and ma.cube_cells and va.kernel_name == "ufcx_cube_own" and va.cube_cells
    _check(oracle, case)


def _partly_permuted_mesh(fraction=0.3, seed=5):
    from dolfinx_mpc_amd.mesh import Mesh

    base = case_cube_periodic(6, 1, 0.3, reorder=(2, 2, 2)).mesh
    rng = np.random.default_rng(seed)
    cells = base.geometry.dofmap.copy()
    for g in np.flatnonzero(rng.random(cells.shape[0] // 6) < fraction):
        c = 6 * g + rng.integers(6)
        p = rng.permutation(4)
        while np.array_equal(p, np.arange(4)):
            p = rng.permutation(4)
        cells[c] = cells[c][p]
    mesh = Mesh(base.geometry.x, cells, "tetrahedron")
    mesh.node_tile_offsets = base.node_tile_offsets
    return mesh


def test_coefficients_reach_the_leftover_cells(oracle):
    """clusters whose cells use another vertex order are assembled by the per-cell kernels through a
    form restricted to those cells -- it must carry the coefficient (the imported text reads w[] for every cell):
    Functions, and a caller-packed array"""
    from dolfinx_mpc_amd import fem

    mesh = _partly_permuted_mesh()
    case = _periodic_case(mesh, "ufcx_leftover_coefficient", with_coefficient=True, constant=fem.Constant(0.7))
    ma, va = _taken(case)
    assert ma.kernel_name == "ufcx_cube" and ma.leftover is not None and ma.cube_cells
    assert va.kernel_name == "ufcx_cube_own" and va.leftover is not None
    _check(oracle, case, "Function coefficient")
    # the same with the coefficient handed over as an already packed array (dolfinx pack_coefficients layout)
    packed = np.ascontiguousarray(case.a.integrals[0].coeffs)
    V = case.V
    cd = 1
    from dolfinx_mpc_amd.codegen import BENCH_PERIODIC_F, generate
    from dolfinx_mpc_amd.quadrature import make_quadrature

    sa, na = generate("stiffness", "tetrahedron", 1, 1, make_quadrature("tetrahedron", 0), coefficient_degree=cd, use_constant=True)
    sl, nl = generate("source", "tetrahedron", 1, 1, make_quadrature("tetrahedron", 5), fexpr=BENCH_PERIODIC_F, coefficient_degree=cd,
                      use_constant=True)
    a2 = fem.form_ufcx([V, V], sa, na, coefficient=packed, constant=fem.Constant(0.7))
    L2 = fem.form_ufcx([V], sl, nl, coefficient=packed.copy(), constant=fem.Constant(0.7))
    case2 = Case("ufcx_leftover_packed", V, a2, L2, case.bcs, case.raw)
    _check(oracle, case2, "packed coefficient")


def test_changed_coefficient_values_are_read_on_the_next_call(oracle):
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import fem

    base = case_cube_periodic(6, 1, 0.3, reorder=(2, 2, 2)).mesh
    V = fem.functionspace(base, ("Lagrange", 1))
    bc = fem.dirichletbc(0.3, fem.locate_dofs_geometrical(V, _walls_yz), V)
    f = fem.Function(V)
    f.interpolate(lambda x: 1.0 + x[0])
    a, L = _forms(V, coefficient=f)
    case = Case("ufcx_cluster_live_coefficient", V, a, L, [bc], periodic_raw(V, [bc]))
    mpc = product_mpc(case)
    A = dm.assemble_matrix(a, mpc, bcs=[bc])
    b = dm.assemble_vector(L, mpc)
    f.interpolate(lambda x: 2.0 - x[1] * x[2])
    dm.assemble_matrix(a, mpc, bcs=[bc], A=A)
    dm.assemble_vector(L, mpc, b=b)
    ref = oracle_outputs(oracle, case)
    assert abs(A.to_scipy().data - ref["A"].data).max() <= 1e-12 * abs(ref["A"].data).max()
    assert abs(b.numpy() - ref["b"]).max() <= 1e-12 * abs(ref["b"]).max()


def test_switches_turn_the_cluster_entries_off(oracle, monkeypatch):
    base = case_cube_periodic(6, 1, 0.3, reorder=(2, 2, 2)).mesh
    case = _periodic_case(base, "ufcx_no_cube")
    monkeypatch.setenv("MPCX_NO_CUBE", "1")
    ma, va = _taken(case)
    assert ma.kernel_name == "ufcx_rowblock" and va.kernel_name == "ufcx_ownblock"
    monkeypatch.delenv("MPCX_NO_CUBE")
    monkeypatch.setenv("MPCX_FORCE_KERNEL", "matrix=ufcx_rowblock,vector=ufcx_ownblock")
    ma, va = _taken(case)
    assert ma.kernel_name == "ufcx_rowblock" and va.kernel_name == "ufcx_ownblock"
    _check(oracle, case)


@pytest.mark.parametrize("mode", ["strict", "reciprocal", "finite", "fast"])
def test_floating_point_modes_of_the_imported_text(oracle, monkeypatch, mode):
    """MPCX_UFCX_FP (csrc/mpcx_ufcx.cpp): every mode stays inside the parity bound on the cluster kernels and on the per-cell
    ones (the oracle compiles the same text with gcc -O2, strict semantics)"""
    monkeypatch.setenv("MPCX_UFCX_FP", mode)
    base = case_cube_periodic(6, 1, 0.3, reorder=(2, 2, 2)).mesh
    case = _periodic_case(base, "ufcx_fp_" + mode, with_coefficient=True)
    ma, va = _taken(case)
    assert ma.kernel_name == "ufcx_cube" and va.kernel_name == "ufcx_cube_own"
    _check(oracle, case, mode)
    monkeypatch.setenv("MPCX_NO_CUBE", "1")
    _check(oracle, case, mode + " per cell")
