"""Hexahedra (Q1): the default cell of python/benchmarks/bench_periodic.py (:38, :199-200 -- ``--tet`` is opt-in).

There is no built-in hexahedron kernel: ``fem.form_*`` on a hexahedral mesh generate the element kernel as UFCx C
text (dolfinx_mpc_amd/codegen.py generate_hex: trilinear geometry, Jacobian at every quadrature point), the oracle
compiles it with gcc and calls it through the function pointer (cpp/assemble_matrix.cpp:438-439), the product
compiles it with hipRTC into the LDS row-block kernels.

CPU: the generated text against closed forms that do not go through the generator's tables (tensor products of the
1D P1 matrices on boxes; exact energies of linear fields on sheared cells; volumes), and the constrained operators'
identities on the oracle.  GPU (-m gpu): product == oracle, both algorithms, orderly / tiled / shuffled numberings,
warped (genuinely trilinear) cells, coefficients and constants, elasticity with slip."""

import numpy as np
import pytest
import scipy.sparse as sp

from dolfinx_mpc_amd import fem
from dolfinx_mpc_amd.mesh import Mesh, create_box, create_unit_cube
from problems import (Case, case_cube_elasticity_slip, case_cube_periodic, empty_raw, oracle_outputs, periodic_raw,
                      product_outputs, warped)


def _kron3(az, ay, ax):
    # local vertex v = x + 2 y + 4 z: z is the slowest index
    return np.kron(az, np.kron(ay, ax))


def _box_element(h):
    """closed-form Q1 stiffness and mass of a box cell with sides h: tensor products of the 1D P1 matrices"""
    K1 = [np.array([[1.0, -1.0], [-1.0, 1.0]]) / hi for hi in h]
    M1 = [hi / 6.0 * np.array([[2.0, 1.0], [1.0, 2.0]]) for hi in h]
    K = _kron3(M1[2], M1[1], K1[0]) + _kron3(M1[2], K1[1], M1[0]) + _kron3(K1[2], M1[1], M1[0])
    return K, _kron3(M1[2], M1[1], M1[0])


def _assemble_closed_form(mesh, Ke):
    d = mesh.geometry.dofmap.astype(np.int64)
    n = mesh.num_nodes
    rows = np.repeat(d, 8, axis=1).reshape(-1)
    cols = np.tile(d, (1, 8)).reshape(-1)
    return sp.csr_matrix((np.tile(Ke.reshape(-1), d.shape[0]), (rows, cols)), shape=(n, n))


def _unconstrained(V, a=None, L=None):
    return Case("hex", V, a, L, [], empty_raw())


def test_box_cells_match_tensor_product_closed_forms(oracle):
    mesh = create_box((0.0, 0.0, 0.0), (1.5, 1.0, 0.8), (3, 2, 4), "hexahedron")
    V = fem.functionspace(mesh, ("Lagrange", 1))
    Ke, Me = _box_element((0.5, 0.5, 0.2))
    A = oracle_outputs(oracle, _unconstrained(V, fem.form_stiffness(V)))["A"]
    M = oracle_outputs(oracle, _unconstrained(V, fem.form_mass(V)))["A"]
    assert abs(A - _assemble_closed_form(mesh, Ke)).max() < 1e-13
    assert abs(M - _assemble_closed_form(mesh, Me)).max() < 1e-14
    # the classic unit-cube Q1 Laplace entries: 1/3 on the diagonal, 0 along edges, -1/12 across faces and the body
    K1, _ = _box_element((1.0, 1.0, 1.0))
    assert np.allclose(K1[0], [1 / 3, 0, 0, -1 / 12, 0, -1 / 12, -1 / 12, -1 / 12])


def test_sheared_cells_linear_fields_and_volume(oracle):
    """affine (sheared) hexahedra: the Q1 space holds every linear field exactly and the 2x2x2 rule integrates the
    transformed integrand exactly: u^T K u = |grad u|^2 vol, 1^T M 1 = vol, b(f = 1 + x - 2y + z/2) = int f phi_i"""
    mesh = create_unit_cube(3, 2, 2, "hexahedron")
    F = np.array([[1.0, 0.3, -0.2], [0.1, 0.9, 0.25], [0.0, -0.15, 1.2]])
    mesh.geometry.x = mesh.geometry.x @ F.T
    vol = abs(np.linalg.det(F))
    V = fem.functionspace(mesh, ("Lagrange", 1))
    x = V.tabulate_dof_coordinates()
    A = oracle_outputs(oracle, _unconstrained(V, fem.form_stiffness(V)))["A"]
    M = oracle_outputs(oracle, _unconstrained(V, fem.form_mass(V)))["A"]
    g = np.array([0.7, -1.3, 0.4])
    u = x @ g + 2.0
    assert abs(u @ (A @ u) - g @ g * vol) < 1e-12
    assert abs(A @ np.ones(x.shape[0])).max() < 1e-13
    assert abs(M.sum() - vol) < 1e-13
    b = oracle_outputs(oracle, _unconstrained(V, None, fem.form_source(V, fem.FN_LINEAR)))["b"]
    f = 1.0 + x[:, 0] - 2.0 * x[:, 1] + 0.5 * x[:, 2]  # FN_LINEAR, the same function on every cell type
    assert abs(b - M @ f).max() < 1e-13  # f is in the space: (f, phi_i) = M f


def test_warped_cells_volume_and_patch(oracle):
    """genuinely trilinear cells: det J is a polynomial of degree 2 per variable, the 2x2x2 rule integrates it exactly
    -> 1^T M 1 is the volume of the (unmoved) unit cube; constants stay in the kernel of the stiffness matrix"""
    mesh = warped(create_unit_cube(4, 3, 5, "hexahedron"))
    V = fem.functionspace(mesh, ("Lagrange", 1))
    A = oracle_outputs(oracle, _unconstrained(V, fem.form_stiffness(V)))["A"]
    M = oracle_outputs(oracle, _unconstrained(V, fem.form_mass(V)))["A"]
    assert abs(M.sum() - 1.0) < 1e-13
    assert abs(A @ np.ones(mesh.num_nodes)).max() < 1e-13
    assert abs(A - A.T).max() < 1e-14
    # and the generated kernel does depend on the quadrature point's Jacobian: an affine kernel would see the
    # corner Jacobian only.  Compare with numpy quadrature of one warped cell written here, not in the generator
    c = 17
    xc = mesh.geometry.x[mesh.geometry.dofmap[c]]
    p = 0.5 + np.array([-1.0, 1.0]) / (2.0 * np.sqrt(3.0))
    Ke = np.zeros((8, 8))
    for X in p:
        for Y in p:
            for Z in p:
                N = lambda t, b: t if b else 1.0 - t  # noqa: E731
                D = lambda b: 1.0 if b else -1.0  # noqa: E731
                dphi = np.array([[D(v & 1) * N(Y, v >> 1 & 1) * N(Z, v >> 2 & 1),
                                  N(X, v & 1) * D(v >> 1 & 1) * N(Z, v >> 2 & 1),
                                  N(X, v & 1) * N(Y, v >> 1 & 1) * D(v >> 2 & 1)] for v in range(8)])  # (8, 3)
                J = xc.T @ dphi  # J[r][d] = sum_v x_v[r] dphi_v/dX_d
                G = dphi @ np.linalg.inv(J)
                Ke += 0.125 * abs(np.linalg.det(J)) * G @ G.T
    d = mesh.geometry.dofmap[c]
    one = oracle_outputs(oracle, _unconstrained(V, fem.form_stiffness(V, cells=np.array([c], dtype=np.int32))))["A"]
    assert abs(one[d][:, d].toarray() - Ke).max() < 1e-14


def test_hex_mesh_topology():
    mesh = create_unit_cube(3, 4, 5, "hexahedron")
    assert mesh.num_cells == 60 and mesh.geometry.dofmap.shape[1] == 8
    f = mesh.exterior_facets()
    assert f.shape[0] == 2 * (12 + 15 + 20)
    mid = mesh.facet_midpoints(f)
    on = np.isclose(mid, 0.0) | np.isclose(mid, 1.0)
    assert on.any(axis=1).all()
    top = mesh.locate_exterior_facets(lambda x: np.isclose(x[2], 1.0))
    V = fem.functionspace(mesh, ("Lagrange", 1, (3,)))
    dofs = fem.locate_dofs_topological(V, 2, top)
    assert np.array_equal(dofs, fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[2], 1.0)))
    ce, ev = mesh.edges()
    assert ce.shape == (60, 12) and ev.shape[0] == 3 * 5 * 6 + 4 * 4 * 6 + 5 * 4 * 5
    V2 = fem.functionspace(mesh, ("Lagrange", 2))  # Q2: 27 dofs per cell (elements.py), generated kernels
    assert V2.element_ndofs == 27 and V2.num_dofs == 7 * 9 * 11
    V3 = fem.functionspace(mesh, ("Lagrange", 3))  # Q3: 64 dofs per cell, four nodes per face in the face's global frame
    assert V3.element_ndofs == 64 and V3.num_dofs == 10 * 13 * 16
    V4 = fem.functionspace(mesh, ("Lagrange", 4))  # Q4 (round 5): 125 dofs per cell, nine nodes per face
    assert V4.element_ndofs == 125 and V4.num_dofs == 13 * 17 * 21
    with pytest.raises(NotImplementedError):
        fem.functionspace(mesh, ("Lagrange", 5))


def test_tiled_hex_mesh_is_the_same_mesh():
    a = create_unit_cube(4, 4, 4, "hexahedron")
    b = create_unit_cube(4, 4, 4, "hexahedron", reorder=(2, 2, 2))
    ka = np.unique(np.round(a.geometry.x[a.geometry.dofmap].reshape(a.num_cells, -1), 9), axis=0)
    kb = np.unique(np.round(b.geometry.x[b.geometry.dofmap].reshape(b.num_cells, -1), 9), axis=0)
    assert np.array_equal(ka, kb) and b.node_tile_offsets is not None


HEX_CASES = [
    lambda: case_cube_periodic(4, cell_type="hexahedron"),
    lambda: case_cube_periodic(5, bc_value=0.3, cell_type="hexahedron", warp=True),
    lambda: case_cube_periodic(6, cell_type="hexahedron", reorder=(2, 2, 2)),
    lambda: case_cube_periodic(4, bc_value=1.0, cell_type="hexahedron", numbering="shuffled", warp=True),
    lambda: case_cube_periodic(6, cell_type="hexahedron", numbering="spatial"),
    lambda: case_cube_elasticity_slip(3, cell_type="hexahedron"),
    lambda: case_cube_elasticity_slip(3, cell_type="hexahedron", warp=True, numbering="shuffled"),
]


def _coefficient_case(warp):
    """python/tests/test_mpc_pipeline.py:45 shape on hexahedra: a Q1 coefficient and a constant in both forms"""
    mesh = create_unit_cube(4, 3, 3, "hexahedron")
    if warp:
        mesh = warped(mesh)
    V = fem.functionspace(mesh, ("Lagrange", 1))
    w = fem.Function(V)
    w.interpolate(lambda x: 1.0 + x[0] * x[1] + 0.5 * x[2])
    c = fem.Constant(2.5)
    dofs = fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[1], 0) | np.isclose(x[1], 1))
    bc = fem.dirichletbc(0.7, dofs, V)
    a = fem.form_stiffness(V, constant=c, coefficient=w) + fem.form_mass(V, coefficient=w)
    L = fem.form_source(V, fem.FN_BENCH_PERIODIC, constant=c, coefficient=w)
    return Case("hex_coefficient" + ("_warped" if warp else ""), V, a, L, [bc], periodic_raw(V, [bc], scale=0.5), scale=0.5)


HEX_CASES += [lambda: _coefficient_case(False), lambda: _coefficient_case(True)]


@pytest.mark.parametrize("make", HEX_CASES, ids=[f"hex{i}" for i in range(len(HEX_CASES))])
def test_oracle_constrained_operators_on_hexahedra(oracle, make):
    """the identities SURVEY 8c lists, on the oracle with the generated kernel: slave rows / columns are the identity,
    A is symmetric, K^T A K of the unconstrained matrix"""
    case = make()
    out = oracle_outputs(oracle, case)
    A = out["A"]
    assert abs(A - A.T).max() < 1e-12 * abs(A).max()
    slaves, masters, coeffs, _, offsets = case.raw
    bs = case.V.dofmap.bs
    n = case.V.num_dofs
    is_bc = np.zeros(n, dtype=np.int8)
    for bc in case.bcs:
        bc.mark_dofs(is_bc)
    for s in slaves[:50]:
        row = A[int(s)].toarray().ravel()
        assert row[int(s)] == 1.0 and abs(np.delete(row, int(s))).max() == 0.0
    # K^T A0 K with Dirichlet rows / columns removed, against the unconstrained oracle matrix
    A0 = oracle_outputs(oracle, Case("u", case.V, case.a, None, [], empty_raw()))["A"]
    K = sp.lil_matrix((n, n))
    K.setdiag(1.0)
    for i, s in enumerate(slaves):
        K[int(s), int(s)] = 0.0
        for j in range(offsets[i], offsets[i + 1]):
            K[int(s), int(masters[j])] = coeffs[j]
    K = K.tocsr()
    R = (K.T @ A0 @ K).tolil()
    bcd = np.flatnonzero(is_bc)
    R[bcd, :] = 0.0
    R[:, bcd] = 0.0
    for d in bcd:
        R[d, d] = case.diagval
    for s in slaves:
        R[int(s), int(s)] = 1.0
    assert abs(R.tocsr() - A).max() < 1e-11 * abs(A).max()
    assert bs in (1, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("alg", ["auto", "rowblock", "atomic", "generated", "no_affine", "points"])
@pytest.mark.parametrize("make", HEX_CASES, ids=[f"hex{i}" for i in range(len(HEX_CASES))])
def test_gpu_hexahedra_match_oracle(oracle, make, alg, monkeypatch):
    """auto: the built-in hexahedron kernels where they apply (scalar stiffness / source forms without coefficient: slots
    of (row block, cell), thread per cell; box cells evaluate the benchmark's right-hand side factor by factor on the
    tensor grid of their Gauss points) + the generated kernels for the constrained cells and everything else;
    rowblock: built-in matrix kernel, generated vector kernel; generated: the generated kernels everywhere
    (MPCX_NO_CUBE); no_affine: the quadrature path of the built-in matrix kernel also on parallelepipeds; points: the
    right-hand side point by point on box cells too (MPCX_BOX_GRID=0)"""
    if alg == "generated":
        monkeypatch.setenv("MPCX_NO_CUBE", "1")
    if alg == "no_affine":
        monkeypatch.setenv("MPCX_HEX_NO_AFFINE", "1")
    monkeypatch.setenv("MPCX_BOX_GRID", "0" if alg == "points" else "1")
    if alg in ("generated", "no_affine", "points"):
        alg = "auto"
    case = make()
    ref = oracle_outputs(oracle, case)
    out = product_outputs(case, algorithm=alg)
    assert np.array_equal(out["A"].indptr, ref["A"].indptr) and np.array_equal(out["A"].indices, ref["A"].indices)
    assert abs(out["A"].data - ref["A"].data).max() <= 1e-12 * max(1.0, abs(ref["A"]).max()), case.name
    for k in ("b", "b_lifted"):
        assert abs(out[k] - ref[k]).max() <= 1e-12 * max(1.0, abs(ref[k]).max()), f"{case.name} {k}"


@pytest.mark.gpu
def test_gpu_hexahedra_default_route():
    import importlib

    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd.la import create_vector

    am = importlib.import_module("dolfinx_mpc_amd.assemble_matrix")
    av = importlib.import_module("dolfinx_mpc_amd.assemble_vector")
    from problems import product_mpc

    case = case_cube_periodic(8, cell_type="hexahedron", reorder=(4, 4, 4))
    mpc = product_mpc(case)
    A = dm.create_matrix(case.a, mpc)
    args, keep = am.matrix_args(case.a, 0, A, mpc, mpc, case.bcs, 2)
    assert args.kernel_name == "hex_cube" and args.second is not None  # built-in bulk + generated kernel on the slave cells
    args, keep = av.vector_args(case.L, 0, create_vector(case.V), mpc, 0)
    assert args.kernel_name == "hex_own" and args.second is not None
    # forms without a built-in twin (here: with a coefficient) run the generated kernel inside the row blocks
    cc = _coefficient_case(False)
    mpc = product_mpc(cc)
    A = dm.create_matrix(cc.a, mpc)
    args, keep = am.matrix_args(cc.a, 0, A, mpc, mpc, cc.bcs, 2)
    assert args.kernel_name == "ufcx_rowblock"
    args, keep = av.vector_args(cc.L, 0, create_vector(cc.V), mpc, 0)
    assert args.kernel_name == "ufcx_ownblock"


@pytest.mark.gpu
def test_gpu_hexahedra_at_size_properties():
    """64^3 hexahedra (274 625 dofs), warped: the matrix is symmetric, constants are in the kernel of the unconstrained
    stiffness matrix, the mass matrix sums to the volume, the row-block and the atomic route agree"""
    import dolfinx_mpc_amd as dm
    from problems import product_mpc

    case = case_cube_periodic(64, cell_type="hexahedron", reorder=(8, 8, 8), warp=True)
    mpc = product_mpc(case)
    A = dm.assemble_matrix(case.a, mpc, bcs=case.bcs).to_scipy()
    A2 = dm.assemble_matrix(case.a, mpc, bcs=case.bcs, algorithm="atomic").to_scipy()
    assert abs(A - A2).max() < 1e-12
    assert abs(A - A.T).max() < 1e-12
    free = product_mpc(Case("free", case.V, None, None, [], empty_raw()))
    A0 = dm.assemble_matrix(case.a, free).to_scipy()
    assert abs(A0 @ np.ones(case.V.num_dofs)).max() < 1e-12
    M = dm.assemble_matrix(fem.form_mass(case.V), free).to_scipy()
    assert abs(M.sum() - 1.0) < 1e-11
    # vector: thread-per-cell built-in kernel (owner-computes) against the generated kernel with device atomics
    b = dm.assemble_vector(case.L, mpc).numpy().copy()
    b2 = dm.assemble_vector(case.L, mpc, algorithm="atomic").numpy().copy()
    assert abs(b - b2).max() <= 1e-12 * max(1.0, abs(b2).max())
    # (the determinant of a trilinear map is quadratic in every variable: two points per direction are exact)
    one = dm.assemble_vector(fem.form_source(case.V, fem.FN_ONE, quadrature_degree=3), free).numpy()
    assert abs(one.sum() - 1.0) < 1e-11


@pytest.mark.gpu
def test_gpu_hexahedra_blocks_split_by_cell_shape(oracle, monkeypatch):
    """a mesh of parallelepipeds (x > 0.375) and genuinely trilinear cells, numbered in tiles of 4 x 4 x 4 nodes = one row block: the row blocks all of whose cells are
    parallelepipeds go to the closed-form kernel instance, the others to the instance that looks at every cell;
    moving the mesh afterwards rebuilds the split"""
    import importlib

    import dolfinx_mpc_amd as dm
    from problems import product_mpc, warped

    am = importlib.import_module("dolfinx_mpc_amd.assemble_matrix")
    monkeypatch.setattr(am, "HEX_MAX_ROWS", 64)
    monkeypatch.setattr(am, "HEX_MAX_NNZ", 64 * 27)
    case = case_cube_periodic(8, cell_type="hexahedron", warp="half", reorder=(4, 4, 4))
    ref = oracle_outputs(oracle, case)
    mpc = product_mpc(case)
    A = dm.create_matrix(case.a, mpc)
    args, keep = am.matrix_args(case.a, 0, A, mpc, mpc, case.bcs, 2)
    chain, u = [], args
    while u is not None:
        chain.append(u)
        u = u.second
    assert [int(u.cube_flags) for u in chain[:2]] == [1, 0] and chain[0].cube_block_ids and chain[1].cube_block_ids
    assert int(chain[0].plan.num_blocks) > 0 and int(chain[1].plan.num_blocks) > 0
    assert chain[-1].n_entities == 0 and chain[-1].n_slave_entities > 0  # the imported kernel on the slave cells
    dm.assemble_matrix(case.a, mpc, bcs=case.bcs, diagval=case.diagval, A=A)
    got = A.to_scipy()
    assert abs(got.data - ref["A"].data).max() <= 1e-12 * max(1.0, abs(ref["A"]).max())
    # move the mesh: every cell becomes trilinear; same matrix object, same form
    warped(case.V.mesh)
    ref2 = oracle_outputs(oracle, case)
    assert abs(ref2["A"] - ref["A"]).max() > 1e-3
    dm.assemble_matrix(case.a, mpc, bcs=case.bcs, diagval=case.diagval, A=A)
    got2 = A.to_scipy()
    assert abs(got2.data - ref2["A"].data).max() <= 1e-12 * max(1.0, abs(ref2["A"]).max())
    b = dm.assemble_vector(case.L, mpc).numpy()
    assert abs(b - ref2["b"]).max() <= 1e-12 * max(1.0, abs(ref2["b"]).max())
