"""The constraint builders of SURVEY 8(f4) against what the reference's builders are specified to produce
(python/src/dolfinx_mpc/multipointconstraint.py:225-501; cpp/PeriodicConstraint.h, cpp/SlipConstraint.h,
cpp/ContactConstraint.h, serial branches).  The reference cannot run here, so each builder is pinned two ways:
(i) a property the construction must have whatever the mesh -- a slave tied to the basis functions of the cell at a
point reproduces every function of the space's polynomial degree there -- and (ii) for the contact-slip condition a
statement-by-statement restatement of the C++ loops (plain Python, no shared code) whose arrays must agree."""

import numpy as np
import pytest

import dolfinx_mpc_amd as dm
from dolfinx_mpc_amd import fem
from dolfinx_mpc_amd.mesh import (CONTACT_BOTTOM_INTERFACE, CONTACT_TOP_INTERFACE, TET_FACETS, MeshTags, create_stacked_cubes,
                                  create_unit_cube, create_unit_square, rotation_matrix)
from dolfinx_mpc_amd.multipointconstraint import create_normal_approximation, locate_points


def _poly(degree):
    if degree == 1:
        return lambda x: 0.3 + 1.1 * x[0] - 0.7 * x[1] + 0.45 * x[2]
    return lambda x: 0.3 + 1.1 * x[0] - 0.7 * x[1] + 0.45 * x[2] + 0.8 * x[0] * x[1] - 0.35 * x[1] * x[2] + 0.6 * x[2] * x[2]


@pytest.mark.parametrize("degree", [1, 2])
@pytest.mark.parametrize("bs", [1, 3])
def test_periodic_constraint_on_non_matching_points(degree, bs):
    """cpp/PeriodicConstraint.h:139-196: the mapped slave point is located in a cell and the slave is tied to that
    cell's dofs with their basis values there (x scale).  Property: for every polynomial p of the space's degree the
    constraint row reproduces scale * p(relation(x_s)); matching points get ONE master with the coefficient exactly"""
    mesh = create_unit_cube(3, 3, 3)
    V = fem.functionspace(mesh, ("Lagrange", degree, (bs,))) if bs > 1 else fem.functionspace(mesh, ("Lagrange", degree))
    x = V.tabulate_dof_coordinates()
    bc = fem.dirichletbc(0.0, fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[2], 0.0)), V)

    def relation(x):  # the face x = 1 onto the face x = 0, squeezed in y: mapped points fall inside cells / facets
        out = x.copy()
        out[0] = 1.0 - x[0]
        out[1] = 0.937 * x[1] + 0.01
        return out

    scale = 0.8
    mpc = dm.MultiPointConstraint(V)
    mpc.create_periodic_constraint_geometrical(V, lambda x: np.isclose(x[0], 1.0), relation, [bc], scale)
    mpc.finalize()
    face = np.flatnonzero(np.isclose(x[:, 0], 1.0) & ~np.isclose(x[:, 2], 0.0))
    assert np.array_equal(mpc.slaves, (face[:, None] * bs + np.arange(bs)[None, :]).reshape(-1))  # bc blocks left out
    p = _poly(degree)
    pv = p(x.T)
    moff, m, c = mpc.masters.offsets, mpc.masters.array, mpc.coefficients()[0]
    for s in mpc.slaves:
        sl = slice(moff[s], moff[s + 1])
        assert moff[s + 1] > moff[s]
        assert np.all(m[sl] % bs == s % bs)  # component k is tied to component k
        target = scale * p(relation(x[s // bs][:, None]))[0]
        assert abs(c[sl] @ pv[m[sl] // bs] - target) < 1e-12
        assert np.all(np.abs(c[sl]) > 500 * np.finfo(float).eps)
    # matching points: exactly one master, coefficient = scale to the last bit
    m2 = dm.MultiPointConstraint(V)

    def plain(x):
        out = x.copy()
        out[0] = 1.0 - x[0]
        return out

    m2.create_periodic_constraint_geometrical(V, lambda x: np.isclose(x[0], 1.0), plain, [bc], scale)
    m2.finalize()
    assert np.all(np.diff(m2.masters.offsets)[m2.slaves] == 1) and np.all(m2.coefficients()[0] == scale)


def _facet_tags(mesh, *markers):
    """MeshTags over exterior facets: markers = (value, predicate on facet midpoints)"""
    f = mesh.exterior_facets()
    mid = mesh.facet_midpoints(f)
    ents, vals = [], []
    for value, pred in markers:
        sel = np.asarray(pred(mid.T), dtype=bool)
        ents.append(f[sel])
        vals.append(np.full(int(sel.sum()), value, dtype=np.int32))
    return MeshTags(mesh, mesh.tdim - 1, np.concatenate(ents, axis=0), np.concatenate(vals))


@pytest.mark.parametrize("degree", [1, 2])
def test_periodic_constraint_topological_equals_geometrical(degree):
    """python/src/dolfinx_mpc/multipointconstraint.py:225-281: the closure dofs of the tagged facets"""
    mesh = create_unit_cube(3, 2, 3)
    V = fem.functionspace(mesh, ("Lagrange", degree))
    mt = _facet_tags(mesh, (7, lambda x: np.isclose(x[0], 1.0)), (3, lambda x: np.isclose(x[1], 0.0)))
    bc = fem.dirichletbc(0.0, fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[2], 1.0)), V)

    def rel(x):
        out = x.copy()
        out[0] = 1.0 - x[0]
        return out

    a, b = dm.MultiPointConstraint(V), dm.MultiPointConstraint(V)
    a.create_periodic_constraint_topological(V, mt, 7, rel, [bc], 0.5)
    b.create_periodic_constraint_geometrical(V, lambda x: np.isclose(x[0], 1.0), rel, [bc], 0.5)
    a.finalize(), b.finalize()
    assert a.slaves.size > 0 and np.array_equal(a.slaves, b.slaves)
    assert np.array_equal(a.masters.array, b.masters.array) and np.array_equal(a.coefficients()[0], b.coefficients()[0])


@pytest.mark.parametrize("degree", [1, 2])
def test_slip_constraint_with_the_reference_signature(degree):
    """create_slip_constraint(space, (meshtags, marker), v, bcs) (multipointconstraint.py:325-399,
    cpp/SlipConstraint.h:16-175) with v = create_normal_approximation (cpp/utils.h:201-267) on a rotated cube: slave =
    the component with the largest |n_i|, masters = the other components of the block, c = -n_i / n_s; blocks
    touched by a Dirichlet condition are left out whole"""
    mesh = create_unit_cube(2, 3, 2)
    R = rotation_matrix([1.0, 2.0, -0.5], 0.4)
    mesh.geometry.x = mesh.geometry.x @ R.T
    V = fem.functionspace(mesh, ("Lagrange", degree, (3,)))
    x0 = V.tabulate_dof_coordinates() @ R  # coordinates in the unrotated frame
    mt = _facet_tags(mesh, (5, lambda x: np.isclose((R.T @ x)[0], 1.0)))
    nh = create_normal_approximation(V, mt, 5)
    on_face = np.flatnonzero(np.isclose(x0[:, 0], 1.0))
    n_exact = R @ np.array([1.0, 0.0, 0.0])
    arr = nh.x.array.reshape(-1, 3)
    assert np.all(arr[np.setdiff1d(np.arange(arr.shape[0]), on_face)] == 0.0)
    for b in on_face:  # sum of k aligned unit normals divided by its squared length: direction n, length 1 / k
        nb = arr[b]
        k = 1.0 / np.linalg.norm(nb)
        assert abs(k - round(k)) < 1e-9 and round(k) >= 1 and np.allclose(nb * k, n_exact * np.sign(nb @ n_exact), atol=1e-12)
    # a Dirichlet condition on ONE component of some face blocks removes those blocks from the constraint
    pinned = on_face[::3]
    bc = fem.dirichletbc(0.0, pinned.astype(np.int32), V, component=1)
    mpc = dm.MultiPointConstraint(V)
    mpc.create_slip_constraint(V, (mt, 5), nh, [bc])
    mpc.finalize()
    free = np.setdiff1d(on_face, pinned)
    s = int(np.argmax(np.abs(n_exact)))
    assert np.array_equal(mpc.slaves, np.sort(free * 3 + s))
    moff, m, c = mpc.masters.offsets, mpc.masters.array, mpc.coefficients()[0]
    others = [k for k in range(3) if k != s]
    for b in free:
        sl = slice(moff[b * 3 + s], moff[b * 3 + s + 1])
        assert np.array_equal(m[sl], b * 3 + np.array(others))
        assert np.allclose(c[sl], [-n_exact[k] / n_exact[s] for k in others], atol=1e-12)


def contact_slip_raw_bruteforce(V, slave_facets, master_facets, nh):
    """Independent restatement (plain loops) of the serial branch of cpp/ContactConstraint.h:359-503 for P1:
    compute_block_contributions (:217-280) then compute_master_contributions (:59-160), concatenated per slave"""
    mesh = V.mesh
    x = mesh.geometry.x
    cells = mesh.geometry.dofmap
    bs = V.dofmap.bs
    snodes = sorted({int(v) for c, f in slave_facets for v in cells[c][TET_FACETS[f]]})
    mcells = []
    for c, f in master_facets:
        if int(c) not in mcells:
            mcells.append(int(c))
    out = {}
    for b in snodes:
        n = nh.x.array[b * bs:(b + 1) * bs]
        s = int(np.argmax(np.abs(n)))
        masters, coeffs = [], []
        for j in range(bs):
            if j != s and abs(n[j]) > 1e-6:
                masters.append(b * bs + j)
                coeffs.append(-n[j] / n[s])
        hit = None
        for c in sorted(mcells):
            xv = x[cells[c]]
            T = np.stack([xv[1] - xv[0], xv[2] - xv[0], xv[3] - xv[0]], axis=1)
            mu = np.linalg.solve(T, x[b] - xv[0])
            lam = np.array([1.0 - mu.sum(), mu[0], mu[1], mu[2]])
            if lam.min() >= -1e-9:
                hit = (c, lam)
                break
        assert hit is not None
        c, lam = hit
        for j in range(4):
            for k in range(bs):
                val = n[k] / n[s] * lam[j]
                if abs(val) > 1e-6:
                    masters.append(int(cells[c][j]) * bs + k)
                    coeffs.append(val)
        out[b * bs + s] = (masters, coeffs)
    return out


@pytest.mark.parametrize("n_top,n_bottom,theta", [(2, 4, 0.0), (2, 3, np.pi / 5), (3, 4, 1.1)])
def test_contact_slip_condition(n_top, n_bottom, theta):
    """create_contact_slip_condition (multipointconstraint.py:435-463, cpp/ContactConstraint.h:359-503) on the two
    stacked cubes of bench_contact_3D.py: the rows say n . u_s = n . u_m(x_s), so they hold for every displacement
    field that is linear across both bodies; and they agree with the plain-loop restatement master by master"""
    mesh, ft, _ = create_stacked_cubes(n_top, n_bottom, theta)
    V = fem.functionspace(mesh, ("Lagrange", 1, (3,)))
    nh = create_normal_approximation(V, ft, CONTACT_BOTTOM_INTERFACE)
    # make the direction generic (all three components in play) but keep it a function of the block
    nh.x.array[:] = (nh.x.array.reshape(-1, 3) + np.where(np.abs(nh.x.array.reshape(-1, 3)).sum(axis=1, keepdims=True) > 0,
                                                            np.array([[0.21, -0.13, 0.0]]), 0.0)).reshape(-1)
    mpc = dm.MultiPointConstraint(V)
    mpc.create_contact_slip_condition(ft, CONTACT_BOTTOM_INTERFACE, CONTACT_TOP_INTERFACE, nh)
    mpc.finalize()
    ref = contact_slip_raw_bruteforce(V, ft.find(CONTACT_BOTTOM_INTERFACE), ft.find(CONTACT_TOP_INTERFACE), nh)
    assert np.array_equal(mpc.slaves, np.array(sorted(ref), dtype=np.int32))
    x = V.tabulate_dof_coordinates()
    G = np.array([[0.3, -1.0, 0.2], [0.5, 0.1, -0.4], [-0.2, 0.7, 0.9]])
    u = (x @ G.T + np.array([0.1, -0.2, 0.3])).reshape(-1)  # linear field, blocked layout
    moff, m, c = mpc.masters.offsets, mpc.masters.array, mpc.coefficients()[0]
    for s in mpc.slaves:
        sl = slice(moff[s], moff[s + 1])
        rm, rc = ref[int(s)]
        # same masters with the same weights (a point on a shared face may be found in a different master cell:
        # compare as accumulated weight per master dof)
        got, want = {}, {}
        for mm, cc in zip(m[sl], c[sl]):
            got[int(mm)] = got.get(int(mm), 0.0) + cc
        for mm, cc in zip(rm, rc):
            want[int(mm)] = want.get(int(mm), 0.0) + cc
        for k in set(got) | set(want):
            assert abs(got.get(k, 0.0) - want.get(k, 0.0)) < 2e-6, (s, k)
        assert abs(c[sl] @ u[m[sl]] - u[s]) < 2e-5 * np.abs(u).max()


def test_contact_slip_raises_when_surfaces_do_not_touch():
    mesh, ft, _ = create_stacked_cubes(2, 3, 0.0)
    top = np.unique(mesh.geometry.dofmap[: 6 * 8])  # the top body's cells come first: lift all its nodes
    x = mesh.geometry.x.copy()
    x[top, 2] += 0.25
    mesh.geometry.x = x
    V = fem.functionspace(mesh, ("Lagrange", 1, (3,)))
    nh = create_normal_approximation(V, ft, CONTACT_BOTTOM_INTERFACE)
    mpc = dm.MultiPointConstraint(V)
    with pytest.raises(RuntimeError, match="No masters found on contact surface"):
        mpc.create_contact_slip_condition(ft, CONTACT_BOTTOM_INTERFACE, CONTACT_TOP_INTERFACE, nh)


def test_locate_points_returns_partition_of_unity_and_misses():
    mesh = create_unit_square(4, 3)
    V = fem.functionspace(mesh, ("Lagrange", 2))
    rng = np.random.default_rng(0)
    pts = np.concatenate([rng.random((50, 2)), np.zeros((50, 1))], axis=1)
    pts[-1] = [1.7, 0.2, 0.0]
    cells, phi = locate_points(V, pts)
    assert cells[-1] == -1 and np.all(cells[:-1] >= 0) and np.allclose(phi[:-1].sum(axis=1), 1.0)
    xd = V.tabulate_dof_coordinates()
    for i in range(49):  # sum_j phi_j(p) x_j = p
        assert np.allclose(phi[i] @ xd[V.dofmap.list[cells[i]]], pts[i], atol=1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("alg", ["atomic", "rowblock"])
@pytest.mark.parametrize("kind", ["contact_slip", "slip", "periodic_nonmatching"])
def test_gpu_assembly_with_built_constraints_matches_oracle(oracle, kind, alg):
    """the constraints these builders produce (several masters per slave, masters in the slave's own block AND across
    the interface, non-matching periodic weights) through the HIP assemblers against the oracle on the same arrays"""
    from problems import Case, oracle_outputs, product_outputs

    if kind == "contact_slip":
        mesh, ft, _ = create_stacked_cubes(2, 3, 0.7)
        V = fem.functionspace(mesh, ("Lagrange", 1, (3,)))
        nh = create_normal_approximation(V, ft, CONTACT_BOTTOM_INTERFACE)
        m = dm.MultiPointConstraint(V)
        m.create_contact_slip_condition(ft, CONTACT_BOTTOM_INTERFACE, CONTACT_TOP_INTERFACE, nh)
        z = mesh.geometry.x @ rotation_matrix([1 / np.sqrt(2), 1 / np.sqrt(2), 0], -0.7).T
        bcs = [fem.dirichletbc(np.array([0.0, 0.0, 0.0]), np.flatnonzero(np.isclose((mesh.geometry.x @ rotation_matrix(
            [1 / np.sqrt(2), 1 / np.sqrt(2), 0], 0.7))[:, 2], 0.0)).astype(np.int32), V)]
        del z
        a, L = fem.form_elasticity(V, 400.0, 250.0), fem.form_source(V, fem.FN_LINEAR)
    elif kind == "slip":
        mesh = create_unit_cube(3, 3, 3)
        mesh.geometry.x = mesh.geometry.x @ rotation_matrix([1.0, 2.0, -0.5], 0.4).T
        V = fem.functionspace(mesh, ("Lagrange", 1, (3,)))
        R = rotation_matrix([1.0, 2.0, -0.5], 0.4)
        mt = _facet_tags(mesh, (5, lambda x: np.isclose((R.T @ x)[0], 1.0)))
        bcs = [fem.dirichletbc(np.array([0.0, 0.1, 0.0]), np.flatnonzero(np.isclose((V.tabulate_dof_coordinates() @ R)[:, 0], 0.0)).astype(np.int32), V)]
        m = dm.MultiPointConstraint(V)
        m.create_slip_constraint(V, (mt, 5), create_normal_approximation(V, mt, 5), bcs)
        a, L = fem.form_elasticity(V, 400.0, 250.0), fem.form_source(V, fem.FN_LINEAR)
    else:
        mesh = create_unit_cube(4, 4, 4)
        V = fem.functionspace(mesh, ("Lagrange", 2))
        bcs = [fem.dirichletbc(0.3, fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[2], 0.0)), V)]

        def relation(x):
            out = x.copy()
            out[0] = 1.0 - x[0]
            out[1] = 0.9 * x[1] + 0.03
            return out

        m = dm.MultiPointConstraint(V)
        m.create_periodic_constraint_geometrical(V, lambda x: np.isclose(x[0], 1.0), relation, bcs, 0.7)
        a, L = fem.form_stiffness(V), fem.form_source(V, fem.FN_POLY3)
    raw = (m._slaves.copy(), m._masters.copy(), m._coeffs.copy(), m._owners.copy(), m._offsets.copy())
    assert raw[0].size > 0
    case = Case("built_" + kind, V, a, L, bcs, raw)
    ref = oracle_outputs(oracle, case)
    out = product_outputs(case, algorithm=alg)
    assert np.array_equal(out["A"].indptr, ref["A"].indptr) and np.array_equal(out["A"].indices, ref["A"].indices)
    assert abs(out["A"].data - ref["A"].data).max() <= 1e-12 * max(1.0, abs(ref["A"]).max())
    for k in ("b", "b_lifted"):
        assert abs(out[k] - ref[k]).max() <= 1e-12 * max(1.0, abs(ref[k]).max()), k


@pytest.mark.parametrize("tensor_order", [0, 1, 2])
@pytest.mark.parametrize("poly_order", [1, 2, 3])
def test_periodic_slaves_of_tensor_spaces(tensor_order, poly_order):
    """the constraint of python/tests/test_nonlinear_assembly.py:117-150 (test_homogenize): scalar, vector and
    TENSOR valued Lagrange P1-P3 spaces on an 8 x 8 square, x = 0 tied to x = 1 -- every component of every dof on the
    slave line is a slave with one master of coefficient 1 in the same component"""
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import fem
    from dolfinx_mpc_amd.mesh import create_unit_square

    mesh = create_unit_square(8, 8)
    shape = [(), (2,), (2, 2)][tensor_order]
    V = fem.functionspace(mesh, ("Lagrange", poly_order, shape)) if shape else fem.functionspace(mesh, ("Lagrange", poly_order))
    bs = int(np.prod(shape)) if shape else 1
    assert V.dofmap.bs == bs

    def rel(x):
        out = np.zeros(x.shape)
        out[0], out[1], out[2] = 1.0 - x[0], x[1], x[2]
        return out

    mpc = dm.MultiPointConstraint(V)
    mpc.create_periodic_constraint_geometrical(V, lambda x: np.isclose(x[0], 0.0), rel, [])
    mpc.finalize()
    nline = 8 * poly_order + 1
    assert mpc.slaves.size == nline * bs
    xc = V.tabulate_dof_coordinates()
    off, m = mpc.masters.offsets, mpc.masters.array
    coef = mpc.coefficients()[0]
    for s in mpc.slaves:
        assert off[s + 1] - off[s] == 1 and abs(coef[off[s]] - 1.0) < 1e-13
        mm = int(m[off[s]])
        assert mm % bs == s % bs
        assert np.allclose(xc[mm // bs], xc[s // bs] + [1.0, 0.0, 0.0])


@pytest.mark.parametrize("cell_type,degrees", [("quadrilateral", (1, 2, 3, 4)), ("hexahedron", (1, 2, 4)), ("tetrahedron", (1, 2, 4)),
                                               ("triangle", (1, 2, 3, 4))])
@pytest.mark.parametrize("N", [3, 5, 8])
def test_multiple_mpc_spaces_sparsity(cell_type, degrees, N):
    """python/tests/test_multispace_mpc.py:12-77: the pattern of a rectangular block with constraints on two DISTINCT
    (cloned) spaces has as many entries as with one constraint on both sides (the reference sweeps degrees 1, 2, 4 on all four
    cell types; degree 4 is laid out since round 5, dolfinx_mpc_amd/elements.py)"""
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import fem
    from dolfinx_mpc_amd.mesh import create_unit_cube, create_unit_square

    two = cell_type in ("quadrilateral", "triangle")
    mesh = create_unit_square(N, N, cell_type) if two else create_unit_cube(N, N, N, cell_type)
    gdim = 2 if two else 3
    atol = 5000 * np.finfo(np.float64).eps

    def periodic_boundary(x):
        return np.isclose(x[0], 1, atol=atol) | np.isclose(x[2], 1, atol=atol)

    def periodic_map(x):
        out = x.copy()
        out[0][np.isclose(x[0], 1, atol=atol)] -= 1
        out[gdim - 1][np.isclose(x[2], 1, atol=atol)] -= 1
        return out

    for deg in degrees:
        V = fem.functionspace(mesh, ("Lagrange", deg))
        Q = V.clone()
        assert Q is not V and Q.num_dofs == V.num_dofs and np.array_equal(Q.dofmap.list, V.dofmap.list)
        mpc_u = dm.MultiPointConstraint(V)
        mpc_u.create_periodic_constraint_geometrical(V, periodic_boundary, periodic_map, [], tol=atol)
        mpc_u.finalize()
        mpc_p = dm.MultiPointConstraint(Q)
        mpc_p.create_periodic_constraint_geometrical(Q, periodic_boundary, periodic_map, [], tol=atol)
        mpc_p.finalize()
        assert mpc_u.slaves.size > 0
        a01 = fem.Form([V, Q], fem.form_mass(V).integrals)  # inner(p, v) dx: rows V, columns Q (pattern only)
        a10 = fem.Form([Q, V], fem.form_mass(V).integrals)
        r0, c0 = dm.create_sparsity_pattern(a01, [mpc_u, mpc_p], where="host")
        r1, c1 = dm.create_sparsity_pattern(a01, [mpc_u, mpc_u], where="host")
        assert c0.size == c1.size and np.array_equal(r0, r1) and np.array_equal(c0, c1)
        r0, c0 = dm.create_sparsity_pattern(a10, [mpc_p, mpc_u], where="host")
        r1, c1 = dm.create_sparsity_pattern(a10, [mpc_u, mpc_u], where="host")
        assert c0.size == c1.size and np.array_equal(r0, r1) and np.array_equal(c0, c1)


def test_point_to_point_constraint(oracle):
    """python/src/dolfinx_mpc/utils/mpc_utils.py:300-420 (demo_elasticity_disconnect.py:176-190): blocks closest to two
    boundary points tied component by component, or along a vector; the constrained elasticity system satisfies the
    K^T A K identity (utils/test.py:202-242) and its solution the constraint"""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    from dolfinx_mpc_amd import fem
    from dolfinx_mpc_amd.mesh import create_unit_cube
    from dolfinx_mpc_amd.utils import create_point_to_point_constraint, determine_closest_block

    mesh = create_unit_cube(3, 3, 3)
    V = fem.functionspace(mesh, ("Lagrange", 1, (3,)))
    x = V.tabulate_dof_coordinates()
    owner, blk = determine_closest_block(V, np.array([1.02, 0.31, 0.36]))
    assert owner == 0 and np.allclose(x[blk[0]], [1.0, 1 / 3, 1 / 3])
    sl, ms, co, ow, off = create_point_to_point_constraint(V, [1.0, 1 / 3, 1 / 3], [1.0, 2 / 3, 2 / 3])
    b0, b1 = blk[0], determine_closest_block(V, [1.0, 2 / 3, 2 / 3])[1][0]
    assert np.array_equal(sl, b0 * 3 + np.arange(3)) and np.array_equal(ms, b1 * 3 + np.arange(3))
    assert np.array_equal(co, np.ones(3)) and np.array_equal(off, [0, 1, 2, 3]) and np.array_equal(ow, np.zeros(3))
    v = np.array([0.5, 0.0, -2.0])
    sl2, ms2, co2, ow2, off2 = create_point_to_point_constraint(V, [1.0, 1 / 3, 1 / 3], [1.0, 2 / 3, 2 / 3], vector=v)
    assert np.array_equal(sl2, [b0 * 3 + 2]) and np.array_equal(ms2, [b0 * 3 + 0, b1 * 3 + 0, b1 * 3 + 2])
    assert np.allclose(co2, [0.25, -0.25, 1.0]) and np.array_equal(off2, [0, 3])
    # the vector constraint in a solve: clamp x = 0, pull on the master point's block, check v . u_slave = v . u_master
    bc = fem.dirichletbc(np.zeros(3), fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[0], 0)), V)
    om = oracle.OracleMPC.from_raw(V, sl2, ms2, co2, ow2, off2)
    a = fem.form_elasticity(V, 1.0, 1.25)
    L = fem.form_source(V, fem.FN_CONSTANT_VEC, constant=np.array([1.0, 0.3, -0.2, 0.5]))
    A = oracle.assemble_matrix(a, om, bcs=[bc])
    b = oracle.assemble_vector(L, om)
    oracle.apply_lifting(b, [a], [[bc]], om)
    dofs = bc.dof_indices()[0]
    b[dofs] = 0.0
    u = spla.spsolve(A.tocsc(), b)
    oracle.homogenize(om, u)
    oracle.backsubstitution(om, u)
    assert abs(v @ u[b0 * 3:b0 * 3 + 3] - v @ u[b1 * 3:b1 * 3 + 3]) < 1e-12 * abs(u).max()


def test_verification_toolkit_agrees_with_the_oracle(oracle):
    """``utils.gather_transformation_matrix`` / ``compare_mpc_lhs`` / ``compare_mpc_rhs`` (the reference's
    utils/test.py:67-265, product side) against the oracle's own restatement on a constrained system with multi-master
    slaves, and the docstring example of utils/test.py:72-86"""
    import scipy.sparse as sp

    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import utils
    from problems import case_delaunay_periodic, oracle_mpc, oracle_outputs

    case = case_delaunay_periodic(2, 1, 5, seed=2)

    class _HostMPC:  # the finalized data without a GPU: what the toolkit reads
        pass

    om = oracle_mpc(oracle, case)
    K_ref = oracle.gather_transformation_matrix(om)
    h = _HostMPC()
    h.function_space = case.V
    h.slaves, h.num_local_slaves = om.slaves, om.num_local_slaves
    h.masters = type("Adj", (), {"offsets": om.masters_offsets, "array": om.masters})()
    h.coefficients = lambda: (om.coeffs, om.masters_offsets)
    K = utils.gather_transformation_matrix(h)
    assert K.shape == K_ref.shape and abs(K - K_ref).max() < 1e-15
    out = oracle_outputs(oracle, case)
    emp = oracle.OracleMPC.empty(case.V)
    A_org = oracle.assemble_matrix(case.a, emp, bcs=case.bcs)
    utils.compare_mpc_lhs(A_org, out["A"], h)
    utils.compare_mpc_rhs(oracle.assemble_vector(case.L, emp), out["b"], h)
    with pytest.raises(AssertionError):
        utils.compare_mpc_lhs(A_org * 1.001, out["A"], h)
    # utils/test.py:72-86: u_1 = alpha u_0 + beta u_2
    ex = _HostMPC()
    ex.function_space = type("V", (), {"num_dofs": 3})()
    ex.slaves, ex.num_local_slaves = np.array([1]), 1
    ex.masters = type("Adj", (), {"offsets": np.array([0, 0, 2, 2]), "array": np.array([0, 2])})()
    ex.coefficients = lambda: (np.array([0.3, 0.7]), None)
    assert np.allclose(utils.gather_transformation_matrix(ex).toarray(), [[1, 0], [0.3, 0.7], [0, 1]])
    assert dm.utils is utils and hasattr(dm, "NonlinearProblem")
