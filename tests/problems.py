"""Problem definitions shared by the oracle tests, the golden-fixture generator
and the GPU parity tests.  Each case mirrors a configuration of the reference's
own test-suite or benchmark (cited per case); the constraint is kept as the
raw ``add_constraint`` arrays (python/src/dolfinx_mpc/multipointconstraint.py:118-153)
so that the oracle and the product finalize it independently.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional

import numpy as np
from scipy.spatial import cKDTree

from dolfinx_mpc_amd import fem
from dolfinx_mpc_amd.mesh import create_unit_cube, create_unit_square, rotation_matrix
from dolfinx_mpc_amd.workloads import (Case, _walls_yz, case_contact_two_body, case_cube_periodic, contact_problem,  # noqa: F401
                                       contact_raw_bruteforce, empty_raw, l2b, periodic_raw, renumbered, stokes_slip_problem, warped)


def dict_constraint_raw(V, s_m_c: Dict[bytes, Dict[bytes, float]], subspace_slave=None, subspace_master=None):
    """python/src/dolfinx_mpc/dictcondition.py semantics on matching nodes."""
    x = V.tabulate_dof_coordinates()
    tree = cKDTree(x)
    bs = V.dofmap.bs

    def find(b):
        p = np.zeros(3)
        v = np.frombuffer(b, dtype=np.float64)
        p[: v.size] = v
        d, i = tree.query(p)
        assert d < 1e-9, f"no dof at {p}"
        return int(i)

    slaves, masters, coeffs, offsets = [], [], [], [0]
    for sp, md in s_m_c.items():
        sb = find(sp)
        for k in range(bs) if subspace_slave is None else [subspace_slave]:
            slaves.append(sb * bs + k)
            for mp, c in md.items():
                masters.append(find(mp) * bs + (k if subspace_master is None else subspace_master))
                coeffs.append(c)
            offsets.append(len(masters))
    return (np.array(slaves, dtype=np.int32), np.array(masters, dtype=np.int64), np.array(coeffs, dtype=np.float64),
            np.zeros(len(masters), dtype=np.int32), np.array(offsets, dtype=np.int32))


# --------------------------------------------------------------------------
def case_square_dict(degree=1, master_point=(1, 1), n=(5, 3), cell_type="triangle") -> Case:
    """python/tests/test_matrix_assembly.py:23-57 / test_vector_assembly.py:22-63 (the reference sweeps degree 1-3 on
    triangles and quadrilaterals: degree 3 and the quadrilaterals run generated kernels, dolfinx_mpc_amd/elements.py)"""
    mesh = create_unit_square(*n, cell_type)
    V = fem.functionspace(mesh, ("Lagrange", degree))
    s_m_c = {l2b([1, 0]): {l2b([0, 1]): 0.43, l2b([1, 1]): 0.11}, l2b([0, 0]): {l2b(list(master_point)): 0.69}}
    tag = "" if cell_type == "triangle" else "_quad"
    return Case(f"square_dict_p{degree}_m{master_point[0]}{master_point[1]}_{n[0]}x{n[1]}{tag}", V, fem.form_stiffness(V),
                fem.form_source(V, fem.FN_SIN2D), [], dict_constraint_raw(V, s_m_c))


def case_same_cell(degree=1, master_point=(1, 1)) -> Case:
    """python/tests/test_matrix_assembly.py:61-102 (slaves sharing a cell)"""
    return case_square_dict(degree, master_point, n=(1, 8))


def case_lifting() -> Case:
    """python/tests/test_lifting.py:19-122: non-zero Dirichlet value next to a slave"""
    mesh = create_unit_square(1, 1)
    V = fem.functionspace(mesh, ("Lagrange", 1))
    u_bc = fem.Function(V)
    u_bc.x.array[:] = 2.3
    dofs = fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[0], 1))
    bc = fem.dirichletbc(u_bc, dofs, V)
    s_m_c = {l2b([0, 0]): {l2b([0, 1]): 1}}
    return Case("lifting_1x1", V, fem.form_stiffness(V), fem.form_source(V, fem.FN_SIN2D), [bc],
                dict_constraint_raw(V, s_m_c))


def case_pipeline(master_point=(1, 1)) -> Case:
    """python/tests/test_mpc_pipeline.py:26-112: coefficients and constants in the forms"""
    mesh = create_unit_square(3, 5)
    V = fem.functionspace(mesh, ("Lagrange", 1))
    g = fem.Function(V)
    g.interpolate(lambda x: np.sin(x[0]) * x[1])
    h = fem.Function(V)
    h.interpolate(lambda x: 2 + x[1] * x[0])
    a = fem.form_stiffness(V, constant=1.5, coefficient=g)
    L = fem.form_source(V, fem.FN_SIN2D, constant=2.0, coefficient=h)
    s_m_c = {l2b([1, 0]): {l2b([0, 1]): 0.43, l2b([1, 1]): 0.11}, l2b([0, 0]): {l2b(list(master_point)): 0.69}}
    return Case(f"pipeline_m{master_point[0]}{master_point[1]}", V, a, L, [], dict_constraint_raw(V, s_m_c))


def case_vector_poisson(slave_space=0, master_space=1, n=(4, 2)) -> Case:
    """python/tests/test_vector_poisson.py:25-133: blocked space, sub-space constraint, bc"""
    mesh = create_unit_square(*n)
    V = fem.functionspace(mesh, ("Lagrange", 1, (2,)))
    dofs = fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[0], 0) & np.isclose(x[1], 0))
    bc = fem.dirichletbc(0.0, dofs, V)
    s_m_c = {l2b([1, 0]): {l2b([1, 1]): 0.1, l2b([0.5, 1]): 0.3}}
    return Case(f"vector_poisson_s{slave_space}m{master_space}", V, fem.form_stiffness(V),
                fem.form_source(V, fem.FN_LINEAR), [bc], dict_constraint_raw(V, s_m_c, slave_space, master_space))


def case_surface_integral(N=4) -> Case:
    """python/tests/test_surface_integral.py:26-144: elasticity + traction on ds(top)"""
    mesh = create_unit_square(N, N)
    V = fem.functionspace(mesh, ("Lagrange", 1, (2,)))
    dofs = fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[0], 0))
    bc = fem.dirichletbc(0.0, dofs, V)
    E, nu = 1.0e2, 0.0
    mu, lmbda = E / (2.0 * (1.0 + nu)), E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu))
    top = mesh.locate_exterior_facets(lambda x: np.isclose(x[1], 1))
    a = fem.form_elasticity(V, mu, lmbda)
    L = fem.form_facet_source(V, top, fem.FN_CONSTANT_VEC, constant=[1.0, 0.0, -9.81e2])
    s_m_c = {l2b([1, i / N]): {l2b([1, 1]): 0.8} for i in range(1, N)}
    return Case(f"surface_integral_{N}", V, a, L, [bc], dict_constraint_raw(V, s_m_c, 1, 1))


def case_integration_domains() -> Case:
    """python/tests/test_integration_domains.py:23-133: several cell sub-domains"""
    mesh = create_unit_square(15, 5)
    V = fem.functionspace(mesh, ("Lagrange", 1))
    mid = mesh.geometry.x[mesh.geometry.dofmap].mean(axis=1)
    left = np.flatnonzero(mid[:, 0] < 0.5).astype(np.int32)
    right = np.flatnonzero(mid[:, 0] >= 0.5).astype(np.int32)
    a = fem.form_stiffness(V, constant=1.0, cells=left) + fem.form_stiffness(V, constant=2.0, cells=right) \
        + fem.form_mass(V, constant=0.3)
    L = fem.form_source(V, fem.FN_SIN2D, constant=1.0, cells=left) + fem.form_source(V, fem.FN_ONE, constant=2.0, cells=right)
    dofs = fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[1], 0))
    bc = fem.dirichletbc(0.0, dofs, V)
    # periodic pairs x=1 -> x=0 (test uses N+1 pairs via a dict)
    return Case("integration_domains", V, a, L, [bc], periodic_raw(V, [bc]))


def case_facet_mass() -> Case:
    """exterior-facet integrals in A (python/tests/test_surface_integral.py:148-219 style Robin term)"""
    mesh = create_unit_square(4, 4)
    V = fem.functionspace(mesh, ("Lagrange", 1))
    right = mesh.locate_exterior_facets(lambda x: np.isclose(x[0], 1))
    a = fem.form_stiffness(V) + fem.form_facet_mass(V, right, constant=3.0)
    L = fem.form_source(V, fem.FN_SIN2D) + fem.form_facet_source(V, right, fem.FN_LINEAR, constant=0.5)
    s_m_c = {l2b([1, 0.5]): {l2b([0, 0.5]): 0.7, l2b([0, 0.75]): 0.2}, l2b([1, 0.25]): {l2b([1, 0.75]): -0.4}}
    return Case("facet_mass", V, a, L, [], dict_constraint_raw(V, s_m_c))


def case_cube_elasticity_slip(N=3, numbering=None, cell_type="tetrahedron", warp=False) -> Case:
    """vector P1 tets (or Q1 hexahedra), slip constraint u.n = 0 on x=1 with a tilted normal
    (cpp/SlipConstraint.h:115-166 output shape: 1 slave + bs-1 same-block masters)"""
    mesh = create_unit_cube(N, N, N, cell_type)
    if warp:
        mesh = warped(mesh)
    if numbering is not None:
        mesh = renumbered(mesh, numbering)
    V = fem.functionspace(mesh, ("Lagrange", 1, (3,)))
    x = V.tabulate_dof_coordinates()
    dofs = fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[0], 0))
    bc = fem.dirichletbc(np.array([0.0, 0.1, -0.2]), dofs, V)
    nrm = np.array([1.0, 0.3, -0.2])
    nrm /= np.linalg.norm(nrm)
    blocks = np.flatnonzero(np.isclose(x[:, 0], 1.0))
    slaves, masters, coeffs, offsets = [], [], [], [0]
    for b in blocks:
        s = int(np.argmax(np.abs(nrm)))
        slaves.append(b * 3 + s)
        for k in range(3):
            if k != s:
                masters.append(b * 3 + k)
                coeffs.append(-nrm[k] / nrm[s])
        offsets.append(len(masters))
    raw = (np.array(slaves, dtype=np.int32), np.array(masters, dtype=np.int64), np.array(coeffs),
           np.zeros(len(masters), dtype=np.int32), np.array(offsets, dtype=np.int32))
    a = fem.form_elasticity(V, 1.0e3 / 2, 0.0)  # bench_contact_3D.py:257-269: E=1e3, nu=0
    L = fem.form_source(V, fem.FN_LINEAR)
    tag = ("" if numbering is None else "_" + numbering) + ("" if cell_type == "tetrahedron" else "_hex") + ("_warped" if warp else "")
    return Case(f"cube_elasticity_slip_n{N}{tag}", V, a, L, [bc], raw)


def case_cube_contact_like(N=3) -> Case:
    """several masters per slave (contact-like, cpp/ContactConstraint.h:908-1174 output shape):
    every interior node of x=1 is tied to three nodes of x=0 with barycentric-like weights"""
    mesh = create_unit_cube(N, N, N)
    V = fem.functionspace(mesh, ("Lagrange", 1))
    x = V.tabulate_dof_coordinates()
    tree = cKDTree(x)
    h = 1.0 / N
    slaves, masters, coeffs, offsets = [], [], [], [0]
    for d in np.flatnonzero(np.isclose(x[:, 0], 1.0)):
        y, z = x[d, 1], x[d, 2]
        if y < h / 2 or y > 1 - h / 2 or z < h / 2 or z > 1 - h / 2:
            continue
        slaves.append(d)
        for (dy, dz, w) in ((0, 0, 0.5), (-h, 0, 0.3), (0, h, 0.2)):
            _, m = tree.query([0.0, y + dy, z + dz])
            masters.append(int(m))
            coeffs.append(w)
        offsets.append(len(masters))
    raw = (np.array(slaves, dtype=np.int32), np.array(masters, dtype=np.int64), np.array(coeffs),
           np.zeros(len(masters), dtype=np.int32), np.array(offsets, dtype=np.int32))
    # Dirichlet dofs may be neither slaves nor masters: masters have z >= h
    dofs = fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[2], 0))
    bc = fem.dirichletbc(1.7, dofs, V)
    return Case(f"cube_contact_like_n{N}", V, fem.form_stiffness(V) + fem.form_mass(V, constant=0.1),
                fem.form_source(V, fem.FN_POLY3), [bc], raw)


def case_lifting_x0_scale_diagval() -> Case:
    """apply_lifting with x0 and scale != 1 (cpp/lifting.h:292-298: be -= Ae[:, j] * scale * (g_j - x0_j)),
    as a Newton step would call it (python/src/dolfinx_mpc/problem.py:265), and diagval != 1."""
    case = case_cube_periodic(3, 1, 1.3)
    rng = np.random.default_rng(42)
    case.x0 = rng.standard_normal(case.V.num_dofs)
    case.scale = -0.75
    case.diagval = 2.5
    case.name = "lifting_x0_scale_diagval"
    return case


def case_cube_single_master(N=12, master_point=(0.5, 0.5, 0.5)) -> Case:
    """Every free node of the face x = 1 is a slave of ONE master node (rigid-link style, like the
    dictcondition tests with a shared master): the master's row collects the columns of all slave
    cells -- more than 128 column blocks: the device pattern builder falls back to the host one."""
    mesh = create_unit_cube(N, N, N)
    V = fem.functionspace(mesh, ("Lagrange", 1))
    x = V.tabulate_dof_coordinates()
    bc = fem.dirichletbc(0.5, fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[0], 0)), V)
    slaves = np.flatnonzero(np.isclose(x[:, 0], 1.0)).astype(np.int32)
    master = int(np.argmin(np.linalg.norm(x - np.array(master_point), axis=1)))
    n = slaves.size
    raw = (slaves, np.full(n, master, dtype=np.int64), np.linspace(0.2, 1.3, n), np.zeros(n, dtype=np.int32),
           np.arange(n + 1, dtype=np.int32))
    return Case(f"cube_single_master_n{N}", V, fem.form_stiffness(V), fem.form_source(V, fem.FN_POLY3), [bc], raw)


def case_p2_vector_elasticity_slip(n=2) -> Case:
    """dense P2^3 forms: the `2 mu eps(u):eps(v)` velocity block of python/demos/demo_stokes.py (30 x 30 local tensor,
    every component coupled) with the slip constraint and boundary data of stokes_slip_problem"""
    V, Q, bcs, raw_v, forms, L0 = stokes_slip_problem(3, n)
    return Case(f"p2_vector_elasticity_slip_n{n}", V, fem.form_elasticity(V, 1.0, 0.0), L0, bcs, raw_v)


# --------------------------------------------------------------------------
# Irregular meshes: Delaunay triangulations of jittered points -- variable valence, no cell
# clusters, no numbering locality, point sets of opposite faces that do not match -- the kind of mesh gmsh hands the
# reference (python/tests/test_cube_contact.py:15-160), with the constraints built by the library's own builders.
# --------------------------------------------------------------------------
def _built_raw(m):
    return (m._slaves.copy(), m._masters.copy(), m._coeffs.copy(), m._owners.copy(), m._offsets.copy())


def _facet_tags_by_marker(mesh, value, marker):
    from dolfinx_mpc_amd.mesh import MeshTags

    f = mesh.locate_exterior_facets(marker)
    return MeshTags(mesh, mesh.tdim - 1, f, np.full(f.shape[0], value, dtype=np.int32))


def case_delaunay_periodic(dim=3, degree=1, n=4, seed=0, scale=1.0, bc_value=0.0, spatial=False) -> Case:
    """periodic Poisson with NON-MATCHING faces: every dof on x = 1 is tied to the dofs of the cell of the face x = 0 its
    image lies in, weighted by their basis values there (python/src/dolfinx_mpc/multipointconstraint.py:225-300 ->
    cpp/PeriodicConstraint.h): several masters per slave, masters shared between slaves, fat master rows"""
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd.mesh import create_delaunay_box, reorder_spatial

    mesh = create_delaunay_box((0.0,) * dim, (1.0,) * dim, (n,) * dim, seed)
    if spatial:
        mesh = reorder_spatial(mesh, tile_nodes=32)
    V = fem.functionspace(mesh, ("Lagrange", degree))

    def walls(x):
        w = np.isclose(x[1], 0) | np.isclose(x[1], 1)
        return (w | np.isclose(x[2], 0) | np.isclose(x[2], 1)) if dim == 3 else w

    bc = fem.dirichletbc(bc_value, fem.locate_dofs_geometrical(V, walls), V)

    def relation(x):
        out = x.copy()
        out[0] = x[0] - 1.0
        return out

    m = dm.MultiPointConstraint(V)
    m.create_periodic_constraint_geometrical(V, lambda x: np.isclose(x[0], 1.0), relation, [bc], scale)
    raw = _built_raw(m)
    assert raw[0].size > 0 and raw[1].size > raw[0].size  # non-matching: more than one master per slave
    L = fem.form_source(V, fem.FN_BENCH_PERIODIC if dim == 3 else fem.FN_SIN2D)
    tag = "_spatial" if spatial else ""
    return Case(f"delaunay_periodic_{dim}d_p{degree}_n{n}{tag}", V, fem.form_stiffness(V), L, [bc], raw)


def case_delaunay_elasticity_slip(dim=3, n=3, seed=3) -> Case:
    """vector P1 elasticity on an irregular mesh with a slip wall built by create_slip_constraint from facet tags and the
    approximated facet normal (cpp/SlipConstraint.h:16-175), clamped on x = 0"""
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd.mesh import create_delaunay_box
    from dolfinx_mpc_amd.multipointconstraint import create_normal_approximation

    mesh = create_delaunay_box((0.0,) * dim, (1.0,) * dim, (n,) * dim, seed)
    # tilt the box so that the wall normal has every component
    R = rotation_matrix([1.0, 2.0, -0.5], 0.4) if dim == 3 else np.array([[np.cos(0.3), -np.sin(0.3), 0], [np.sin(0.3), np.cos(0.3), 0], [0, 0, 1.0]])
    mesh.geometry.x = mesh.geometry.x @ R.T
    V = fem.functionspace(mesh, ("Lagrange", 1, (dim,)))
    mt = _facet_tags_by_marker(mesh, 5, lambda x: np.isclose((R.T @ x)[0], 1.0))
    xd = V.tabulate_dof_coordinates() @ R
    val = np.array([0.0, 0.1, -0.05])[:dim]
    bcs = [fem.dirichletbc(val, np.flatnonzero(np.isclose(xd[:, 0], 0.0)).astype(np.int32), V)]
    m = dm.MultiPointConstraint(V)
    m.create_slip_constraint(V, (mt, 5), create_normal_approximation(V, mt, 5), bcs)
    raw = _built_raw(m)
    assert raw[0].size > 0
    return Case(f"delaunay_elasticity_slip_{dim}d_n{n}", V, fem.form_elasticity(V, 400.0, 250.0), fem.form_source(V, fem.FN_LINEAR),
                bcs, raw)


def case_delaunay_contact(n_top=2, n_bottom=3, seed=7, theta=0.0) -> Case:
    """two irregular bodies with non-matching interface triangulations, inelastic contact built by
    create_contact_inelastic_condition (cpp/ContactConstraint.h:908-1174): up to three masters per slave and component"""
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd.mesh import (CONTACT_BOTTOM, CONTACT_BOTTOM_INTERFACE, CONTACT_TOP, CONTACT_TOP_INTERFACE,
                                      create_stacked_delaunay)

    mesh, ft, _ct = create_stacked_delaunay(n_top, n_bottom, seed, theta)
    V = fem.functionspace(mesh, ("Lagrange", 1, (3,)))
    u0 = fem.Function(V)
    bc_bottom = fem.dirichletbc(u0, fem.locate_dofs_topological(V, 2, ft.find(CONTACT_BOTTOM)), V)
    u_top = fem.Function(V)
    u_top.interpolate(lambda x: np.stack([np.zeros(x.shape[1]), np.zeros(x.shape[1]), np.full(x.shape[1], -0.2)]))
    bc_top = fem.dirichletbc(u_top, fem.locate_dofs_topological(V, 2, ft.find(CONTACT_TOP)), V)
    m = dm.MultiPointConstraint(V)
    m.create_contact_inelastic_condition(ft, CONTACT_BOTTOM_INTERFACE, CONTACT_TOP_INTERFACE)
    raw = _built_raw(m)
    assert raw[0].size > 0
    a = fem.form_elasticity(V, 500.0, 0.0)
    L = fem.form_source(V, fem.FN_CONSTANT_VEC, constant=[1.0, 0.3, -0.2, -1.0])
    return Case(f"delaunay_contact_{n_top}_{n_bottom}", V, a, L, [bc_bottom, bc_top], raw)


def irregular_cases() -> List[Callable[[], Case]]:
    return [
        lambda: case_delaunay_periodic(3, 1, 4),
        lambda: case_delaunay_periodic(3, 2, 3, seed=1, scale=0.7, bc_value=0.4),
        lambda: case_delaunay_periodic(2, 1, 7, seed=2),
        lambda: case_delaunay_periodic(2, 2, 5, seed=4, bc_value=1.5),
        lambda: case_delaunay_periodic(3, 2, 4, seed=5, spatial=True),  # the same kind of mesh after reorder_spatial
        lambda: case_delaunay_elasticity_slip(3, 3),
        lambda: case_delaunay_elasticity_slip(2, 5),
        lambda: case_delaunay_contact(2, 3),
    ]


def element_sweep_cases() -> List[Callable[[], Case]]:
    """the cell / degree sweep of python/tests/test_matrix_assembly.py:23-26, 61-64 and test_vector_assembly.py:22-24 beyond
    what the built-in operators cover: degree 3 on triangles, degree 1-3 on quadrilaterals (both master choices, and the
    slaves-sharing-a-cell mesh), plus Q2 on hexahedra (python/tests/test_stokes_channelflow.py:21-22 uses Q2 velocities)"""
    out = []
    for cell, degs in (("triangle", (3,)), ("quadrilateral", (1, 2, 3))):
        for d in degs:
            out.append(lambda cell=cell, d=d: case_square_dict(d, (1, 1), (5, 3), cell))
            out.append(lambda cell=cell, d=d: case_square_dict(d, (0, 1), (1, 8), cell))
    out.append(case_hex_q2_periodic)
    # the order-3 members of python/tests/test_stokes_channelflow.py:21-23 in 3D: P3 tetrahedra, Q3 hexahedra
    out.append(lambda: case_cube_p3_periodic("tetrahedron", 2))
    out.append(lambda: case_cube_p3_periodic("hexahedron", 2))
    # degree 4 (python/tests/test_multispace_mpc.py:16 sweeps it): triangles / quadrilaterals with the dictionary constraint,
    # P4 tetrahedra (35 dofs per cell, three per shared face), Q4 hexahedra (125 dofs per cell)
    out.append(lambda: case_square_dict(4, (1, 1), (5, 3), "triangle"))
    out.append(lambda: case_square_dict(4, (0, 1), (3, 4), "quadrilateral"))
    out.append(lambda: case_cube_p3_periodic("tetrahedron", 2, degree=4))
    out.append(lambda: case_cube_p3_periodic("hexahedron", 1, degree=4))
    return out


def case_cube_p3_periodic(cell, N=2, degree=3) -> Case:
    """periodic Poisson with degree 3 / 4 on tetrahedra (20 / 35 dofs per cell, one / three per face) / hexahedra (64 / 125 dofs
    per cell, four / nine per face: more dof blocks than a row-block plan lists per entity, so 'auto' takes the per-entity
    kernels)"""
    mesh = create_unit_cube(N, N, N, cell)
    V = fem.functionspace(mesh, ("Lagrange", degree))
    bc = fem.dirichletbc(0.2, fem.locate_dofs_geometrical(V, _walls_yz), V)
    a = fem.form_stiffness(V) + fem.form_mass(V, constant=0.7)
    return Case(f"{cell[:3]}_p{degree}_periodic_n{N}", V, a, fem.form_source(V, fem.FN_POLY3), [bc], periodic_raw(V, [bc]))


def case_hex_q2_periodic(N=3) -> Case:
    """periodic Poisson with Q2 on hexahedra (27 dofs per cell, generated kernels), Dirichlet walls, mass + stiffness"""
    mesh = create_unit_cube(N, N, N, "hexahedron")
    V = fem.functionspace(mesh, ("Lagrange", 2))
    bc = fem.dirichletbc(0.2, fem.locate_dofs_geometrical(V, _walls_yz), V)
    a = fem.form_stiffness(V) + fem.form_mass(V, constant=0.7)
    return Case(f"hex_q2_periodic_n{N}", V, a, fem.form_source(V, fem.FN_POLY3), [bc], periodic_raw(V, [bc]))


def all_small_cases() -> List[Callable[[], Case]]:
    return [
        lambda: case_square_dict(1, (1, 1)),
        lambda: case_square_dict(1, (0, 1)),
        lambda: case_square_dict(2, (1, 1)),
        lambda: case_square_dict(2, (0, 1)),
        lambda: case_same_cell(1, (1, 1)),
        lambda: case_same_cell(2, (0, 1)),
        case_lifting,
        lambda: case_pipeline((1, 1)),
        lambda: case_pipeline((0, 1)),
        lambda: case_vector_poisson(0, 0),
        lambda: case_vector_poisson(0, 1),
        lambda: case_vector_poisson(1, 0),
        lambda: case_surface_integral(4),
        case_integration_domains,
        case_facet_mass,
        lambda: case_cube_periodic(4, 1, 0.0),
        lambda: case_cube_periodic(3, 2, 0.0),
        lambda: case_cube_periodic(4, 1, 2.3),
        lambda: case_cube_periodic(4, 1, 0.0, reorder=(2, 2, 2)),
        lambda: case_cube_periodic(4, 2, 0.0, reorder=(2, 2, 2)),  # P2 with the tile-wise dof numbering
        lambda: case_cube_elasticity_slip(3),
        lambda: case_cube_contact_like(3),
        case_lifting_x0_scale_diagval,
        lambda: case_contact_two_body(2),  # config 4's shape: nested interface grids
        lambda: case_contact_two_body(2, 3, np.pi / 3),  # non-matching grids, rotated: 3 masters per slave
        lambda: case_contact_two_body(4, 6, 0.0, reorder=(2, 2, 2)),  # tiled numbering (row-block hints per body)
        lambda: case_p2_vector_elasticity_slip(2),  # dense P2^3 block of the Stokes demo
    ]


# --------------------------------------------------------------------------
def oracle_mpc(po, case: Case):
    return po.OracleMPC.from_raw(case.V, *case.raw)


def oracle_outputs(po, case: Case, fast=False):
    """Everything the reference pipeline produces up to the solve
    (python/benchmarks/bench_periodic.py:95-109): A, b (assembled), b after lifting."""
    mpc = oracle_mpc(po, case)
    out = {}
    if case.a is not None:
        out["A"] = po.assemble_matrix(case.a, mpc, bcs=case.bcs, diagval=case.diagval, fast=fast)
    if case.L is not None:
        b = po.assemble_vector(case.L, mpc, fast=fast)
        out["b"] = b.copy()
        if case.a is not None and case.bcs:
            x0 = None if case.x0 is None else [case.x0]
            po.apply_lifting(b, [case.a], [case.bcs], mpc, x0=x0, scale=case.scale, fast=fast)
            out["b_lifted"] = b.copy()
    return out


def product_mpc(case: Case):
    import dolfinx_mpc_amd as dm

    mpc = dm.MultiPointConstraint(case.V)
    mpc.add_constraint(case.V, *case.raw)
    mpc.finalize()
    return mpc


def product_outputs(case: Case, algorithm=None):
    """Same pipeline through the public API on the GPU (HIP kernels via the C ABI)."""
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd.la import Vector

    mpc = product_mpc(case)
    out = {}
    if case.a is not None:
        A = dm.assemble_matrix(case.a, mpc, bcs=case.bcs, diagval=case.diagval, algorithm=algorithm)
        out["A"] = A.to_scipy()
    if case.L is not None:
        b = dm.assemble_vector(case.L, mpc, algorithm=algorithm)
        out["b"] = b.numpy().copy()
        if case.a is not None and case.bcs:
            x0 = None
            if case.x0 is not None:
                import torch

                v = Vector(case.V.num_dofs)
                v.array.copy_(torch.from_numpy(case.x0))
                x0 = [v]
            dm.apply_lifting(b, [case.a], [case.bcs], mpc, x0=x0, scale=case.scale)
            out["b_lifted"] = b.numpy().copy()
    return out


def device_rows(A, rows):
    """{row: (cols, vals)} of a few rows of a device CSR (MPCMatrix)"""
    rp = A.d_rowptr
    out = {}
    for r in rows:
        lo, hi = int(rp[r].item()), int(rp[r + 1].item())
        out[int(r)] = (A.d_cols[lo:hi].cpu().numpy(), A.vals[lo:hi].cpu().numpy())
    return out


def assert_sub_box_rows_match_oracle(po, mesh, ncells, V, A, N, corner, nb, wall_y0=False):
    """Oracle comparison at sizes the oracle cannot assemble: the cells of the box of ``nb``^3 cubes at cube
    index ``corner`` of a unit-cube mesh of N^3 cubes are re-meshed on their own, their scalar P2 stiffness matrix is
    assembled by the ORACLE (cpp/assemble_matrix.cpp:417-548 restated; no constraint inside the box, homogeneous Dirichlet
    condition on the wall y = 0 if the box touches it), and every row of a dof strictly inside the box -- its whole cell patch
    lies in the box -- must equal the row of the big device matrix ``A`` (space ``V`` on ``mesh``) entry by entry; dofs are
    matched by their position on the half-step grid.  Returns the number of rows compared."""
    from dolfinx_mpc_amd import fem
    from dolfinx_mpc_amd.mesh import Mesh
    from dolfinx_mpc_amd.workloads import empty_raw

    h = 1.0 / N
    lo = np.array(corner, dtype=np.float64) * h
    hi = lo + nb * h
    xg = mesh.geometry.x
    cells = mesh.geometry.dofmap[:ncells]
    tol = 1e-9
    node_in = np.all((xg >= lo - tol) & (xg <= hi + tol), axis=1)
    sel = np.flatnonzero(node_in[cells].all(axis=1))
    assert sel.size == 6 * nb ** 3
    nodes = np.unique(cells[sel])
    new = -np.ones(xg.shape[0], dtype=np.int64)
    new[nodes] = np.arange(nodes.size)
    sub = Mesh(xg[nodes].copy(), new[cells[sel]].astype(np.int32), "tetrahedron")
    Vs = fem.functionspace(sub, ("Lagrange", V.degree))
    bcs = []
    if wall_y0:
        bcs = [fem.dirichletbc(0.0, fem.locate_dofs_geometrical(Vs, lambda x: np.isclose(x[1], 0.0)), Vs)]
    ref = oracle_outputs(po, Case("sub_box", Vs, fem.form_stiffness(Vs), None, bcs, empty_raw()))["A"].tocsr()
    M = 2 * N + 1

    def key(x):
        g = np.rint(x * (2 * N)).astype(np.int64)
        return (g[:, 2] * M + g[:, 1]) * M + g[:, 0]

    Xs = Vs.tabulate_dof_coordinates()
    Xb = V.tabulate_dof_coordinates()
    inb = np.flatnonzero(np.all((Xb >= lo - tol) & (Xb <= hi + tol), axis=1))
    big_of_key = dict(zip(key(Xb[inb]).tolist(), inb.tolist()))
    big_of_sub = np.array([big_of_key[k] for k in key(Xs).tolist()], dtype=np.int64)
    strictly = np.all((Xs > lo + 0.25 * h) & (Xs < hi - 0.25 * h), axis=1)
    if wall_y0:  # the wall itself is a mesh boundary, not a cut: rows next to it (and on it) are complete
        strictly = ((Xs[:, 0] > lo[0] + 0.25 * h) & (Xs[:, 0] < hi[0] - 0.25 * h) & (Xs[:, 2] > lo[2] + 0.25 * h)
                    & (Xs[:, 2] < hi[2] - 0.25 * h) & (Xs[:, 1] < hi[1] - 0.25 * h))
    rows_s = np.flatnonzero(strictly)
    got = device_rows(A, big_of_sub[rows_s])
    scale = abs(ref).max()
    sub_of_big = {int(bg): s for s, bg in enumerate(big_of_sub)}
    for s in rows_s:
        cols_b, vals_b = got[int(big_of_sub[s])]
        row_ref = ref.getrow(s)
        want = dict(zip(row_ref.indices.tolist(), row_ref.data.tolist()))
        seen = 0
        for c, v in zip(cols_b.tolist(), vals_b.tolist()):
            sc = sub_of_big.get(c)
            if sc is None:
                assert abs(v) <= 1e-12 * scale, "an entry towards a dof outside the box in a row whose patch lies inside"
                continue
            assert abs(v - want.get(sc, 0.0)) <= 1e-12 * scale, (s, sc, v, want.get(sc))
            seen += sc in want
        assert seen == len(want)
    return int(rows_s.size)
