"""The constant tables of the closed-form cluster kernels against the oracle's element tensors (CPU: the tables are
compile-time data of libmpcx.so, read through mpcx_p2_cluster_tables).  On a parallelepiped cluster -- the six tets of
one sheared, stretched cube -- the sum of the six P2 element tensors must be sum_m M_m K_m with M = c C^T C / |det J|."""
import ctypes as C

import numpy as np

from dolfinx_mpc_amd import _native, fem
from dolfinx_mpc_amd.clusters import fans_from_topology
from dolfinx_mpc_amd.mesh import TET_EDGES, create_unit_cube
from problems import Case, empty_raw, oracle_outputs


def _tables():
    L = _native.lib()
    k = np.zeros((6, 27, 27))
    coupled = np.zeros((27, 27), dtype=np.int32)
    ev = np.zeros((19, 2), dtype=np.int32)
    rs = np.zeros(28, dtype=np.int32)
    assert L.mpcx_p2_cluster_tables(k.ctypes.data, coupled.ctypes.data, ev.ctypes.data, rs.ctypes.data) == 0
    return k, coupled, ev, rs


def test_p2_cluster_tables_shape():
    k, coupled, ev, rs = _tables()
    assert coupled.sum() == 393 and rs[-1] == 393  # six element tensors hold 600 entries
    assert np.array_equal(coupled, coupled.T) and abs(k - k.transpose(0, 2, 1)).max() < 1e-15
    assert (np.count_nonzero(k, axis=0)[coupled == 0] == 0).all()
    # vertex 0 and 7 (the shared edge) and the body diagonal couple to everything; ring vertices sit in two tets
    deg = coupled.sum(axis=1)
    assert deg[0] == deg[7] == 27 and set(deg[1:7]) == {14}
    # constants lie in the kernel of the stiffness operator: every row of every K_m sums to zero
    assert abs(k.sum(axis=2)).max() < 1e-14


def test_p2_cluster_tables_reproduce_the_oracle_on_a_sheared_cluster(oracle):
    k, coupled, ev, rs = _tables()
    mesh = create_unit_cube(1, 1, 1)
    Amat = np.array([[1.3, 0.2, 0.1], [0.05, 0.9, 0.3], [0.2, -0.1, 1.1]])
    mesh.geometry.x = mesh.geometry.x.copy() @ Amat.T + np.array([0.3, -0.2, 0.5])
    V = fem.functionspace(mesh, ("Lagrange", 2))
    c0 = 2.5
    ref = oracle_outputs(oracle, Case("u", V, fem.form_stiffness(V, constant=c0), None, [], empty_raw()))["A"].toarray()
    verts, left = fans_from_topology(mesh.geometry.x, mesh.geometry.dofmap, 6)
    assert verts.shape == (1, 8) and left.size == 0
    v = verts[0]
    cells, dm = mesh.geometry.dofmap, V.dofmap.list
    dofs = np.full(27, -1)
    for c in range(6):
        corner = [int(np.flatnonzero(v == cells[c][i])[0]) for i in range(4)]
        for i in range(4):
            dofs[corner[i]] = dm[c][i]
        for e, (ia, ib) in enumerate(TET_EDGES):
            lo, hi = sorted((corner[ia], corner[ib]))
            idx = int(np.flatnonzero((ev[:, 0] == lo) & (ev[:, 1] == hi))[0])
            dofs[8 + idx] = dm[c][4 + e]
    assert (dofs >= 0).all() and len(set(dofs.tolist())) == 27
    X = mesh.geometry.x
    j = [X[v[1]] - X[v[0]], X[v[2]] - X[v[0]], X[v[4]] - X[v[0]]]
    Cm = [np.cross(j[1], j[2]), np.cross(j[2], j[0]), np.cross(j[0], j[1])]
    s = c0 / abs(j[0] @ Cm[0])
    M = [s * (Cm[d] @ Cm[e]) for d in range(3) for e in range(d, 3)]
    Aloc = sum(M[m] * k[m] for m in range(6))
    got = np.zeros_like(ref)
    got[np.ix_(dofs, dofs)] = Aloc
    assert abs(got - ref).max() <= 1e-13 * abs(ref).max()


def _p1_tables():
    L = _native.lib()
    k6, k9, hex6 = np.zeros((6, 8, 8)), np.zeros((9, 8, 8)), np.zeros((6, 8, 8))
    assert L.mpcx_p1_cluster_tables(k6.ctypes.data, k9.ctypes.data, hex6.ctypes.data) == 0
    return k6, k9, hex6


def _sheared(mesh):
    Amat = np.array([[1.3, 0.2, 0.1], [0.05, 0.9, 0.3], [0.2, -0.1, 1.1]])
    mesh.geometry.x = mesh.geometry.x.copy() @ Amat.T + np.array([0.3, -0.2, 0.5])
    return mesh


def _metric(X, v, c0=1.0):
    j = [X[v[1]] - X[v[0]], X[v[2]] - X[v[0]], X[v[4]] - X[v[0]]]
    Cm = [np.cross(j[1], j[2]), np.cross(j[2], j[0]), np.cross(j[0], j[1])]
    det = j[0] @ Cm[0]
    return Cm, det, [c0 / abs(det) * (Cm[d] @ Cm[e]) for d in range(3) for e in range(d, 3)]


def test_p1_cluster_table_reproduces_the_oracle_on_a_sheared_cluster(oracle):
    """matrix_cube_affine_kernel: A_ij = sum_m M_m K_m(i, j) for the 46 coupled vertex pairs of a parallelepiped cluster"""
    k6, _, _ = _p1_tables()
    mesh = _sheared(create_unit_cube(1, 1, 1))
    V = fem.functionspace(mesh, ("Lagrange", 1))
    ref = oracle_outputs(oracle, Case("u", V, fem.form_stiffness(V, constant=0.7), None, [], empty_raw()))["A"].toarray()
    verts, _ = fans_from_topology(mesh.geometry.x, mesh.geometry.dofmap, 6)
    v = verts[0]
    _, _, M = _metric(mesh.geometry.x, v, 0.7)
    got = np.zeros_like(ref)
    got[np.ix_(v, v)] = sum(M[m] * k6[m] for m in range(6))
    assert abs(got - ref).max() <= 1e-13 * abs(ref).max()


def test_elasticity_cluster_table_reproduces_the_oracle_on_a_sheared_cluster(oracle):
    """matrix_cube_elasticity_rowpair_kernel: Q_ij = C Kp(i, j) C^T / |det J|,
    A[(i,a),(j,b)] = mu Q^{ba} + lambda Q^{ab} + delta_ab mu tr Q"""
    _, k9, _ = _p1_tables()
    mesh = _sheared(create_unit_cube(1, 1, 1))
    V = fem.functionspace(mesh, ("Lagrange", 1, (3,)))
    mu, lmbda = 1.7, 0.6
    ref = oracle_outputs(oracle, Case("u", V, fem.form_elasticity(V, mu, lmbda), None, [], empty_raw()))["A"].toarray()
    verts, _ = fans_from_topology(mesh.geometry.x, mesh.geometry.dofmap, 6)
    v = verts[0]
    Cm, det, _ = _metric(mesh.geometry.x, v)
    Cmat = np.stack(Cm, axis=1)  # columns = cofactor columns: C[r][d]
    got = np.zeros_like(ref)
    for i in range(8):
        for j in range(8):
            Kp = k9[:, i, j].reshape(3, 3)
            Q = Cmat @ Kp @ Cmat.T / abs(det)
            blk = mu * Q.T + lmbda * Q + mu * np.trace(Q) * np.eye(3)
            got[3 * v[i]:3 * v[i] + 3, 3 * v[j]:3 * v[j] + 3] += blk
    assert abs(got - ref).max() <= 1e-13 * abs(ref).max()


def test_hexahedron_closed_form_reproduces_the_oracle_on_a_parallelepiped(oracle):
    """matrix_hex_kernel, closed-form path: the 2 x 2 x 2 Gauss rule is exact on a parallelepiped, so the generated
    kernel (run by the oracle) and sum_m M_m K_m must agree"""
    _, _, hex6 = _p1_tables()
    mesh = _sheared(create_unit_cube(1, 1, 1, "hexahedron"))
    V = fem.functionspace(mesh, ("Lagrange", 1))
    ref = oracle_outputs(oracle, Case("u", V, fem.form_stiffness(V, constant=1.9), None, [], empty_raw()))["A"].toarray()
    v = mesh.geometry.dofmap[0]
    _, _, M = _metric(mesh.geometry.x, v, 1.9)
    got = np.zeros_like(ref)
    got[np.ix_(v, v)] = sum(M[m] * hex6[m] for m in range(6))
    assert abs(got - ref).max() <= 1e-13 * abs(ref).max()
