"""``dolfinx_mpc.utils`` names that belong to the constraint builders and the solve
(python/src/dolfinx_mpc/utils/mpc_utils.py): ``create_normal_approximation`` (:422-438, the direction field the slip
constraints take), ``rotation_matrix`` (:35-48), ``rigid_motions_nullspace`` (:163-213, the near-null space of the
elasticity operators for the multigrid preconditioner), ``determine_closest_block`` / ``create_point_to_point_constraint``
(:216-420, the constraint arrays that tie the dofs of two boundary points, python/demos/demo_elasticity_disconnect.py:176-190)."""

import numpy as np

from .la import NullSpace
from .mesh import rotation_matrix
from .multipointconstraint import create_normal_approximation, locate_points

__all__ = ["create_normal_approximation", "rotation_matrix", "locate_points", "rigid_motions_nullspace",
           "determine_closest_block", "create_point_to_point_constraint"]


def rigid_motions_nullspace(V) -> NullSpace:
    """Translations and rotations of a 2D / 3D vector space as an orthonormal set of vectors over its dofs
    (python/src/dolfinx_mpc/utils/mpc_utils.py:163-213): 3 vectors in 2D, 6 in 3D.  The coordinates are taken about
    the centroid of the dofs -- the same span, better conditioned.  Hand it to ``A.setNearNullSpace``."""
    gdim = V.mesh.geometry.dim
    if gdim not in (2, 3) or V.dofmap.bs != gdim:
        raise ValueError("rigid_motions_nullspace: a vector space with as many components as the mesh has dimensions")
    x = V.tabulate_dof_coordinates()
    x = x - x.mean(axis=0)
    nb = x.shape[0]
    dim = 3 if gdim == 2 else 6
    basis = np.zeros((dim, nb, gdim))
    for i in range(gdim):
        basis[i, :, i] = 1.0
    if gdim == 2:
        basis[2, :, 0], basis[2, :, 1] = -x[:, 1], x[:, 0]
    else:
        basis[3, :, 0], basis[3, :, 1] = -x[:, 1], x[:, 0]
        basis[4, :, 0], basis[4, :, 2] = x[:, 2], -x[:, 0]
        basis[5, :, 2], basis[5, :, 1] = x[:, 1], -x[:, 2]
    vecs = basis.reshape(dim, nb * gdim)
    # modified Gram-Schmidt (dolfinx.la.orthonormalize)
    for i in range(dim):
        for j in range(i):
            vecs[i] -= (vecs[j] @ vecs[i]) * vecs[j]
        vecs[i] /= np.linalg.norm(vecs[i])
    return NullSpace(list(vecs))


def determine_closest_block(V, point):
    """(owning process, [dof block]) of the dof block closest to ``point`` among the dofs of the cells that touch the
    boundary (python/src/dolfinx_mpc/utils/mpc_utils.py:216-297; single process: the owner is 0).  The reference first
    finds the closest boundary CELL and then the closest block of that cell; here the closest block of all boundary cells is
    taken, which is the same block whenever that block belongs to the closest cell (a point on or near the boundary)."""
    mesh = V.mesh
    cells = np.unique(mesh.exterior_facets()[:, 0])
    if cells.size == 0:
        return 0, []
    blocks = np.unique(V.dofmap.list[cells].reshape(-1))
    x = V.tabulate_dof_coordinates()[blocks]
    p = np.zeros(3)
    p[: np.size(point)] = np.asarray(point, dtype=np.float64).reshape(-1)
    return 0, [int(blocks[int(np.argmin(np.linalg.norm(x - p[None, :], axis=1)))])]


def create_point_to_point_constraint(V, slave_point, master_point, vector=None):
    """(slaves, masters, coeffs, owners, offsets) for ``MultiPointConstraint.add_constraint`` tying the dof block closest to
    ``slave_point`` to the block closest to ``master_point`` (python/src/dolfinx_mpc/utils/mpc_utils.py:300-420, single
    process).  ``vector`` None: every component of the slave block equals the same component of the master block.
    With a ``vector`` v (one entry per component): ONE slave, the component s of largest |v|, constrained so that
    v . u_slave_block = v . u_master_block:  u_s = sum_{i != s} (-v_i / v_s) u_slave_block[i] + sum_i (v_i / v_s) u_master[i]
    (components with v_i = 0 are left out)."""
    _, sb = determine_closest_block(V, slave_point)
    _, mb = determine_closest_block(V, master_point)
    if not sb or not mb:
        raise RuntimeError("create_point_to_point_constraint: the mesh has no boundary cells")
    bs = V.dofmap.bs
    sblock, mblock = sb[0], mb[0]
    masters_all = np.arange(mblock * bs, mblock * bs + bs, dtype=np.int64)
    if vector is None:
        slaves = np.arange(sblock * bs, sblock * bs + bs, dtype=np.int32)
        masters = masters_all
        coeffs = np.ones(bs, dtype=np.float64)
        offsets = np.arange(0, bs + 1, dtype=np.int32)
    else:
        v = np.asarray(vector, dtype=np.float64).reshape(-1)
        assert v.size == bs, "one vector entry per component of the space"
        zero = np.isclose(v, 0.0)
        s = int(np.argmax(np.abs(v)))
        assert not zero[s], "the vector must not vanish"
        slaves = np.array([sblock * bs + s], dtype=np.int32)
        m, c = [], []
        for i in range(bs):  # the slave block's other components first (mpc_utils.py:332-336), then the master block
            if i != s and not zero[i]:
                m.append(sblock * bs + i)
                c.append(-v[i] / v[s])
        for i in range(bs):
            if not zero[i]:
                m.append(int(masters_all[i]))
                c.append(v[i] / v[s])
        masters, coeffs = np.asarray(m, dtype=np.int64), np.asarray(c, dtype=np.float64)
        offsets = np.array([0, masters.size], dtype=np.int32)
    owners = np.zeros(masters.size, dtype=np.int32)
    return slaves, masters, coeffs, owners, offsets


# ---------------------------------------------------------------------------------------------------------
# The verification toolkit of python/src/dolfinx_mpc/utils/test.py (the reference's demos and tests check their
# constrained systems with it): host-side, scipy, single process.
# ---------------------------------------------------------------------------------------------------------
def log_info(message: str) -> None:
    """``dolfinx_mpc.utils.log_info`` (mpc_utils.py:150-160): a line on the root process"""
    import logging

    logging.getLogger("dolfinx_mpc_amd").info(message)


def gather_PETScMatrix(A, root: int = 0):
    """the assembled matrix as a scipy CSR matrix (utils/test.py:152-175; here: ``MPCMatrix.to_scipy`` or a scipy matrix)"""
    return A.to_scipy() if hasattr(A, "to_scipy") else A.tocsr()


def gather_PETScVector(b, root: int = 0) -> np.ndarray:
    """the assembled vector as a numpy array (utils/test.py:178-193)"""
    if hasattr(b, "numpy"):
        return b.numpy()
    return np.asarray(b.x.array if hasattr(b, "x") else b)


def gather_transformation_matrix(constraint, root: int = 0):
    """K (num_dofs x (num_dofs - num_slaves)) with u = K u_reduced (utils/test.py:67-149): the row of a free dof holds a 1
    in its reduced column, the row of a slave its coefficients in the reduced columns of its masters"""
    import scipy.sparse

    n = constraint.function_space.num_dofs
    slaves = np.asarray(constraint.slaves[: constraint.num_local_slaves], dtype=np.int64)
    is_slave = np.zeros(n, dtype=bool)
    is_slave[slaves] = True
    reduced = np.cumsum(~is_slave) - 1  # column of every free dof
    off, masters = constraint.masters.offsets, np.asarray(constraint.masters.array, dtype=np.int64)
    coeffs = constraint.coefficients()[0]
    free = np.flatnonzero(~is_slave)
    rows, cols, vals = [free], [reduced[free]], [np.ones(free.size, dtype=coeffs.dtype)]
    for s in slaves:
        lo, hi = off[s], off[s + 1]
        if hi > lo:
            rows.append(np.full(hi - lo, s, dtype=np.int64))
            cols.append(reduced[masters[lo:hi]])
            vals.append(coeffs[lo:hi])
    return scipy.sparse.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                                   shape=(n, free.size)).tocsr()


def compare_CSR(A, B, atol=1e-10) -> None:
    """utils/test.py:196-199"""
    assert abs(A - B).max() < atol


def compare_mpc_lhs(A_org, A_mpc, mpc, root: int = 0, atol=5e3 * np.finfo(np.float64).resolution) -> None:
    """``A_mpc`` without its slave rows / columns equals ``K^H A_org K`` (utils/test.py:202-242)"""
    K = gather_transformation_matrix(mpc)
    A0, A1 = gather_PETScMatrix(A_org), gather_PETScMatrix(A_mpc)
    dt = np.complex128 if np.iscomplexobj(K.data) or np.iscomplexobj(A0.data) else np.float64
    K, A0 = K.astype(dt), A0.astype(dt)
    KTAK = K.conj().T @ A0 @ K
    n = mpc.function_space.num_dofs
    free = np.setdiff1d(np.arange(n), np.asarray(mpc.slaves[: mpc.num_local_slaves], dtype=np.int64))
    compare_CSR(KTAK, A1.tocsr()[free, :][:, free], atol=atol)


def compare_mpc_rhs(b_org, b, constraint, root: int = 0) -> None:
    """``b`` is zero at the slaves and equals ``K^H b_org`` elsewhere (utils/test.py:245-265)"""
    K = gather_transformation_matrix(constraint)
    b0, b1 = gather_PETScVector(b_org), gather_PETScVector(b)
    n = constraint.function_space.num_dofs
    slaves = np.asarray(constraint.slaves[: constraint.num_local_slaves], dtype=np.int64)
    free = np.setdiff1d(np.arange(n), slaves)
    assert np.allclose(b1[slaves], 0)
    assert np.allclose(b1[free], K.conj().T @ b0)


__all__ += ["log_info", "gather_PETScMatrix", "gather_PETScVector", "gather_transformation_matrix", "compare_CSR", "compare_mpc_lhs",
            "compare_mpc_rhs"]
