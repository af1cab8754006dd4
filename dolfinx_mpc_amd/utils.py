"""``dolfinx_mpc.utils`` names that belong to the constraint builders and the solve
(python/src/dolfinx_mpc/utils/mpc_utils.py): ``create_normal_approximation`` (:422-438, the direction field the slip
constraints take), ``rotation_matrix`` (:35-48), ``rigid_motions_nullspace`` (:163-213, the near-null space of the
elasticity operators for the multigrid preconditioner)."""

import numpy as np

from .la import NullSpace
from .mesh import rotation_matrix
from .multipointconstraint import create_normal_approximation, locate_points

__all__ = ["create_normal_approximation", "rotation_matrix", "locate_points", "rigid_motions_nullspace"]


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
