"""``rigid_motions_nullspace`` (python/src/dolfinx_mpc/utils/mpc_utils.py:163-213): orthonormal, and in the kernel of the
unconstrained elasticity operator (oracle-assembled) in 2D and 3D."""

import numpy as np
import pytest

from dolfinx_mpc_amd import fem
from dolfinx_mpc_amd.mesh import create_box, create_rectangle
from dolfinx_mpc_amd.utils import rigid_motions_nullspace


@pytest.mark.parametrize("dim", [2, 3])
def test_rigid_motions_span_the_kernel(oracle, dim):
    from oracle import pyoracle as po

    mesh = create_rectangle((0.0, 0.0), (2.0, 1.0), (4, 3)) if dim == 2 else create_box((0, 0, 0), (2.0, 1.0, 1.5), (3, 2, 2))
    V = fem.functionspace(mesh, ("Lagrange", 1, (dim,)))
    ns = rigid_motions_nullspace(V)
    B = ns.basis()
    assert ns.dim == (3 if dim == 2 else 6) and B.shape == (V.num_dofs, ns.dim)
    assert abs(B.T @ B - np.eye(ns.dim)).max() < 1e-14  # la.is_orthonormal, mpc_utils.py:203
    om = po.OracleMPC.from_raw(V, np.zeros(0, np.int32), np.zeros(0, np.int64), np.zeros(0), np.zeros(0, np.int32), np.zeros(1, np.int32))
    A = po.assemble_matrix(fem.form_elasticity(V, 1.0, 1.25), om)
    assert abs(A @ B).max() < 1e-13 * abs(A).max()
    assert np.linalg.matrix_rank(B) == ns.dim


def test_rigid_motions_needs_a_vector_space():
    from dolfinx_mpc_amd.mesh import create_unit_cube

    with pytest.raises(ValueError):
        rigid_motions_nullspace(fem.functionspace(create_unit_cube(2, 2, 2), ("Lagrange", 1)))
