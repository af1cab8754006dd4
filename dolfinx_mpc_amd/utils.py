"""``dolfinx_mpc.utils`` names that belong to the constraint builders (python/src/dolfinx_mpc/utils/mpc_utils.py):
``create_normal_approximation`` (:422-438, the direction field the slip constraints take) and ``rotation_matrix``
(:35-48)."""

from .mesh import rotation_matrix
from .multipointconstraint import create_normal_approximation, locate_points

__all__ = ["create_normal_approximation", "rotation_matrix", "locate_points"]
