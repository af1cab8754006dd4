"""dolfinx_mpc_amd -- MI355X-native constrained finite-element assembly.

Drop-in for the hot path of dolfinx_mpc (``assemble_matrix``,
``assemble_vector``, ``apply_lifting`` and the ``MultiPointConstraint`` data
they read; python/src/dolfinx_mpc/__init__.py:11-25 exports the same names).
The callers either side of it -- constraint builders, ``LinearProblem`` / ``NonlinearProblem``, ``utils`` -- mirror the
reference's names too; I/O and PETSc itself are out of scope, see DESIGN.md.
"""

from .assemble_matrix import (
    assemble_matrix,
    assemble_matrix_nest,
    create_matrix,
    create_matrix_nest,
    create_sparsity_pattern,
)
from .assemble_vector import (
    apply_lifting,
    assemble_vector,
    assemble_vector_nest,
    create_vector_nest,
    set_bc,
)
from .multipointconstraint import MPCData, MultiPointConstraint
from . import common  # noqa: F401  (Timer, list_timings: the reference's "~MPC: ..." scopes, also roctx ranges)
from . import utils  # noqa: F401  (dolfinx_mpc.utils: constraint helpers, near-null space, the verification toolkit)
from .problem import LinearProblem, NonlinearProblem
from . import _native as _native_mod

_native_mod.start_preload()  # (a no-op without a device or with MPCX_PRELOAD=0)

__all__ = [
    "assemble_matrix",
    "create_matrix_nest",
    "assemble_matrix_nest",
    "assemble_vector",
    "apply_lifting",
    "assemble_vector_nest",
    "create_vector_nest",
    "MultiPointConstraint",
    "MPCData",
    "create_sparsity_pattern",
    "create_matrix",
    "set_bc",
    "LinearProblem",
    "NonlinearProblem",
    "utils",
    "common",
]
