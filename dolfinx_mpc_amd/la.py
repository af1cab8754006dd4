"""Device-resident linear algebra containers (stand-ins for the PETSc ``Mat`` /
``Vec`` the reference assembles into, python/src/dolfinx_mpc/mpc.cpp:273-297)."""

from __future__ import annotations

import contextlib

import numpy as np

from . import _native


class Vector:
    """fp64 vector over the local (owned + ghost) dofs, resident in HBM."""

    def __init__(self, n: int, device=None):
        import torch

        self.device = device if device is not None else _native.require_gpu()
        self.array = torch.zeros(n, dtype=torch.float64, device=self.device)

    def set(self, value: float):
        self.array.fill_(value)

    def numpy(self) -> np.ndarray:
        return self.array.detach().cpu().numpy()

    # PETSc-flavoured no-ops so reference-style drivers read the same
    @contextlib.contextmanager
    def localForm(self):
        yield self

    def ghostUpdate(self, addv=None, mode=None):
        """single process: nothing to exchange (bench_periodic.py:108)"""
        return None

    @property
    def size(self) -> int:
        return self.array.numel()


def create_vector(V) -> Vector:
    return Vector(V.num_dofs)


class MPCMatrix:
    """CSR matrix with the MPC sparsity pattern; values live on the GPU."""

    def __init__(self, rowptr: np.ndarray, cols: np.ndarray, ncols: int, device=None):
        import torch

        self.device = device if device is not None else _native.require_gpu()
        self.rowptr = rowptr
        self.cols = cols
        self.shape = (rowptr.size - 1, ncols)
        self.d_rowptr = torch.from_numpy(rowptr).to(self.device)
        self.d_cols = torch.from_numpy(cols).to(self.device)
        self.vals = torch.zeros(cols.size, dtype=torch.float64, device=self.device)
        self._plans = {}

    @property
    def nnz(self) -> int:
        return self.cols.size

    def zeroEntries(self):
        self.vals.zero_()

    def assemble(self):
        """single process: no off-rank rows to ship (assemble_matrix.py:64)"""
        return None

    def to_scipy(self):
        import scipy.sparse

        return scipy.sparse.csr_matrix((self.vals.detach().cpu().numpy(), self.cols, self.rowptr), shape=self.shape)

    def norm(self) -> float:
        import torch

        return float(torch.linalg.vector_norm(self.vals))
