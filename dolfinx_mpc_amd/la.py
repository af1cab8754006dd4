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
    """CSR matrix with the MPC sparsity pattern; pattern and values live on the GPU.
    ``rowptr`` is 64-bit (include/mpcx.h ``mpcx_nnz_t``): a matrix may hold more than 2^31 - 1
    entries on one GPU; column indices are 32-bit.  ``rowptr`` / ``cols`` may be given as numpy
    arrays or as device tensors (the device pattern builder hands over tensors, so a 17 GB column
    array never visits the host unless somebody asks for ``A.cols``)."""

    def __init__(self, rowptr, cols, ncols: int, device=None):
        import torch

        self.device = device if device is not None else _native.require_gpu()
        if isinstance(rowptr, torch.Tensor):
            self.d_rowptr = rowptr.to(self.device, torch.int64)
            self._rowptr = None
        else:
            self._rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
            self.d_rowptr = torch.from_numpy(self._rowptr).to(self.device)
        if isinstance(cols, torch.Tensor):
            self.d_cols = cols.to(self.device, torch.int32)
            self._cols = None
        else:
            self._cols = np.ascontiguousarray(cols, dtype=np.int32)
            self.d_cols = torch.from_numpy(self._cols).to(self.device)
        self.shape = (self.d_rowptr.numel() - 1, ncols)
        self.vals = torch.zeros(self.d_cols.numel(), dtype=torch.float64, device=self.device)
        self._plans = {}

    @property
    def rowptr(self) -> np.ndarray:
        """host copy of the row offsets (int64), downloaded on first use"""
        if self._rowptr is None:
            self._rowptr = self.d_rowptr.cpu().numpy()
        return self._rowptr

    @property
    def cols(self) -> np.ndarray:
        """host copy of the column indices (int32), downloaded on first use"""
        if self._cols is None:
            self._cols = self.d_cols.cpu().numpy()
        return self._cols

    @property
    def nnz(self) -> int:
        return self.d_cols.numel()

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
