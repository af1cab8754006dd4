"""Smoothed-aggregation multigrid as a preconditioner for the device CG solver: the role PETSc's GAMG / hypre's
BoomerAMG play in the reference's benchmark solve (python/benchmarks/bench_periodic.py:112-149; SURVEY section 8f
rank 3: "SpMV / CG (+ Jacobi/AMG)").  A caller of the hot path, not part of it: plain torch sparse algebra on the
device (hipSPARSE behind ``torch.sparse``), no custom kernels.

Set-up (once per matrix)
  * rows that hold nothing but their diagonal -- slave dofs and Dirichlet dofs of the constrained matrix
    (cpp/assemble_matrix.cpp:711-724, python/src/dolfinx_mpc/assemble_matrix.py:59-62) -- are left out of the hierarchy:
    the smoother solves them exactly;
  * aggregates: the dofs (nodes of a blocked space) that fall into one cell of a grid of three mesh widths, from the
    dof coordinates -- the size the distance-two aggregates of GAMG have on a structured mesh; coarse levels reuse the
    centroids of their aggregates;
  * tentative prolongator from a near-null space ``B`` (n x k; default: the constants of every component, k = bs --
    the translations; ``near_null`` = the rigid body modes of ``utils.rigid_motions_nullspace`` for elasticity, what
    ``A.setNearNullSpace`` gives GAMG in python/benchmarks/bench_contact_3D.py:287,320): on every aggregate the rows of
    ``B`` are orthonormalised (batched Cholesky-QR of the k x k Gram matrices, eigen-decomposition where an aggregate is
    too small to carry all k modes -- those columns are dropped), ``P_tent = blockdiag(Q_a)``, and the triangular factors
    are the near-null space of the coarse level (k dofs per aggregate), so that ``P_tent B_coarse = B`` exactly on every
    level; smoothed with one damped Jacobi step, ``P = (I - 4 / (3 rho) D^-1 A) P_tent``; Galerkin coarse operator
    ``P^T A P`` (two sparse-sparse products);
  * coarsest level (<= ``coarse_size`` rows): dense Cholesky.
Cycle: V(1, 1) with a degree-3 Chebyshev smoother on [rho / 10, 1.1 rho] of ``D^-1 A`` (rho from a few power
iterations), symmetric, so that the cycle is a valid CG preconditioner."""

from __future__ import annotations

from typing import List, Optional

import numpy as np


def _csr(rowptr, cols, vals, shape):
    import torch

    return torch.sparse_csr_tensor(rowptr, cols, vals, size=shape)


def _power_rho(A, dinv, iters: int = 12) -> float:
    """largest eigenvalue of D^-1 A (estimate from below, padded by the caller)"""
    import torch

    n = A.shape[0]
    g = torch.Generator(device=dinv.device).manual_seed(11)
    v = torch.rand(n, generator=g, device=dinv.device, dtype=torch.float64) - 0.5
    rho = torch.ones((), dtype=torch.float64, device=dinv.device)
    for _ in range(iters):  # (no host round trip inside the iteration)
        v = v / torch.linalg.vector_norm(v)
        w = dinv * (A @ v)
        rho = torch.dot(v, w)
        v = w
    return abs(float(rho))


class _Mat:
    """CSR arrays in the layout of the library's SpMV kernel (include/mpcx.h mpcx_spmv: 64-bit offsets, 32-bit columns);
    ``torch.sparse`` builds the hierarchy, this kernel applies it (hipSPARSE's CSR product through torch took 95 ms for the
    254 M entries of the 256^3 matrix, mpcx_spmv takes 1.3 ms)"""

    def __init__(self, T):
        import torch

        self.shape = tuple(T.shape)
        self.rowptr = T.crow_indices().to(torch.int64).contiguous()
        self.cols = T.col_indices().to(torch.int32).contiguous()
        self.vals = T.values().contiguous()

    def __matmul__(self, x):
        import torch

        from . import _device as D
        from . import _native

        y = torch.empty(self.shape[0], dtype=torch.float64, device=x.device)
        x = x.contiguous()
        _native.check(_native.lib().mpcx_spmv(self.shape[0], self.rowptr.data_ptr(), self.cols.data_ptr(), self.vals.data_ptr(),
                                              x.data_ptr(), y.data_ptr(), D.stream_ptr()), "mpcx_spmv")
        return y


class _Level:
    def __init__(self, A, dinv, rho, active):
        self.A, self.dinv, self.rho, self.active = A, dinv, rho, active
        self.P = self.PT = None
        self.chol = None
        self.n = A.shape[0]
        self.nnz = A.vals.numel()


class SmoothedAggregation:
    """hierarchy for a symmetric positive definite CSR matrix on the device

    Args:
        rowptr, cols, vals: CSR arrays (device tensors; offsets may be int64, columns int32)
        coords: (n / bs, 3) coordinates of the dof blocks (device or host)
        bs: block size of the space (the dofs of a node are aggregated together)
        near_null: (n, k) near-null space (numpy or tensor), e.g. the six rigid body modes; None: the constants of every
            component
    """

    def __init__(self, rowptr, cols, vals, coords, bs: int = 1, coarse_size: int = 3000, max_levels: int = 12,
                 cell_widths: float = 3.0, near_null=None):
        import torch

        dev = vals.device
        n = rowptr.numel() - 1
        nnz = vals.numel()
        it = torch.int32 if nnz < 2 ** 31 - 1 else torch.int64
        A = _csr(rowptr.to(it), cols.to(it), vals, (n, n))
        X = torch.as_tensor(np.asarray(coords) if not torch.is_tensor(coords) else coords, dtype=torch.float64, device=dev)
        self.bs = bs
        if near_null is None:
            B = torch.eye(bs, dtype=torch.float64, device=dev).repeat(n // bs, 1)
        else:
            B = torch.as_tensor(np.asarray(near_null) if not torch.is_tensor(near_null) else near_null, dtype=torch.float64,
                                device=dev).reshape(n, -1).clone()
        self.near_null_dim = k = int(B.shape[1])
        self.levels: List[_Level] = []
        ext = X.max(0).values - X.min(0).values
        flat = ext <= 1e-12 * float(ext.max())  # a planar mesh: the mesh width comes from the directions it extends in
        h = float((torch.prod(torch.where(flat, torch.ones_like(ext), ext)) / max(X.shape[0], 1)) ** (1.0 / max(int((~flat).sum()), 1)))
        while True:
            n = A.shape[0]
            crow, col, val = A.crow_indices(), A.col_indices(), A.values()
            rows = torch.repeat_interleave(torch.arange(n, device=dev), (crow[1:] - crow[:-1]).long())
            diag = torch.zeros(n, dtype=torch.float64, device=dev)
            isd = rows == col
            diag.index_add_(0, rows[isd], val[isd])
            offd = torch.zeros(n, dtype=torch.float64, device=dev)
            offd.index_add_(0, rows[~isd], val[~isd].abs())
            # (a coarse dof whose mode an aggregate could not carry has an empty row: it stays at zero)
            dinv = torch.where(diag != 0, 1.0 / torch.where(diag != 0, diag, torch.ones_like(diag)), torch.zeros_like(diag))
            active = offd > 1e-14 * diag.abs()  # identity rows (slaves, Dirichlet dofs) stay out of the hierarchy
            B = B * active[:, None]
            Am = _Mat(A)
            lvl = _Level(Am, dinv, 1.1 * _power_rho(Am, dinv), active)
            self.levels.append(lvl)
            if n <= coarse_size or len(self.levels) >= max_levels:
                break
            # ---- aggregates from the coordinates of the dof blocks
            nb = n // bs
            act_b = active.view(nb, bs).any(dim=1)
            cell = cell_widths * h
            key3 = torch.floor((X - X.min(0).values) / cell).long()
            ext = key3.max(0).values + 1
            key = (key3[:, 2] * ext[1] + key3[:, 1]) * ext[0] + key3[:, 0]
            key = torch.where(act_b, key, torch.full_like(key, -1))
            uniq, agg = torch.unique(key, return_inverse=True)
            if int(uniq[0]) == -1:
                agg = agg - 1  # inactive blocks: aggregate -1
                nagg = uniq.numel() - 1
            else:
                nagg = uniq.numel()
            if nagg == 0 or nagg * k >= 0.7 * n:
                break
            cnt = torch.zeros(nagg, dtype=torch.float64, device=dev)
            sel = agg >= 0
            cnt.index_add_(0, agg[sel], torch.ones(int(sel.sum()), dtype=torch.float64, device=dev))
            Xc = torch.zeros((nagg, 3), dtype=torch.float64, device=dev)
            Xc.index_add_(0, agg[sel], X[sel])
            Xc /= cnt[:, None]
            # ---- tentative prolongator: Q_a R_a = B restricted to aggregate a (see the module docstring)
            b_idx = torch.nonzero(sel).reshape(-1)
            comp = torch.arange(bs, device=dev)
            prow = (b_idx[:, None] * bs + comp[None, :]).reshape(-1)
            keep = active[prow]  # a masked component of an otherwise active block
            prow = prow[keep]
            pagg = agg[b_idx][:, None].expand(-1, bs).reshape(-1)[keep]
            Br = B[prow]  # (m, k)
            G = torch.zeros((nagg, k, k), dtype=torch.float64, device=dev)
            G.index_add_(0, pagg, Br[:, :, None] * Br[:, None, :])
            eye_k = torch.eye(k, dtype=torch.float64, device=dev)
            if k == 1:  # (scalar problems: the factor of a 1 x 1 Gram matrix is a square root)
                Lc, info = torch.sqrt(G.clamp(min=0.0)), torch.zeros(nagg, dtype=torch.int32, device=dev)
            else:
                Lc, info = torch.linalg.cholesky_ex(G)
            dR = torch.diagonal(Lc, dim1=1, dim2=2)
            bad = (info != 0) | ~torch.isfinite(dR).all(dim=1) | (dR.min(dim=1).values ** 2 <= 1e-10 * dR.max(dim=1).values ** 2)
            R = Lc.transpose(1, 2).contiguous()  # G = R^T R
            R[bad] = eye_k
            W = torch.linalg.solve_triangular(R, eye_k.expand(nagg, k, k), upper=True)  # Q_a = B_a W_a
            if bool(bad.any()):
                lam, U = torch.linalg.eigh(G[bad])
                good = lam > 1e-10 * lam[:, -1:].clamp(min=1e-300)
                sq = torch.sqrt(lam.clamp(min=0.0))
                W[bad] = U * torch.where(good, 1.0 / torch.where(good, sq, torch.ones_like(sq)), torch.zeros_like(sq))[:, None, :]
                R[bad] = (U * torch.where(good, sq, torch.zeros_like(sq))[:, None, :]).transpose(1, 2)
            Qr = torch.einsum("mk,mkc->mc", Br, W[pagg])
            pcol = (pagg[:, None] * k + torch.arange(k, device=dev)[None, :])
            prow = prow[:, None].expand(-1, k)
            nzq = Qr != 0
            prow, pcol, pval = prow[nzq], pcol[nzq], Qr[nzq]
            B = R.reshape(nagg * k, k)  # the coarse near-null space
            del Br, G, W, Qr, Lc
            nc = nagg * k
            Pt = torch.sparse_coo_tensor(torch.stack([prow, pcol]), pval, (n, nc)).coalesce().to_sparse_csr()
            AP = A @ Pt
            omega = 4.0 / (3.0 * lvl.rho)
            # P = Pt - omega D^-1 (A Pt): scale the rows of AP, then add
            ap_rows = torch.repeat_interleave(torch.arange(n, device=dev), (AP.crow_indices()[1:] - AP.crow_indices()[:-1]).long())
            APs = torch.sparse_coo_tensor(torch.stack([ap_rows, AP.col_indices().long()]), -omega * dinv[ap_rows] * AP.values(),
                                          (n, nc))
            P = (Pt.to_sparse_coo() + APs).coalesce()
            # rows of inactive dofs stay empty
            pr = P.indices()[0]
            okr = active[pr]
            P = torch.sparse_coo_tensor(P.indices()[:, okr], P.values()[okr], (n, nc)).coalesce()
            Pc = P.to_sparse_csr()
            PTc = P.transpose(0, 1).coalesce().to_sparse_csr()
            lvl.P, lvl.PT = _Mat(Pc), _Mat(PTc)
            A = PTc @ (A @ Pc)
            del Pc, PTc, P, Pt, AP, APs
            X = Xc
            h = cell
            bs = k  # coarse nodes carry one dof per near-null mode
        last = self.levels[-1]
        nL = last.n
        if nL <= 20000:
            D = A.to_dense()  # (A: the torch tensor of the last level)
            D = 0.5 * (D + D.T)
            dd = torch.diagonal(D)
            dd += (dd == 0).to(D.dtype)  # empty rows (dropped modes): identity
            Lc, info = torch.linalg.cholesky_ex(D)
            last.chol = Lc if int(info) == 0 else None  # not positive definite to rounding: smoother sweeps instead

    # ------------------------------------------------------------------------------------------------
    def _smooth(self, lvl: _Level, x, b, degree: int = 3):
        """Chebyshev iteration for D^-1 A on [rho / 10, 1.1 rho] (x updated in place and returned)"""
        lmax, lmin = lvl.rho, lvl.rho / 10.0
        theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
        sigma = theta / delta
        rho_k = 1.0 / sigma
        r = lvl.dinv * (b - lvl.A @ x)
        d = r / theta
        for _ in range(degree):
            x = x + d
            r = r - lvl.dinv * (lvl.A @ d)
            rho_n = 1.0 / (2.0 * sigma - rho_k)
            d = rho_n * rho_k * d + (2.0 * rho_n / delta) * r
            rho_k = rho_n
        return x

    def vcycle(self, b, level: int = 0):
        import torch

        lvl = self.levels[level]
        if level == len(self.levels) - 1:
            if lvl.chol is not None:
                return torch.cholesky_solve(b[:, None], lvl.chol)[:, 0]
            x = torch.zeros_like(b)
            for _ in range(4):
                x = self._smooth(lvl, x, b)
            return x
        x = self._smooth(lvl, torch.zeros_like(b), b)
        r = b - lvl.A @ x
        xc = self.vcycle(lvl.PT @ r, level + 1)
        x = x + lvl.P @ xc
        # post-smoothing with the same polynomial keeps the cycle symmetric
        return self._smooth(lvl, x, b)

    def operator_complexity(self) -> float:
        return sum(lv.nnz for lv in self.levels) / self.levels[0].nnz

    def sizes(self) -> list:
        return [int(lv.n) for lv in self.levels]


from .krylov import pcg  # noqa: E402,F401  (the solver lives with MINRES in krylov.py)
