"""``LinearProblem`` with the reference's call shape
(python/src/dolfinx_mpc/problem.py:353-600): assemble the constrained system with the
HIP assemblers and solve it without leaving the GPU -- Jacobi-preconditioned conjugate
gradients on the assembled CSR matrix (include/mpcx.h: ``mpcx_spmv``, ``mpcx_cg_*``) in
place of the reference's PETSc KSP.  Single (non-nest) forms, symmetric positive definite
problems (Poisson, elasticity); SURVEY section 8f rank 3.
"""

from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

from . import _device as D
from . import _native
from .assemble_matrix import assemble_matrix, create_matrix
from .assemble_vector import apply_lifting, assemble_vector, set_bc
from .fem import DirichletBC, Form, Function
from .la import MPCMatrix, Vector, create_vector
from .multipointconstraint import MultiPointConstraint


def spmv(A: MPCMatrix, x: Vector, y: Optional[Vector] = None) -> Vector:
    """y = A x on the device."""
    if y is None:
        y = Vector(A.shape[0])
    if A.is_block_scalar:
        return _spmv_block_scalar(A, x, y)
    rc = _native.lib().mpcx_spmv(A.shape[0], A.d_rowptr.data_ptr(), A.d_cols.data_ptr(), A.vals.data_ptr(),
                                 x.array.data_ptr(), y.array.data_ptr(), D.stream_ptr())
    _native.check(rc, "mpcx_spmv")
    return y


def _spmv_block_scalar(A: MPCMatrix, x: Vector, y: Vector) -> Vector:
    """y = A x straight from block-scalar storage (include/mpcx.h mpcx_spmv_blockscalar): (S (x) I, masked) x plus the
    overlay (master contributions, slave / Dirichlet diagonals) -- a ninth of the value traffic of the scalar CSR"""
    import torch

    A._wait_ready()
    c = A._compact
    L = _native.lib()
    st = D.stream_ptr()
    bs = c["bs"]
    _native.check(L.mpcx_spmv_blockscalar(A.shape[0] // bs, A.d_rowptr.data_ptr(), A.d_cols.data_ptr(), bs, c["svals"].data_ptr(),
                                          c["mask"].data_ptr(), x.array.data_ptr(), y.array.data_ptr(), st), "mpcx_spmv_blockscalar")
    if c["ov_pos"] is not None and c["ov_pos"].numel():
        if c["ov_rc"] is None:  # rows / columns of the overlay positions, once per plan
            rows = (torch.searchsorted(A.d_rowptr, c["ov_pos"], right=True) - 1).to(torch.int32).contiguous()
            c["ov_rc"] = (rows, A.d_cols[c["ov_pos"]].contiguous())
        r, cl = c["ov_rc"]
        _native.check(L.mpcx_spmv_coo_add(r.numel(), r.data_ptr(), cl.data_ptr(), c["ov_val"].data_ptr(), x.array.data_ptr(),
                                          y.array.data_ptr(), st), "mpcx_spmv_coo_add")
    if c["diag_pos"] is not None and c["diag_pos"].numel():
        d = c["diag_dofs"]
        dv = torch.full((d.numel(),), float(c["diagval"]), dtype=torch.float64, device=A.device)
        _native.check(L.mpcx_spmv_coo_add(d.numel(), d.data_ptr(), d.data_ptr(), dv.data_ptr(), x.array.data_ptr(),
                                          y.array.data_ptr(), st), "mpcx_spmv_coo_add")
    return y


def cg(A: MPCMatrix, b: Vector, x: Optional[Vector] = None, rtol: float = 1e-10, atol: float = 0.0,
       max_it: int = 10000, check_every: int = 25):
    """Solve A x = b (A symmetric positive definite) by Jacobi-preconditioned CG, x0 = 0.
    Converged when |r| <= max(rtol |b|, atol); the residual norm is read back every
    ``check_every`` iterations only (the iteration itself has no host round trip).
    Returns (x, info) with info = {"iterations", "residual_norm", "b_norm", "converged"}."""
    import torch

    if A.shape[0] != A.shape[1]:
        raise RuntimeError("cg needs a square matrix")
    n = A.shape[0]
    L = _native.lib()
    st = D.stream_ptr()
    dev = A.vals.device
    if x is None:
        x = Vector(n)
    work = torch.empty((5, n), dtype=torch.float64, device=dev)  # dinv, r, z, p, Ap
    dinv, r, z, p, Ap = (work[i] for i in range(5))
    scal = torch.zeros(8, dtype=torch.float64, device=dev)
    rp, cl, vl = A.d_rowptr.data_ptr(), A.d_cols.data_ptr(), A.vals.data_ptr()
    _native.check(L.mpcx_inverse_diagonal(n, rp, cl, vl, dinv.data_ptr(), st), "mpcx_inverse_diagonal")
    _native.check(L.mpcx_cg_start(n, dinv.data_ptr(), b.array.data_ptr(), x.array.data_ptr(), r.data_ptr(),
                                  z.data_ptr(), p.data_ptr(), scal.data_ptr(), st), "mpcx_cg_start")
    bb = float(scal[6].item())
    tol2 = max(rtol * rtol * bb, atol * atol)
    rr = bb
    k = 0
    converged = rr <= tol2
    while not converged and k < max_it:
        for _ in range(min(check_every, max_it - k)):
            rc = L.mpcx_cg_step(n, rp, cl, vl, dinv.data_ptr(), x.array.data_ptr(), r.data_ptr(), z.data_ptr(),
                                p.data_ptr(), Ap.data_ptr(), scal.data_ptr(), k, st)
            if rc != 0:
                _native.check(rc, "mpcx_cg_step")
            k += 1
        rr = float(scal[4 + (k & 1)].item())
        if not np.isfinite(rr):
            raise RuntimeError("cg: the residual is not finite (matrix not positive definite?)")
        converged = rr <= tol2
    return x, {"iterations": k, "residual_norm": float(np.sqrt(rr)), "b_norm": float(np.sqrt(bb)),
               "converged": bool(converged)}


def multigrid_cg(A: MPCMatrix, b: Vector, V, x: Optional[Vector] = None, rtol: float = 1e-10, atol: float = 0.0,
                 max_it: int = 500, **_unused):
    """Solve A x = b by CG preconditioned with a smoothed-aggregation V-cycle built from A and the dof coordinates of
    ``V`` (dolfinx_mpc_amd/amg.py).  Returns (x, info); info also carries the set-up time, the level sizes and the
    operator complexity."""
    import time

    import torch

    from .amg import SmoothedAggregation, pcg

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mg = SmoothedAggregation(A.d_rowptr, A.d_cols, A.vals, V.tabulate_dof_coordinates(), bs=V.dofmap.bs)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t0
    A0 = mg.levels[0].A
    xt, info = pcg(lambda v: A0 @ v, mg.vcycle, b.array, rtol=rtol, atol=atol, max_it=max_it)
    torch.cuda.synchronize()
    if x is None:
        x = Vector(A.shape[0])
    x.array.copy_(xt)
    info.update(setup_s=t_setup, solve_s=time.perf_counter() - t0 - t_setup, levels=mg.sizes(),
                operator_complexity=mg.operator_complexity(), pc_type="gamg")
    return x, info


class LinearProblem:
    """a(u, v) = L(v) with a multi point constraint
    (python/src/dolfinx_mpc/problem.py:353-600).

    Args:
        a, L: bilinear and linear form
        mpc: the (finalized) multi point constraint
        bcs: Dirichlet conditions
        u: solution function on ``mpc.function_space`` (created if None)
        solver_options: {"rtol", "atol", "max_it", "check_every"} for the CG solver
            (the reference's ``petsc_options`` play this role); ``"pc_type"``: ``"jacobi"`` (default: the fused CG
            kernels of libmpcx) or ``"gamg"`` (smoothed-aggregation multigrid V-cycle, dolfinx_mpc_amd/amg.py -- the
            preconditioner family of the reference's benchmark solve, bench_periodic.py:112-149)
    """

    def __init__(self, a: Form, L: Form, mpc: MultiPointConstraint, bcs: Optional[Sequence[DirichletBC]] = None,
                 u: Optional[Function] = None, solver_options: Optional[dict] = None):
        if not isinstance(mpc, MultiPointConstraint):
            raise NotImplementedError("LinearProblem: nest / blocked systems are not supported")
        mpc._not_finalized()
        self._a, self._L, self._mpc = a, L, mpc
        self.bcs = [] if bcs is None else list(bcs)
        if u is None:
            u = Function(mpc.function_space)
        elif u.function_space is not mpc.function_space:
            # python/src/dolfinx_mpc/problem.py:464-467
            raise ValueError("The input function has to be in the function space in the multi-point constraint")
        self.u = u
        self._A = create_matrix(a, mpc)
        self._b = create_vector(mpc.function_space)
        self._x = Vector(mpc.function_space.num_dofs)
        self.solver_options = dict(solver_options or {})
        self.info: dict = {}

    @property
    def A(self) -> MPCMatrix:
        return self._A

    @property
    def b(self) -> Vector:
        return self._b

    def assemble(self):
        """A, b with lifting and boundary values, as ``solve`` does before the linear solve
        (problem.py:537-585)."""
        assemble_matrix(self._a, self._mpc, bcs=self.bcs, A=self._A)
        assemble_vector(self._L, self._mpc, b=self._b)
        apply_lifting(self._b, [self._a], [self.bcs], self._mpc)
        set_bc(self._b, self.bcs)
        return self._A, self._b

    def _solve_multigrid(self, **opts):
        _, info = multigrid_cg(self._A, self._b, self._mpc.function_space, x=self._x, **opts)
        return info

    def solve(self) -> Function:
        """Assemble, solve on the device, impose the constraint on the slaves
        (``homogenize`` + ``backsubstitution``, problem.py:589-598) and return ``u``."""
        self.assemble()
        opts = dict(self.solver_options)
        pc = str(opts.pop("pc_type", "jacobi")).lower()
        if pc in ("gamg", "amg"):
            self.info = self._solve_multigrid(**opts)
        elif pc == "jacobi":
            _, self.info = cg(self._A, self._b, x=self._x, **opts)
        else:
            raise NotImplementedError(f"LinearProblem: pc_type {pc!r} (jacobi, gamg)")
        if not self.info["converged"]:
            raise RuntimeError(f"LinearProblem: CG did not converge: {self.info}")
        self._mpc.homogenize(self._x)
        self._mpc.backsubstitution(self._x)
        self.u.x.array[:] = self._x.numpy()
        return self.u
