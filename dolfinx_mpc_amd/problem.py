"""``LinearProblem`` with the reference's call shape
(python/src/dolfinx_mpc/problem.py:353-600): assemble the constrained system with the
HIP assemblers and solve it without leaving the GPU -- Jacobi-preconditioned conjugate
gradients on the assembled CSR matrix (include/mpcx.h: ``mpcx_spmv``, ``mpcx_cg_*``) in
place of the reference's PETSc KSP; symmetric positive definite problems (Poisson, elasticity) by CG with a Jacobi
or smoothed-aggregation preconditioner, nest systems (Stokes: ``a = [[a00, a01], [a10, None]]``, one constraint per
block row, python/src/dolfinx_mpc/problem.py:418-447) by MINRES with an additive field split
(python/tests/test_stokes_channelflow.py:107-125).  SURVEY section 8f rank 3.
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


class _View:
    """a device tensor in the place of a ``Vector`` (sub-vectors of a nest system are views of one tensor)"""

    def __init__(self, t):
        self.array = t


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


class NestOperator:
    """y = [sum_j A_ij x_j]_i on one concatenated device tensor; ``None`` blocks are zero (PETSc ``MatNest``)"""

    def __init__(self, blocks: Sequence[Sequence[Optional[MPCMatrix]]]):
        self.blocks = [list(r) for r in blocks]
        nb = len(self.blocks)
        rows = [next((b.shape[0] for b in r if b is not None), None) for r in self.blocks]
        cols = [next((self.blocks[i][j].shape[1] for i in range(nb) if self.blocks[i][j] is not None), None) for j in range(nb)]
        self.sizes = [r if r is not None else c for r, c in zip(rows, cols)]
        if any(sz is None for sz in self.sizes) or any(c is not None and c != sz for c, sz in zip(cols, self.sizes)):
            raise ValueError("NestOperator: block sizes do not match / an empty block row and column")
        self.offsets = np.concatenate([[0], np.cumsum(self.sizes)]).astype(np.int64)
        self.n = int(self.offsets[-1])

    def split(self, t):
        return [t[self.offsets[i]:self.offsets[i + 1]] for i in range(len(self.sizes))]

    def __call__(self, x):
        import torch

        x = x.contiguous()
        y = torch.zeros_like(x)
        xs, ys = self.split(x), self.split(y)
        tmp = None
        for i, row in enumerate(self.blocks):
            first = True
            for j, blk in enumerate(row):
                if blk is None:
                    continue
                if first:
                    spmv(blk, _View(xs[j]), _View(ys[i]))
                    first = False
                else:
                    if tmp is None or tmp.numel() < self.sizes[i]:
                        tmp = torch.empty(max(self.sizes), dtype=x.dtype, device=x.device)
                    spmv(blk, _View(xs[j]), _View(tmp[: self.sizes[i]]))
                    ys[i].add_(tmp[: self.sizes[i]])
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



def _warn_unused_options(where: str, unused: dict):
    """solver options this backend does not know are reported, not dropped silently (a PETSc-style ``ksp_rtol`` handed
    to a routine that spells it ``rtol`` used to vanish)"""
    if unused:
        import warnings

        warnings.warn(f"dolfinx_mpc_amd.{where}: unrecognised solver options ignored: {sorted(unused)}", RuntimeWarning, stacklevel=3)

def multigrid_cg(A: MPCMatrix, b: Vector, V, x: Optional[Vector] = None, rtol: float = 1e-10, atol: float = 0.0,
                 max_it: int = 500, **_unused):
    """Solve A x = b by CG preconditioned with a smoothed-aggregation V-cycle built from A and the dof coordinates of
    ``V`` (dolfinx_mpc_amd/amg.py).  Returns (x, info); info also carries the set-up time, the level sizes and the
    operator complexity."""
    _warn_unused_options("multigrid_cg", _unused)
    import time

    import torch

    from .amg import SmoothedAggregation, pcg

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ns = getattr(A, "near_nullspace", None)
    mg = SmoothedAggregation(A.d_rowptr, A.d_cols, A.vals, V.tabulate_dof_coordinates(), bs=V.dofmap.bs,
                             near_null=None if ns is None else ns.basis())
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t0
    A0 = mg.levels[0].A
    xt, info = pcg(lambda v: A0 @ v, mg.vcycle, b.array, rtol=rtol, atol=atol, max_it=max_it)
    torch.cuda.synchronize()
    if x is None:
        x = Vector(A.shape[0])
    x.array.copy_(xt)
    info.update(setup_s=t_setup, solve_s=time.perf_counter() - t0 - t_setup, levels=mg.sizes(),
                operator_complexity=mg.operator_complexity(), pc_type="gamg", near_null_dim=mg.near_null_dim)
    return x, info


def _inverse_diagonal(A: MPCMatrix):
    """1 / diag(A) as a device tensor (empty diagonals -> 1: PETSc's Jacobi does the same for the pressure slaves)"""
    import torch

    n = A.shape[0]
    dinv = torch.empty(n, dtype=torch.float64, device=A.device)
    _native.check(_native.lib().mpcx_inverse_diagonal(n, A.d_rowptr.data_ptr(), A.d_cols.data_ptr(), A.vals.data_ptr(),
                                                      dinv.data_ptr(), D.stream_ptr()), "mpcx_inverse_diagonal")
    return torch.where(torch.isfinite(dinv) & (dinv != 0), dinv, torch.ones_like(dinv))


def fieldsplit_minres(A: Sequence[Sequence[Optional[MPCMatrix]]], b: Sequence[Vector], spaces, P=None, pc_types=None,
                      rtol: float = 1e-10, atol: float = 0.0, max_it: int = 2000, check_every: int = 10, **_unused):
    """Solve the symmetric nest system A x = b by MINRES with an additive field split
    (python/tests/test_stokes_channelflow.py:107-125, python/demos/demo_stokes_nest.py:231-252): block i of the
    preconditioner is built from ``P[i][i]`` if a preconditioner matrix is given, else from ``A[i][i]``, else it is the
    identity; ``pc_types[i]``: ``"gamg"`` (one smoothed-aggregation V-cycle, dolfinx_mpc_amd/amg.py), ``"jacobi"`` or
    ``"none"``; default: ``gamg`` for the first block, ``jacobi`` for the others (the reference's demo set-up).
    Returns (list of solution tensors, info)."""
    _warn_unused_options("fieldsplit_minres", _unused)
    import time

    import torch

    from .amg import SmoothedAggregation
    from .krylov import minres

    op = NestOperator(A)
    nb = len(op.sizes)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pcs, kinds = [], []
    for i in range(nb):
        M = None if P is None else P[i][i]
        M = A[i][i] if M is None else M
        kind = (pc_types[i] if pc_types is not None else ("gamg" if i == 0 else "jacobi")).lower()
        if M is None or kind == "none":
            pcs.append(lambda r: r.clone())
            kinds.append("none")
        elif kind in ("gamg", "amg"):
            V = spaces[i]
            ns = getattr(M, "near_nullspace", None)
            mg = SmoothedAggregation(M.d_rowptr, M.d_cols, M.vals, V.tabulate_dof_coordinates(), bs=V.dofmap.bs,
                                     near_null=None if ns is None else ns.basis())
            pcs.append(mg.vcycle)
            kinds.append("gamg" + str(mg.sizes()))
        elif kind == "jacobi":
            dinv = _inverse_diagonal(M)
            pcs.append(lambda r, dinv=dinv: dinv * r)
            kinds.append("jacobi")
        else:
            raise NotImplementedError(f"fieldsplit_minres: pc_type {kind!r} (gamg, jacobi, none)")

    def M_inv(r):
        return torch.cat([pc(ri.contiguous()) for pc, ri in zip(pcs, op.split(r))])

    rhs = torch.cat([bi.array for bi in b])
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t0
    x, info = minres(op, M_inv, rhs, rtol=rtol, atol=atol, max_it=max_it, check_every=check_every)
    torch.cuda.synchronize()
    info.update(setup_s=t_setup, solve_s=time.perf_counter() - t0 - t_setup, ksp_type="minres", pc_type="fieldsplit",
                fieldsplit=kinds)
    return op.split(x), info


class LinearProblem:
    """a(u, v) = L(v) with a multi point constraint
    (python/src/dolfinx_mpc/problem.py:353-600).

    Args:
        a, L: bilinear and linear form; nest systems: ``a`` a list of lists of forms (``None`` = empty block), ``L`` a
            list of forms (``None`` = zero), as in python/src/dolfinx_mpc/problem.py:418-447
        mpc: the (finalized) multi point constraint; nest systems: one per block row
        bcs: Dirichlet conditions
        u: solution function on ``mpc.function_space`` (created if None); nest systems: a list of functions
        solver_options: {"rtol", "atol", "max_it", "check_every"} for the Krylov solver
            (the reference's ``petsc_options`` play this role); ``"pc_type"``: ``"jacobi"`` (default: the fused CG
            kernels of libmpcx) or ``"gamg"`` (smoothed-aggregation multigrid V-cycle, dolfinx_mpc_amd/amg.py -- the
            preconditioner family of the reference's benchmark solve, bench_periodic.py:112-149; it uses
            ``problem.A.near_nullspace`` if ``A.setNearNullSpace`` was called).  Nest systems: MINRES with an additive
            field split (test_stokes_channelflow.py:107-125); ``"fieldsplit_pc_types"``: one of ``gamg`` / ``jacobi`` /
            ``none`` per block (default ``gamg`` for the first, ``jacobi`` for the others)
        P: nest systems: forms of a preconditioner matrix (``[[None, None], [None, mass]]``: blocks that are ``None`` fall
            back to the system's own diagonal block), python/src/dolfinx_mpc/problem.py:470-480,
            python/demos/demo_stokes_nest.py:226-228
    """

    def __init__(self, a, L, mpc, bcs: Optional[Sequence[DirichletBC]] = None, u=None, solver_options: Optional[dict] = None,
                 P=None):
        self._nest = isinstance(mpc, (list, tuple))
        self._a, self._L, self._mpc = a, L, mpc
        self.bcs = [] if bcs is None else list(bcs)
        self.solver_options = dict(solver_options or {})
        self.info: dict = {}
        self._P_forms, self._P = P, None
        if self._nest:
            from .assemble_matrix import create_matrix_nest
            from .assemble_vector import create_vector_nest

            mpcs = list(mpc)
            for m in mpcs:
                if not isinstance(m, MultiPointConstraint):
                    raise TypeError("LinearProblem: a sequence of MultiPointConstraint objects for a nest system")
                m._not_finalized()
            if len(a) != len(mpcs) or len(L) != len(mpcs) or any(len(r) != len(mpcs) for r in a):
                raise ValueError("LinearProblem: a, L and the constraints do not have the same number of blocks")
            if u is None:
                u = [Function(m.function_space) for m in mpcs]
            else:
                u = list(u)
                for ui, m in zip(u, mpcs):
                    if ui.function_space is not m.function_space:
                        raise ValueError("The input function has to be in the function space in the multi-point constraint")
            self.u = u
            self._A = create_matrix_nest(a, mpcs)
            self._b = create_vector_nest(L, mpcs)
            if P is not None:
                self._P = create_matrix_nest(P, mpcs)
            self._x = None
            return
        if not isinstance(mpc, MultiPointConstraint):
            raise TypeError("LinearProblem: a MultiPointConstraint (or a sequence of them for a nest system)")
        mpc._not_finalized()
        if u is None:
            u = Function(mpc.function_space)
        elif u.function_space is not mpc.function_space:
            # python/src/dolfinx_mpc/problem.py:464-467
            raise ValueError("The input function has to be in the function space in the multi-point constraint")
        self.u = u
        self._A = create_matrix(a, mpc)
        self._b = create_vector(mpc.function_space)
        self._x = Vector(mpc.function_space.num_dofs)

    @property
    def A(self):
        return self._A

    @property
    def b(self):
        return self._b

    @property
    def P_mat(self):
        return self._P

    def assemble(self):
        """A, b with lifting and boundary values, as ``solve`` does before the linear solve
        (problem.py:537-585)."""
        if self._nest:
            from .assemble_matrix import assemble_matrix_nest
            from .assemble_vector import assemble_vector_nest

            mpcs = list(self._mpc)
            assemble_matrix_nest(self._A, self._a, mpcs, bcs=self.bcs)
            if self._P is not None:
                assemble_matrix_nest(self._P, self._P_forms, mpcs, bcs=self.bcs)
            assemble_vector_nest(self._b, self._L, mpcs)
            apply_lifting(self._b, self._a, self.bcs, mpcs)
            for bi, m in zip(self._b, mpcs):  # dolfinx bcs_by_block: the conditions that live in the block's space
                set_bc(bi, [bc for bc in self.bcs if m.function_space.contains(bc.function_space)])
            return self._A, self._b
        assemble_matrix(self._a, self._mpc, bcs=self.bcs, A=self._A)
        assemble_vector(self._L, self._mpc, b=self._b)
        apply_lifting(self._b, [self._a], [self.bcs], self._mpc)
        set_bc(self._b, self.bcs)
        return self._A, self._b

    def _solve_multigrid(self, **opts):
        _, info = multigrid_cg(self._A, self._b, self._mpc.function_space, x=self._x, **opts)
        return info

    def _solve_nest(self):
        opts = dict(self.solver_options)
        ksp = str(opts.pop("ksp_type", "minres")).lower()
        pc = str(opts.pop("pc_type", "fieldsplit")).lower()
        if ksp != "minres" or pc != "fieldsplit":
            raise NotImplementedError(f"LinearProblem (nest): ksp_type {ksp!r} / pc_type {pc!r} (minres with fieldsplit)")
        mpcs = list(self._mpc)
        xs, self.info = fieldsplit_minres(self._A, self._b, [m.function_space for m in mpcs], P=self._P,
                                          pc_types=opts.pop("fieldsplit_pc_types", None), **opts)
        if not self.info["converged"]:
            raise RuntimeError(f"LinearProblem: MINRES did not converge: {self.info}")
        for xi, ui, m in zip(xs, self.u, mpcs):
            v = Vector(m.function_space.num_dofs)
            v.array.copy_(xi)
            m.homogenize(v)
            m.backsubstitution(v)
            ui.x.array[:] = v.numpy()
        return self.u

    def solve(self):
        """Assemble, solve on the device, impose the constraint on the slaves
        (``homogenize`` + ``backsubstitution``, problem.py:589-598) and return ``u`` (nest systems: the list)."""
        self.assemble()
        if self._nest:
            return self._solve_nest()
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


class NonlinearProblem:
    """F(u; v) = 0 with a multi point constraint by Newton's method -- the reference's ``NonlinearProblem``
    (python/src/dolfinx_mpc/problem.py:155-352: PETSc SNES ``newtonls`` with its residual / Jacobian callbacks
    ``assemble_residual_mpc`` :88-152 and ``assemble_jacobian_mpc`` :26-85), with the same call order round the hot path:

        u <- x;  mpc.homogenize(u);  mpc.backsubstitution(u)              (the forms read u)
        b  = assemble_vector(F, mpc);  apply_lifting(b, [J], [bcs], mpc, x0=[x], scale=-1);  set_bc(b, bcs, x0=x, scale=-1)
        A  = assemble_matrix(J, mpc, bcs);   solve A dx = b;   x <- x - dx

    Args:
        F, J: residual (rank 1) and Jacobian (rank 2) forms whose coefficient is ``u`` (no symbolic differentiation here:
            the caller -- a form compiler -- supplies J, e.g. ``fem.forms_nonlinear_poisson``)
        u: the unknown, a ``fem.Function`` on ``mpc.function_space``; its values are the initial guess
        mpc, bcs: constraint and Dirichlet conditions
        solver_options: ``snes_rtol`` / ``snes_atol`` (residual norm), ``snes_stol`` (step), ``snes_max_it``; linear solves:
            ``ksp_type`` ``"bicgstab"`` (device, Jacobi-preconditioned; default) or ``"preonly"`` with ``pc_type`` ``"lu"`` --
            a sparse direct solve on the HOST (scipy SuperLU), the analogue of the reference tests' MUMPS LU
            (python/tests/test_nonlinear_assembly.py:82-93) for small systems; ``ksp_rtol``, ``ksp_max_it``.
    """

    def __init__(self, F: Form, u: Function, mpc: MultiPointConstraint, bcs: Optional[Sequence[DirichletBC]] = None,
                 J: Optional[Form] = None, solver_options: Optional[dict] = None):
        if J is None:
            raise ValueError("NonlinearProblem: the Jacobian form J is required (there is no symbolic differentiation here)")
        if not isinstance(mpc, MultiPointConstraint):
            raise NotImplementedError("NonlinearProblem: single (non-nest) systems")
        mpc._not_finalized()
        if u.function_space is not mpc.function_space:
            raise ValueError("The input function has to be in the function space in the multi-point constraint")
        self._F, self._J, self._u, self.mpc = F, J, u, mpc
        self.bcs = [] if bcs is None else list(bcs)
        self._A = create_matrix(J, mpc)
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

    @property
    def x(self) -> Vector:
        return self._x

    def _assign_u(self):
        """u <- x, then the constraint on the slaves (problem.py:101-113).  Skipped when neither x nor u changed since the
        last assignment (the Jacobian callback right after the residual callback of the same Newton iterate: ADVICE r5)"""
        import torch

        seen = getattr(self, "_assigned", None)
        if seen is not None and torch.equal(seen[0], self._x.array) and np.array_equal(seen[1], self._u.x.array):
            return
        self._u.x.array[:] = self._x.numpy()
        self.mpc.homogenize(self._u)
        self.mpc.backsubstitution(self._u)
        self._assigned = (self._x.array.clone(), np.array(self._u.x.array, copy=True))

    def assemble_residual(self) -> Vector:
        """problem.py:88-152"""
        self._assign_u()
        assemble_vector(self._F, self.mpc, b=self._b)
        apply_lifting(self._b, [self._J], [self.bcs], self.mpc, x0=[self._x], scale=-1.0)
        set_bc(self._b, self.bcs, x0=self._x, scale=-1.0)
        return self._b

    def assemble_jacobian(self) -> MPCMatrix:
        """problem.py:26-85: the reference's callback copies x into u and imposes the constraint before it assembles
        (problem.py:60-72), so that a call on its own after changing ``problem.x`` sees the current iterate"""
        self._assign_u()
        assemble_matrix(self._J, self.mpc, bcs=self.bcs, diagval=1.0, A=self._A)
        return self._A

    def _linear_solve(self, opts) -> "object":
        import torch

        from .krylov import bicgstab

        ksp = str(opts.get("ksp_type", "bicgstab")).lower()
        pc = str(opts.get("pc_type", "jacobi")).lower()
        A, b = self._A, self._b
        if ksp == "preonly" and pc == "lu":
            import scipy.sparse.linalg as spla

            dx = spla.splu(A.to_scipy().tocsc()).solve(b.numpy())
            self.info["linear"] = {"ksp_type": "preonly", "pc_type": "lu (host, scipy SuperLU)"}
            return torch.from_numpy(dx).to(b.array.device)
        if ksp != "bicgstab" or pc != "jacobi":
            raise NotImplementedError(f"NonlinearProblem: ksp_type {ksp!r} / pc_type {pc!r} (bicgstab + jacobi, preonly + lu)")
        dinv = _inverse_diagonal(A)
        Av = _View(None)

        def mv(v):
            y = torch.empty_like(v)
            Av.array = v.contiguous()
            spmv(A, Av, _View(y))
            return y

        dx, li = bicgstab(mv, lambda r: dinv * r, b.array, rtol=float(opts.get("ksp_rtol", 1e-12)),
                          max_it=int(opts.get("ksp_max_it", 5000)))
        if not li["converged"]:
            raise RuntimeError(f"NonlinearProblem: the linear solve did not converge: {li}")
        self.info.setdefault("linear_iterations", []).append(li["iterations"])
        return dx

    def solve(self):
        """Newton iteration without line search (``snes_linesearch_type none``); returns (u, converged_reason,
        iterations) with PETSc's reason codes: 2 = |F| < atol, 3 = |F| < rtol |F0|, 4 = |dx| < stol |x|, -5 = max_it."""
        import torch

        opts = self.solver_options
        rtol, atol = float(opts.get("snes_rtol", 1e-8)), float(opts.get("snes_atol", 1e-50))
        stol, max_it = float(opts.get("snes_stol", 1e-8)), int(opts.get("snes_max_it", 50))
        self._x.array.copy_(torch.from_numpy(np.ascontiguousarray(self._u.x.array, dtype=np.float64)))
        self.info = {"residual_norms": []}
        reason, it = 0, 0
        b = self.assemble_residual()
        f0 = fn = float(torch.linalg.vector_norm(b.array))
        self.info["residual_norms"].append(fn)
        if not np.isfinite(fn):
            reason = -4  # SNES_DIVERGED_FNORM_NAN
        elif fn < atol:
            reason = 2
        elif max_it <= 0:
            reason = -5
        while reason == 0:
            self.assemble_jacobian()
            dx = self._linear_solve(opts)
            self._x.array.sub_(dx)
            it += 1
            # PETSc's SNESConvergedDefault order: the NEW residual first (atol, then rtol), the step size last
            dxn, xn = float(torch.linalg.vector_norm(dx)), float(torch.linalg.vector_norm(self._x.array))
            b = self.assemble_residual()
            fn = float(torch.linalg.vector_norm(b.array))
            self.info["residual_norms"].append(fn)
            if not np.isfinite(fn):
                reason = -4
                break
            if fn < atol:
                reason = 2
                break
            if fn <= rtol * f0:
                reason = 3
                break
            if dxn < stol * xn:
                reason = 4
                break
            if it >= max_it:
                reason = -5
                break
        self._assign_u()
        self.info.update(iterations=it, converged_reason=reason)
        return self._u, reason, it
