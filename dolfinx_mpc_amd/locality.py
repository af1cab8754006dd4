"""Automatic locality: a mesh whose numbering carries no locality (a mesh read from a file, a Delaunay mesh, shuffled
nodes) is assembled on an internal, spatially reordered TWIN of the problem, and every result is handed back in the
caller's numbering -- caller numbering in, caller numbering out, no caller action (until round 4
the caller had to call ``mesh.reorder_spatial`` and rebuild everything, and got a warning otherwise).

Why a twin and not an indirection inside the kernels: the row-block kernels keep CONTIGUOUS CSR row ranges in LDS and
gather coordinates / dofmap rows of the entities that touch them; without locality every gather misses L2 (a 24-byte
coordinate costs a 64-128-byte line) and every entity touches as many blocks as it has dofs.  The data has to be laid
out in a local order physically, so the twin owns reordered copies of the mesh, the dofmaps, the constraint and the
Dirichlet markers (the same arrays the reference's ``create_*`` calls would hold after DOLFINx's own dof reordering;
numbering is not part of the reference's contract) and the existing kernels run on them unchanged.  The values reach the
caller's numbering inside the kernels: every write of a float64 kernel goes through ``mpcx_matrix_args_t::val_map`` (twin CSR
position -> caller CSR position) / ``mpcx_vector_args_t::row_map`` (round 5; ``MPCX_TWIN_HANDBACK=eager`` brings back round
4's passes -- ``mpcx_permute_values``, 8 + 8 + 4 bytes per entry, and a gather of the vector -- which block-scalar storage and
the other scalar types still use; ``lazy`` defers that pass until the values are read).

The reference's call sequence stays what it is (python/src/dolfinx_mpc/assemble_matrix.py:21-65,
assemble_vector.py:25-104): the twin is consulted inside ``assemble_matrix`` / ``assemble_vector`` / ``apply_lifting``.

Switch: ``MPCX_AUTO_REORDER`` = ``0`` off, ``1`` always (tests), unset: meshes of at least ``MPCX_AUTO_REORDER_MIN_CELLS``
(default 50 000) cells without tile hints on one process.

Memory: the twin holds a second copy of the mesh, of every dofmap / constraint / Function it has been shown, and -- per
matrix -- a second pattern plus 2 x 4 (8 beyond 2^32 entries) bytes per entry for the two permutation indices (config 2:
+ 3 GB per matrix; a second value array only with the eager / lazy hand-back).  The copies of Forms, Functions, Dirichlet conditions and constraints live exactly as long
as the caller's objects do (weak references: a time loop that rebuilds its forms every step does not accumulate twins),
a matrix's twin as long as the matrix; a (matrix, form) pair the twin could not represent is remembered and not retried."""

from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Optional

import numpy as np

from . import _device as D
from . import _native
from .fem import Constant, DirichletBC, Form, Function, FunctionSpace, Integral
from .mesh import Mesh, renumber


def wanted(mesh: Mesh) -> bool:
    mode = os.environ.get("MPCX_AUTO_REORDER", "")
    if mode == "0" or getattr(mesh, "_is_twin", False):
        return False
    if getattr(mesh, "partition", None) is not None and mesh.partition.get("world", 1) > 1:
        return False
    if mesh.node_tile_offsets is not None:
        return False
    if mode == "1":
        return True
    return mesh.num_cells >= int(os.environ.get("MPCX_AUTO_REORDER_MIN_CELLS", 50000))


def _morton_orders(mesh: Mesh):
    """(node_new_of_old, cell_old_of_new): nodes along a Z-order curve, cells by their lowest new node (the rule of
    ``mesh.reorder_spatial``); on the device when there is one (100 M cells: sorts of seconds on the host)."""
    x = mesh.geometry.x
    cells = mesh.geometry.dofmap
    try:
        import torch

        gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        gpu = False
    if not gpu:
        lo, hi = x.min(axis=0), x.max(axis=0)
        span = np.where(hi > lo, hi - lo, 1.0)
        q = np.minimum(((x - lo) / span * (1 << 21)).astype(np.int64), (1 << 21) - 1)
        code = np.zeros(x.shape[0], dtype=np.int64)
        for b in range(21):
            for d in range(3):
                code |= ((q[:, d] >> b) & 1) << (3 * b + d)
        order = np.argsort(code, kind="stable")
        perm = np.empty_like(order)
        perm[order] = np.arange(order.size)
        key = perm[cells.astype(np.int64)].min(axis=1)
        return perm, np.argsort(key, kind="stable")
    import torch

    dev = _native.require_gpu()
    xt = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    lo, hi = xt.min(dim=0).values, xt.max(dim=0).values
    span = torch.where(hi > lo, hi - lo, torch.ones_like(hi))
    q = torch.clamp(((xt - lo) / span * float(1 << 21)).to(torch.int64), max=(1 << 21) - 1)
    code = torch.zeros(x.shape[0], dtype=torch.int64, device=dev)
    for b in range(21):
        for d in range(3):
            code |= ((q[:, d] >> b) & 1) << (3 * b + d)
    order = torch.argsort(code, stable=True)
    perm = torch.empty_like(order)
    perm[order] = torch.arange(order.numel(), device=dev)
    ct = torch.from_numpy(np.ascontiguousarray(cells)).to(dev).long()
    key = perm[ct].min(dim=1).values
    cell_order = torch.argsort(key, stable=True)
    return perm.cpu().numpy(), cell_order.cpu().numpy()


def _gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:  # pragma: no cover
        return False


def _renumber(mesh: Mesh, perm: np.ndarray, cell_order: np.ndarray):
    """(the renumbered mesh, cell_new_of_old): ``mesh.renumber`` with the gathers on the device when there is one -- the
    host's fancy indexing of 4 x 10^8 node ids took 12 of the 23 s a shuffled 256^3 mesh spent in the library before its
    first assembly (tools/probes/twin_setup_profile.py)"""
    if not _gpu():
        inv = np.empty(cell_order.size, dtype=np.int64)
        inv[cell_order] = np.arange(cell_order.size)
        return renumber(mesh, perm, cell_order), inv
    import torch

    dev = _native.require_gpu()
    nn, nc, nv = mesh.num_nodes, mesh.num_cells, mesh.geometry.dofmap.shape[1]
    p = torch.from_numpy(np.ascontiguousarray(perm, dtype=np.int64)).to(dev)
    o = torch.from_numpy(np.ascontiguousarray(cell_order, dtype=np.int64)).to(dev)
    xt = torch.from_numpy(np.ascontiguousarray(mesh.geometry.x)).to(dev)
    ct = torch.from_numpy(np.ascontiguousarray(mesh.geometry.dofmap)).to(dev)
    x2, c2 = torch.empty_like(xt), torch.empty_like(ct)
    inv = torch.empty_like(o)
    # (the library's own gathers: torch's advanced indexing with 4 x 10^8 indices faulted on this stack)
    _native.check(_native.lib().mpcx_renumber_mesh(xt.data_ptr(), nn, ct.data_ptr(), nc, nv, p.data_ptr(), o.data_ptr(), x2.data_ptr(),
                                                   c2.data_ptr(), inv.data_ptr(), D.stream_ptr()), "mpcx_renumber_mesh")
    out = Mesh(x2.cpu().numpy(), c2.cpu().numpy(), mesh.cell_name)
    return out, inv.cpu().numpy()


class _WeakIdCache:
    """identity-keyed cache whose entries die with their key object (the twin used to keep every Form, Function,
    DirichletBC and constraint it had seen alive, with their device buffers and plans).  Values must not refer to the key."""

    def __init__(self):
        self._d = {}

    def get(self, obj):
        hit = self._d.get(id(obj))
        if hit is None:
            return None
        if hit[0]() is not obj:  # (the id was reused by a new object before the callback ran)
            self._d.pop(id(obj), None)
            return None
        return hit[1]

    def put(self, obj, value):
        key, d = id(obj), self._d

        def drop(_ref, key=key, d=d):
            d.pop(key, None)

        d[key] = (weakref.ref(obj, drop), value)
        return value

    def __len__(self):
        return len(self._d)


class Twin:
    """the reordered copy of one mesh and of everything built on it that an assembly call has been given"""

    def __init__(self, mesh: Mesh):
        self.mesh = mesh
        perm, cell_order = _morton_orders(mesh)
        self.node_new_of_old = perm
        self.cell_old_of_new = cell_order
        self.mesh2, self.cell_new_of_old = _renumber(mesh, perm, cell_order)
        self.mesh2._is_twin = True
        self.mesh2.node_tile_offsets = np.arange(0, self.mesh2.num_nodes, int(os.environ.get("MPCX_AUTO_REORDER_TILE", 512)),
                                                 dtype=np.int32)
        self.version = mesh.geometry.version
        self._spaces, self._functions, self._bcs, self._forms, self._mpcs = (_WeakIdCache() for _ in range(5))
        self._failed = _WeakIdCache()  # matrix -> ids of the forms whose twin pattern / plan could not be represented

    # -- geometry ------------------------------------------------------------------------------------------------
    def sync_geometry(self):
        """a moved mesh (mesh.geometry.x = ...) moves the twin"""
        if self.version != self.mesh.geometry.version:
            x2 = np.empty_like(self.mesh.geometry.x)
            x2[self.node_new_of_old] = self.mesh.geometry.x
            self.mesh2.geometry.x = x2
            self.version = self.mesh.geometry.version

    # -- spaces --------------------------------------------------------------------------------------------------
    def space(self, V: FunctionSpace):
        """(twin space, unrolled dof permutation new_of_old as numpy int64, the same on the device)"""
        hit = self._spaces.get(V)
        if hit is not None:
            return hit
        bs = V.dofmap.bs
        V2 = FunctionSpace(self.mesh2, ("Lagrange", V.degree), (bs,) if bs > 1 else None)
        # local dof order inside a cell is kept by the renumbering (local vertex order is), so the dofmaps of a cell and
        # of its twin cell list the same dofs position by position
        nblocks = V.num_dofs // bs
        d_pu = None
        if V.degree == 1 and not getattr(V, "general", False):
            new_of_old = np.asarray(self.node_new_of_old, dtype=np.int64)  # P1: the dofs are the nodes
            ok = V2.num_dofs == V.num_dofs
        elif _gpu():
            import torch

            dev = _native.require_gpu()
            t_new = torch.full((nblocks,), -1, dtype=torch.int64, device=dev)
            old_dm = torch.from_numpy(np.ascontiguousarray(V.dofmap.list)).to(dev)
            new_dm = torch.from_numpy(np.ascontiguousarray(V2.dofmap.list)).to(dev)
            inv = torch.from_numpy(np.ascontiguousarray(self.cell_new_of_old, dtype=np.int64)).to(dev)
            _native.check(_native.lib().mpcx_dof_permutation(old_dm.data_ptr(), new_dm.data_ptr(), inv.data_ptr(), old_dm.shape[0],
                                                             old_dm.shape[1], t_new.data_ptr(), D.stream_ptr()), "mpcx_dof_permutation")
            ok = V2.num_dofs == V.num_dofs and int(t_new.min().item()) >= 0
            new_of_old = t_new.cpu().numpy()
            del t_new, old_dm, new_dm, inv
        else:
            new_of_old = np.full(nblocks, -1, dtype=np.int64)
            new_of_old[V.dofmap.list.reshape(-1)] = V2.dofmap.list[self.cell_new_of_old].reshape(-1)
            ok = V2.num_dofs == V.num_dofs and not (new_of_old < 0).any()
        if not ok:
            raise _native.PlanNotRepresentable("automatic reordering: the space has dofs no cell refers to")
        pu = (new_of_old[:, None] * bs + np.arange(bs)[None, :]).reshape(-1) if bs > 1 else new_of_old
        if _gpu():
            import torch

            d_pu = torch.from_numpy(np.ascontiguousarray(pu)).to(_native.require_gpu())
        self._spaces.put(V, (V2, pu, d_pu))
        return V2, pu, d_pu

    # -- values --------------------------------------------------------------------------------------------------
    def function(self, f: Function) -> Function:
        V2, pu, _ = self.space(f.function_space)
        hit = self._functions.get(f)
        if hit is None:
            hit = self._functions.put(f, [Function(V2), None])
        cur = f.x._data
        if hit[1] is None or not D.same_values(cur, hit[1]):
            hit[1] = cur.copy()
            hit[0].x._data[pu] = cur
        return hit[0]

    def bc(self, bc: DirichletBC) -> DirichletBC:
        V2, pu, _ = self.space(bc.function_space)
        b2 = self._bcs.get(bc)
        if b2 is None:
            b2 = DirichletBC.__new__(DirichletBC)
            b2.function_space = V2
            b2._dofs = np.ascontiguousarray(pu[bc._dofs], dtype=np.int32)
            b2.value = bc.value
            self._bcs.put(bc, b2)
        v = bc.value
        if isinstance(v, Function):
            b2.value = self.function(v)
        elif isinstance(v, Constant):
            b2.value = v
        else:
            arr = np.asarray(v, dtype=np.float64)
            if arr.size > max(bc.function_space.dofmap.bs, 1) and arr.size == bc.function_space.num_dofs:
                out = np.empty_like(arr.reshape(-1))
                out[pu] = arr.reshape(-1)
                b2.value = out
            else:
                b2.value = v
        return b2

    def bcs(self, bcs):
        return [self.bc(b) for b in bcs]

    # -- forms ---------------------------------------------------------------------------------------------------
    def form(self, form: Form) -> Form:
        hit = self._forms.get(form)
        if hit is None:
            if any(getattr(integ.kernel, "ufcx_transforms", None) is not None for integ in form.integrals):
                # the cell permutation words depend on the GLOBAL vertex numbering, which the twin changes: such forms are
                # assembled in the caller's numbering (the callers fall back on PlanNotRepresentable)
                raise _native.PlanNotRepresentable("imported kernel with dof transformations: no locality twin")
            spaces2 = [self.space(V)[0] for V in form.function_spaces]
            integrals, sources = [], []
            for integ in form.integrals:
                ents = np.asarray(integ.entities)
                n = ents.shape[0]
                if integ.itype == "cell":
                    if n == self.mesh.num_cells and np.array_equal(ents, np.arange(n, dtype=ents.dtype)):
                        order, ents2 = self.cell_old_of_new, np.arange(n, dtype=ents.dtype)
                    else:  # a sum over a subset: any order; ascending twin cells keep the reads local
                        mapped = self.cell_new_of_old[ents.astype(np.int64)]
                        order = np.argsort(mapped, kind="stable")
                        ents2 = mapped[order].astype(ents.dtype)
                else:
                    mapped = self.cell_new_of_old[ents[:, 0].astype(np.int64)]
                    order = np.argsort(mapped, kind="stable")
                    ents2 = np.stack([mapped[order], ents[order, 1].astype(np.int64)], axis=1).astype(ents.dtype)
                integrals.append(Integral(integ.itype, np.ascontiguousarray(ents2), integ.kernel, None, integ.constant))
                sources.append(order)
            hit = self._forms.put(form, (Form(spaces2, integrals), sources, [None] * len(integrals)))
        form2, orders, packed = hit
        # coefficients are live: Functions through their twins, packed arrays re-ordered when their values changed
        for k, (integ, integ2) in enumerate(zip(form.integrals, form2.integrals)):
            c = integ.coefficient
            if c is None:
                integ2.coefficient = None
            elif isinstance(c, np.ndarray):
                if packed[k] is None or not D.same_values(c, packed[k][0]):
                    packed[k] = (c.copy(), np.ascontiguousarray(c[orders[k]]))
                integ2.coefficient = packed[k][1]
            elif isinstance(c, (list, tuple)):
                integ2.coefficient = [self.function(g) for g in c]
            else:
                integ2.coefficient = self.function(c)
            integ2.constant = integ.constant
        return form2

    # -- constraints ---------------------------------------------------------------------------------------------
    def mpc(self, mpc):
        from .multipointconstraint import MultiPointConstraint

        hit = self._mpcs.get(mpc)
        if hit is not None:
            return hit
        mpc._not_finalized()
        V2, pu, _ = self.space(mpc.function_space)
        slaves = np.asarray(mpc.slaves, dtype=np.int64)
        madj = mpc.masters
        moff = np.asarray(madj.offsets, dtype=np.int64)
        coeffs, _coff = mpc.coefficients()
        lo, hi = moff[slaves], moff[slaves + 1]
        cnt = hi - lo
        offsets = np.zeros(slaves.size + 1, dtype=np.int64)
        np.cumsum(cnt, out=offsets[1:])
        idx = np.repeat(lo - offsets[:-1], cnt) + np.arange(int(offsets[-1]))
        masters = np.asarray(madj.array, dtype=np.int64)[idx]
        m2 = MultiPointConstraint(V2)
        m2.add_constraint(V2, pu[slaves].astype(np.int32), pu[masters].astype(np.int64), np.asarray(coeffs)[idx],
                          np.zeros(masters.size, dtype=np.int32), offsets.astype(np.int32))
        m2.finalize()
        self._mpcs.put(mpc, m2)
        return m2

    # -- matrices ------------------------------------------------------------------------------------------------
    def matrix(self, A, form: Form, mpc0, mpc1):
        """(twin matrix, src, wide): the twin of A (pattern of the twin form / constraints) and, for every entry of A, the
        position of its value in the twin"""
        import torch

        from .assemble_matrix import create_matrix

        failed = self._failed.get(A)
        if failed is not None and id(form) in failed:
            raise _native.PlanNotRepresentable("automatic reordering: not representable for this matrix and form (remembered)")
        hit = getattr(A, "_twin", None)
        if hit is not None and hit[0] is self:
            return hit[1:]
        try:
            return self._build_matrix(A, form, mpc0, mpc1)
        except _native.PlanNotRepresentable:
            self.remember_failure(A, form)
            raise

    def remember_failure(self, A, form: Form):
        """the (matrix, form) pair is assembled in the caller's numbering from now on: the failed build is not repeated"""
        failed = self._failed.get(A)
        if failed is None:
            failed = self._failed.put(A, set())
        failed.add(id(form))

    def _build_matrix(self, A, form: Form, mpc0, mpc1):
        import torch

        from .assemble_matrix import create_matrix

        form2 = self.form(form)
        A2 = create_matrix(form2, self.mpc(mpc0), self.mpc(mpc1))
        if A2.nnz != A.nnz or A2.shape != A.shape:
            raise _native.PlanNotRepresentable("automatic reordering: the twin pattern differs from the caller's")
        L = _native.lib()
        V0, V1 = form.function_spaces
        dev = A.device

        def new_of_old(V):
            return torch.from_numpy(self.space(V)[1].astype(np.int32)).to(dev)

        n0 = new_of_old(V0)
        n1 = n0 if V1 is V0 else new_of_old(V1)
        wide = A.nnz >= 2 ** 32
        dest = torch.empty(max(A.nnz, 1), dtype=torch.int64 if wide else torch.int32, device=dev)
        bad = torch.zeros(1, dtype=torch.int32, device=dev)
        rc = L.mpcx_csr_permutation(A.shape[0], A.d_rowptr.data_ptr(), A.d_cols.data_ptr(), n0.data_ptr(), n1.data_ptr(),
                                    A2.d_rowptr.data_ptr(), A2.d_cols.data_ptr(), dest.data_ptr(), int(wide), bad.data_ptr(),
                                    D.stream_ptr())
        _native.check(rc, "mpcx_csr_permutation")
        if int(bad.item()):
            raise _native.PlanNotRepresentable("automatic reordering: an entry of the caller's pattern has no counterpart in the twin's")
        A._twin = (self, A2, dest, wide)
        return A2, dest, wide


def twin_of(mesh: Mesh) -> Optional[Twin]:
    """the mesh's twin (built on first use) when automatic reordering applies to it"""
    if not wanted(mesh):
        return None
    tw = getattr(mesh, "_twin", None)
    if tw is None:
        tw = mesh._twin = Twin(mesh)
    tw.sync_geometry()
    return tw


# ---------------------------------------------------------------------------------------------------------------
# the three entry points of the path, on the twin
# ---------------------------------------------------------------------------------------------------------------
def assemble_matrix(tw: Twin, form: Form, mpc0, mpc1, bcs, diagval, A, alg: int):
    """the twin's assembly and the hand-back of the values, both on A's library stream (one chain: nothing waits on the
    caller's stream, so the vector assembly of the same step runs beside it)"""
    import importlib

    import torch

    from .la import side_stream

    am = importlib.import_module(__package__ + ".assemble_matrix")
    A2, src, wide = tw.matrix(A, form, mpc0, mpc1)
    form2, m0, bcs2 = tw.form(form), tw.mpc(mpc0), tw.bcs(bcs)
    m1 = m0 if mpc1 is mpc0 else tw.mpc(mpc1)
    D.mesh_device(form2.mesh)
    mode = os.environ.get("MPCX_TWIN_HANDBACK", "fused")
    lazy = mode == "lazy"
    # "fused" (default): the twin's kernels write every value straight to its position in the caller's CSR
    # (mpcx_matrix_args_t::val_map = the inverse of ``src``): no second pass over the values, and the twin keeps no value
    # array of its own.  float64 only; a form that takes block-scalar storage keeps its compact values in the twin and is
    # handed back by the pass below.
    A2._write_through = None
    if mode == "fused" and A.dtype == torch.float64 and A._exchange is None:
        maps = getattr(A2, "_val_map", None)
        if maps is None:
            L = _native.lib()
            inv = torch.empty_like(src)
            _native.check(L.mpcx_invert_permutation(A.nnz, src.data_ptr(), int(wide), inv.data_ptr(), D.stream_ptr()),
                          "mpcx_invert_permutation")
            omap = odelta = None
            bad = torch.zeros(1, dtype=torch.int32, device=src.device)
            # MPCX_TWIN_WRITE_ORDER=1: consecutive lanes write consecutive addresses of a caller's row (out_map / out_delta).
            # Measured and not the default: config 2 shuffled 4.88 ms per step against 4.55 ms with the plain scatter -- the
            # cost is the random 216-byte row segments in HBM, not the order of the lanes inside them.
            if os.environ.get("MPCX_TWIN_WRITE_ORDER", "0") == "1":
                omap, odelta = torch.empty_like(src), torch.empty(max(A.nnz, 1), dtype=torch.int16, device=src.device)
                _native.check(L.mpcx_write_out_order(A2.shape[0], A2.d_rowptr.data_ptr(), inv.data_ptr(), int(wide), omap.data_ptr(),
                                                     odelta.data_ptr(), bad.data_ptr(), D.stream_ptr()), "mpcx_write_out_order")
                if int(bad.item()):
                    omap = odelta = None  # (the kernels scatter through ``inv``)
            maps = A2._val_map = (inv, omap, odelta)
        A2._write_through = (A, maps[0], wide, maps[1], maps[2])
    with side_stream("matrix", A):
        am._assemble_matrix_on_stream(form2, m0, m1, bcs2, diagval, A2, alg)
        A._compact_stale = False
        through, A2._write_through = A2._write_through is not None, None  # (no reference cycle A <-> A2 is left behind)
        if through and not A2._compact_stale:
            A._twin_stale = False  # (the values are in place)
        elif lazy:
            # the values stay in the twin's matrix until somebody reads ``A.vals`` (to_scipy, a solver, an exchange): like
            # block-scalar storage, the pass that writes them to the caller's CSR positions runs on demand, once per assembly
            # (PETSc keeps its own internal ordering behind MatSetValuesLocal as well)
            A._twin_stale = True
        else:
            A._twin_stale = False
            _native.check(_native.lib().mpcx_permute_values(A.nnz, src.data_ptr(), int(wide), A2.vals.data_ptr(), A.vals.data_ptr(),
                                                            D.stream_ptr()), "mpcx_permute_values")
    return A


def hand_back(A):
    """``A.vals`` of a matrix assembled with MPCX_TWIN_HANDBACK=lazy: the twin's values written to the caller's positions, on
    the current stream (which has already been ordered after the assembly by ``A._wait_ready()``)"""
    A._twin_stale = False
    tw, A2, src, wide = A._twin
    _ = A.vals  # (allocates on first use; the flag is already cleared)
    _native.check(_native.lib().mpcx_permute_values(A.nnz, src.data_ptr(), int(wide), A2.vals.data_ptr(), A._vals.data_ptr(),
                                                    D.stream_ptr()), "mpcx_permute_values")


def assemble_vector(tw: Twin, form: Form, mpc, b, alg: int):
    import importlib

    import torch

    from .la import Vector, side_stream

    av = importlib.import_module(__package__ + ".assemble_vector")
    b2 = getattr(b, "_twin", None)
    if b2 is None or b2[0] is not tw:
        b2 = b._twin = (tw, Vector(b.size))
    b2 = b2[1]
    form2, m2 = tw.form(form), tw.mpc(mpc)
    V2, _, d_pu = tw.space(mpc.function_space)
    D.mesh_device(form2.mesh)
    # "fused" (default): the twin's kernels add straight into the caller's vector (mpcx_vector_args_t::row_map = the inverse
    # of the dof permutation); otherwise the twin's vector is gathered into the caller's by a pass of its own
    b2._write_through = None
    if os.environ.get("MPCX_TWIN_HANDBACK", "fused") == "fused" and b.array.dtype == torch.float64:
        rm = getattr(V2, "_row_map", None)
        if rm is None:
            rm = torch.empty(d_pu.numel(), dtype=torch.int32, device=d_pu.device)
            rm[d_pu] = torch.arange(d_pu.numel(), dtype=torch.int32, device=d_pu.device)
            V2._row_map = rm
        b2._write_through = (b, rm)
    with side_stream("vector", b):
        av._assemble_vector_on_stream(form2, m2, b2, alg)
        through, b2._write_through = b2._write_through is not None, None
        if not through:
            torch.index_select(b2.array, 0, d_pu, out=b.array)
    return b


def apply_lifting(tw: Twin, b, forms, bcs, mpc, x0, scale):
    import torch

    from . import apply_lifting as al_mod
    from .la import Vector

    _, _, d_pu = tw.space(mpc.function_space)
    lift = Vector(b.size)
    forms2 = [None if f is None else tw.form(f) for f in forms]
    bcs2 = [tw.bcs(list(g)) for g in bcs]
    x02 = None
    if x0:
        x02 = []
        for f, v in zip(forms, x0):
            arr = v.array if hasattr(v, "array") else v
            if f is None:
                x02.append(v)
                continue
            _, _, dpu1 = tw.space(f.function_spaces[1])
            t = arr if isinstance(arr, torch.Tensor) else torch.as_tensor(np.asarray(arr), device=lift.device)
            out = torch.empty_like(t)
            out[dpu1] = t
            w = Vector(out.numel())
            w.array.copy_(out)
            x02.append(w)
    al_mod(lift, forms2, bcs2, tw.mpc(mpc), x0=x02, scale=scale)
    b.array.add_(torch.index_select(lift.array, 0, d_pu))
