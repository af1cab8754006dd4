"""``MultiPointConstraint`` with the reference's API surface
(python/src/dolfinx_mpc/multipointconstraint.py:87-631), backed by flat arrays.

``finalize()`` produces what cpp/MultiPointConstraint.h:36-126 produces
(``is_slave``, sorted ``slaves``, slave->masters/coeffs/owners adjacency over
all local dofs, ``cell_to_slaves``) through the native host routines
``mpcx_mpc_finalize`` / ``mpcx_cell_to_slaves``; device mirrors are created
lazily for the HIP kernels.
"""

from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, Optional, Sequence

import numpy as np

from .common import timed
from . import _native
from .fem import DirichletBC, FunctionSpace


class AdjacencyList:
    """Minimal ``dolfinx.graph.AdjacencyList``: ``array``, ``offsets``, ``links``."""

    def __init__(self, array: np.ndarray, offsets: np.ndarray):
        self.array = array
        self.offsets = offsets

    @property
    def num_nodes(self) -> int:
        return self.offsets.size - 1

    def links(self, i: int) -> np.ndarray:
        return self.array[self.offsets[i] : self.offsets[i + 1]]

    def num_links(self, i: int) -> int:
        return int(self.offsets[i + 1] - self.offsets[i])


class MPCData:
    """python/src/dolfinx_mpc/multipointconstraint.py:44-84"""

    def __init__(self, slaves, masters, coeffs, owners, offsets):
        self.slaves = np.asarray(slaves, dtype=np.int32)
        self.masters = np.asarray(masters, dtype=np.int64)
        self.coeffs = np.asarray(coeffs)
        self.owners = np.asarray(owners, dtype=np.int32)
        self.offsets = np.asarray(offsets, dtype=np.int32)


def _barycentric(mesh, cells, p):
    """barycentric coordinates of points p[i] in cells[i] (affine simplices) and the cells' vertex coordinates"""
    x = mesh.geometry.x
    tdim = mesh.tdim
    xv = x[mesh.geometry.dofmap[cells]]  # (n, tdim+1, 3)
    J = np.transpose(xv[:, 1:, :] - xv[:, :1, :], (0, 2, 1))[:, :, :tdim]  # (n, 3, tdim)
    rhs = p - xv[:, 0, :]
    if tdim == 2:
        J, rhs = J[:, :2, :], rhs[:, :2]
    mu = np.linalg.solve(J, rhs[:, :, None])[:, :, 0]
    lam = np.concatenate([1.0 - mu.sum(axis=1, keepdims=True), mu], axis=1)
    return lam, xv


def locate_points(V: FunctionSpace, pts: np.ndarray, cells: Optional[np.ndarray] = None, tol: float = 1e-10, k: int = 16):
    """For every point the index of a cell (of ``cells``, default all owned cells) that contains it, or -1, and the
    values of V's scalar basis functions of that cell at the point, (n, nd) in cell-dof order -- what the reference
    gets from its bounding-box-tree collision search + ``evaluate_basis_functions`` (cpp/ContactConstraint.h:466-475,
    cpp/PeriodicConstraint.h:139-150).  Candidates: the cells with the nearest centroids, then every cell for points
    the shortlist missed.  Affine simplices; Lagrange P1 / P2 in barycentric form."""
    from scipy.spatial import cKDTree

    mesh = V.mesh
    tdim = mesh.tdim
    pts = np.asarray(pts, dtype=np.float64).reshape(-1, 3)
    cand = np.arange(mesh.num_owned_cells, dtype=np.int64) if cells is None else np.unique(np.asarray(cells, dtype=np.int64))
    found = np.full(pts.shape[0], -1, dtype=np.int64)
    lam_found = np.zeros((pts.shape[0], tdim + 1))
    if cand.size and pts.shape[0]:
        cent = mesh.geometry.x[mesh.geometry.dofmap[cand]].mean(axis=1)
        kk = int(min(k, cand.size))
        _, near = cKDTree(cent).query(pts, k=kk)
        near = near.reshape(pts.shape[0], kk)

        def inside(lam, xv):
            h = np.linalg.norm(xv[:, 1, :] - xv[:, 0, :], axis=1)
            return lam.min(axis=1) >= -tol / np.maximum(h, 1e-300) - 1e-10

        for col in range(kk):
            todo = np.flatnonzero(found < 0)
            if todo.size == 0:
                break
            cc = cand[near[todo, col]]
            lam, xv = _barycentric(mesh, cc, pts[todo])
            ok = inside(lam, xv)
            found[todo[ok]] = cc[ok]
            lam_found[todo[ok]] = lam[ok]
        for i in np.flatnonzero(found < 0):  # points the nearest-centroid shortlist missed: try every candidate cell
            lam, xv = _barycentric(mesh, cand, np.repeat(pts[i][None, :], cand.size, axis=0))
            ok = np.flatnonzero(inside(lam, xv))
            if ok.size:
                found[i], lam_found[i] = cand[ok[0]], lam[ok[0]]
    if V.degree == 1:
        basis = lam_found
    else:
        from .mesh import TET_EDGES, TRI_EDGES

        le = TET_EDGES if tdim == 3 else TRI_EDGES
        basis = np.concatenate([lam_found * (2.0 * lam_found - 1.0), 4.0 * lam_found[:, le[:, 0]] * lam_found[:, le[:, 1]]], axis=1)
    basis[found < 0] = 0.0
    return found, basis


@timed("~MPC: Facet normal projection")
def create_normal_approximation(V: FunctionSpace, meshtags, value: int):
    """python/src/dolfinx_mpc/utils/mpc_utils.py:422-438 / cpp/utils.h:201-267: a Function in the (vector) space V
    whose blocks in the closure of the facets tagged ``value`` hold the sum of the adjacent tagged facets' unit normals
    (each aligned with the first one), divided by its squared length -- the reference divides by ``abs(acc)`` with
    ``acc`` the sum of squares (cpp/utils.h:254-261); only the direction matters to the slip constraints."""
    from .fem import Function
    from .mesh import TET_EDGES, TET_FACETS, TRI_EDGES, TRI_FACETS

    mesh = V.mesh
    tdim = mesh.tdim
    bs = V.dofmap.bs
    nh = Function(V)
    facets = meshtags.find(value)
    if facets.shape[0] == 0:
        return nh
    lf = TET_FACETS if tdim == 3 else TRI_FACETS
    le = TET_EDGES if tdim == 3 else TRI_EDGES
    x = mesh.geometry.x
    cells, loc = facets[:, 0].astype(np.int64), facets[:, 1].astype(np.int64)
    fv = mesh.geometry.dofmap[cells][np.arange(cells.size)[:, None], lf[loc]]  # (nf, tdim) vertices of the facets
    if tdim == 3:
        nrm = np.cross(x[fv[:, 1]] - x[fv[:, 0]], x[fv[:, 2]] - x[fv[:, 0]])
    else:
        t = x[fv[:, 1]] - x[fv[:, 0]]
        nrm = np.stack([t[:, 1], -t[:, 0], np.zeros(t.shape[0])], axis=1)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    acc = {}
    nv = tdim + 1
    for f in range(facets.shape[0]):
        ldofs = list(lf[loc[f]])
        if V.degree == 2:
            ldofs += [nv + e for e in range(le.shape[0]) if le[e][0] in lf[loc[f]] and le[e][1] in lf[loc[f]]]
        for b in V.dofmap.list[cells[f]][ldofs]:
            b = int(b)
            if b not in acc:
                acc[b] = (nrm[f].copy(), nrm[f].copy())  # (first normal, running sum)
            else:
                n0, tot = acc[b]
                d = float(n0 @ nrm[f])
                tot += (d / abs(d) if d != 0.0 else 1.0) * nrm[f]
    arr = nh.x.array
    for b, (_, tot) in acc.items():
        a2 = float(tot[:bs] @ tot[:bs])
        arr[b * bs:(b + 1) * bs] = tot[:bs] / a2 if a2 > 1e-10 else tot[:bs]
    return nh


class MultiPointConstraint:
    """Hold data for multi point constraint relationships.

    Args:
        V: The function space
        dtype: scalar type of the coefficients: float64 (the tuned kernels), float32, complex64, complex128 (the general
            per-entity kernels, csrc/mpcx_scalar.hip) -- the reference's four instantiations,
            python/src/dolfinx_mpc/multipointconstraint.py:55-64
    """

    def __init__(self, V: FunctionSpace, dtype=np.float64):
        _native.scalar_id(dtype)  # (raises NotImplementedError for anything else)
        dtype = np.dtype(dtype)
        self._slaves = np.array([], dtype=np.int32)
        self._masters = np.array([], dtype=np.int64)
        self._coeffs = np.array([], dtype=dtype)
        self._owners = np.array([], dtype=np.int32)
        self._offsets = np.array([0], dtype=np.int32)
        self.V = V
        self.finalized = False
        self._dtype = dtype
        self._dev = None
        self._cache = {}

    # -- building -------------------------------------------------------------
    def add_constraint(self, V: FunctionSpace, slaves, masters, coeffs, owners, offsets):
        """Add constraint given by numpy arrays
        (python/src/dolfinx_mpc/multipointconstraint.py:118-153): local slave
        dofs, global master dofs, coefficients, owners, offsets."""
        assert V is self.V
        self._already_finalized()
        slaves = np.asarray(slaves, dtype=np.int32)
        if len(slaves) > 0:
            offsets = np.asarray(offsets, dtype=np.int32)
            self._offsets = np.append(self._offsets, offsets[1:] + len(self._masters)).astype(np.int32)
            self._slaves = np.append(self._slaves, slaves).astype(np.int32)
            self._masters = np.append(self._masters, np.asarray(masters, dtype=np.int64)).astype(np.int64)
            self._coeffs = np.array(np.append(self._coeffs, coeffs), dtype=self._dtype)
            self._owners = np.append(self._owners, np.asarray(owners, dtype=np.int32)).astype(np.int32)

    def add_constraint_from_mpc_data(self, V: FunctionSpace, mpc_data: MPCData):
        self._already_finalized()
        self.add_constraint(V, mpc_data.slaves, mpc_data.masters, mpc_data.coeffs, mpc_data.owners, mpc_data.offsets)

    def finalize(self, where: Optional[str] = None) -> None:
        """Finalize the constraint (python/src/dolfinx_mpc/multipointconstraint.py:169-223): is_slave, the sorted
        slave list, the slave -> masters / coefficients / owners adjacency over all local dofs
        (cpp/MultiPointConstraint.h:36-126) and cell -> slaves (cpp/mpc_helpers.h:19-94).

        ``where``: "device" (HIP kernels: mark -> scan -> fill; the arrays stay in HBM where the assembly kernels
        read them and are downloaded only when a host accessor is used), "host" (the C++ routines) or None = env
        MPCX_FINALIZE, default: device when a GPU is present.  Both give the same arrays."""
        import os

        self._already_finalized()
        V = self.V
        nd = V.num_dofs
        imap = V.dofmap.index_map
        nowned = imap.size_local * V.dofmap.index_map_bs
        ns = self._slaves.size
        nm = self._masters.size
        # the finalize routines move fp64 coefficients; any other scalar type rides along as its POSITION in the input
        # (exactly representable) and is put in place afterwards
        true_coeffs = None
        if self._dtype != np.float64:
            true_coeffs = np.ascontiguousarray(self._coeffs, dtype=self._dtype)
        raw = dict(slaves=np.ascontiguousarray(self._slaves, dtype=np.int32),
                   masters=np.ascontiguousarray(self._masters, dtype=np.int64),
                   coeffs=(np.ascontiguousarray(self._coeffs, dtype=np.float64) if true_coeffs is None
                           else np.arange(nm, dtype=np.float64)),
                   owners=np.ascontiguousarray(self._owners, dtype=np.int32),
                   offsets=np.ascontiguousarray(self._offsets, dtype=np.int32))
        assert raw["offsets"].size == ns + 1 and raw["offsets"][-1] == nm and raw["coeffs"].size == nm and raw["owners"].size == nm
        if where is None:
            where = os.environ.get("MPCX_FINALIZE")
        if where is None:
            import torch

            where = "device" if torch.cuda.is_available() else "host"
        self._host = {}  # host copies of the finalized arrays (filled by the host routine, or lazily from the device)
        self._devt = None  # device tensors (filled by the device routine, or lazily from the host)
        if not (where.lower() == "device" and self._finalize_device(nd, nowned, ns, nm, raw)):
            self._finalize_host(nd, nowned, ns, nm, raw)
        if true_coeffs is not None:
            if self._devt is not None:
                import torch

                order = self._devt["coeffs"].to(torch.int64)
                self._devt["coeffs"] = torch.from_numpy(true_coeffs).to(order.device)[order].contiguous()
            else:
                self._host["coeffs"] = np.ascontiguousarray(true_coeffs[self._host["coeffs"].astype(np.int64)])
        # single process: the extended function space is V itself
        # (cpp/mpc_helpers.h:165-168)
        self.finalized = True
        del (self._slaves, self._masters, self._coeffs, self._owners, self._offsets)

    def _finalize_host(self, nd, nowned, ns, nm, raw):
        L = _native.lib()
        V = self.V
        is_slave = np.zeros(nd, dtype=np.int8)
        sorted_slaves = np.zeros(ns, dtype=np.int32)
        nloc = C.c_int32(0)
        moff = np.zeros(nd + 1, dtype=np.int32)
        mloc = np.zeros(nm, dtype=np.int32)
        cout = np.zeros(nm, dtype=np.float64)
        oout = np.zeros(nm, dtype=np.int32)
        p = _native._ptr
        rc = L.mpcx_mpc_finalize(nd, nowned, ns, p(raw["slaves"]), p(raw["masters"]), p(raw["coeffs"]), p(raw["owners"]),
                                 p(raw["offsets"]), p(is_slave), p(sorted_slaves), C.cast(C.byref(nloc), C.c_void_p), p(moff),
                                 p(mloc), p(cout), p(oout))
        _native.check(rc, "mpcx_mpc_finalize")
        # duplicates in the user's slave list collapse in the marker
        nuniq = int(is_slave.sum())
        self._num_local_slaves = int(nloc.value)
        self._num_slaves = nuniq
        # cell -> slaves (owned cells)
        dm = V.dofmap.list
        nc = dm.shape[0]
        c2s_off = np.zeros(nc + 1, dtype=np.int32)
        total = L.mpcx_cell_to_slaves(nc, dm.shape[1], V.dofmap.bs, p(dm), p(is_slave), p(c2s_off), None)
        if total < 0:
            _native.check(int(total), "mpcx_cell_to_slaves")
        c2s = np.zeros(int(total), dtype=np.int32)
        total = L.mpcx_cell_to_slaves(nc, dm.shape[1], V.dofmap.bs, p(dm), p(is_slave), p(c2s_off), p(c2s))
        self._host = dict(is_slave=is_slave, slaves=sorted_slaves[:nuniq].copy(), moff=moff, masters=mloc, coeffs=cout,
                          owners=oout, c2s_off=c2s_off, c2s=c2s)

    def _finalize_device(self, nd, nowned, ns, nm, raw) -> bool:
        """the same on the device (include/mpcx.h mpcx_mpc_finalize_device / mpcx_cell_to_slaves_device); False if
        the slave list holds a dof twice (the host routine's sequential semantics apply then)"""
        import torch

        from . import _device as D
        from . import _prims

        L = _native.lib()
        V = self.V
        dev = _native.require_gpu()
        st = D.stream_ptr()
        d = {k: D._to_dev(v, dev) for k, v in raw.items()}
        t = dict(is_slave=torch.empty(nd, dtype=torch.int8, device=dev),
                 slaves=torch.empty(max(ns, 1), dtype=torch.int32, device=dev),
                 moff=torch.empty(nd + 1, dtype=torch.int32, device=dev),
                 masters=torch.empty(max(nm, 1), dtype=torch.int32, device=dev),
                 coeffs=torch.empty(max(nm, 1), dtype=torch.float64, device=dev),
                 owners=torch.empty(max(nm, 1), dtype=torch.int32, device=dev))
        nloc = torch.zeros(1, dtype=torch.int32, device=dev)
        work = torch.empty(2 * nd + 2, dtype=torch.int32, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        args = (nd, nowned, ns, d["slaves"].data_ptr(), d["masters"].data_ptr(), d["coeffs"].data_ptr(), d["owners"].data_ptr(),
                d["offsets"].data_ptr(), t["is_slave"].data_ptr(), t["slaves"].data_ptr(), nloc.data_ptr(), t["moff"].data_ptr(),
                t["masters"].data_ptr(), t["coeffs"].data_ptr(), t["owners"].data_ptr(), work.data_ptr(), flag.data_ptr())
        temp, nb = _prims._workspace(lambda tp, n_: L.mpcx_mpc_finalize_device(*args, tp, n_, st), dev)
        _native.check(L.mpcx_mpc_finalize_device(*args, temp.data_ptr(), C.byref(nb), st), "mpcx_mpc_finalize_device")
        f = int(flag.item())
        if f & 1:
            raise RuntimeError("mpcx_mpc_finalize failed (-1): mpcx_mpc_finalize: slave index out of range")
        if f & 2:
            raise RuntimeError("mpcx_mpc_finalize failed (-2): mpcx_mpc_finalize: master index out of range (single-process "
                               "backend: global master index must equal a local dof)")
        if f & 4:
            return False
        nuniq = int(work[2 * nd].item())  # total of the slave-marker scan
        self._num_slaves = nuniq
        self._num_local_slaves = int(nloc.item())
        t["slaves"] = t["slaves"][:nuniq].contiguous() if nuniq else torch.zeros(0, dtype=torch.int32, device=dev)
        del work, temp
        # cell -> slaves over the device-resident dofmap (the one the assembly kernels read)
        dm = D.space_device(V)["dofmap"]
        nc, ndc = V.dofmap.list.shape
        if ndc * V.dofmap.bs > 256:
            raise RuntimeError("mpcx_cell_to_slaves: element too large")
        counts = torch.empty(max(nc, 1), dtype=torch.int32, device=dev)
        _native.check(L.mpcx_cell_to_slaves_device(nc, ndc, V.dofmap.bs, dm.data_ptr(), t["is_slave"].data_ptr(),
                                                   counts.data_ptr(), None, None, st), "mpcx_cell_to_slaves_device")
        c2s_off = _prims.scan_i32(counts[:nc])
        total = int(c2s_off[-1].item())
        c2s = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        if total:
            _native.check(L.mpcx_cell_to_slaves_device(nc, ndc, V.dofmap.bs, dm.data_ptr(), t["is_slave"].data_ptr(),
                                                       counts.data_ptr(), c2s_off.data_ptr(), c2s.data_ptr(), st),
                          "mpcx_cell_to_slaves_device")
        t["c2s_off"], t["c2s"] = c2s_off, c2s[:total]
        t["masters"], t["coeffs"], t["owners"] = t["masters"][:nm], t["coeffs"][:nm], t["owners"][:nm]
        self._devt = t
        return True

    def _h(self, name: str) -> np.ndarray:
        """host copy of a finalized array (downloaded from the device on first use)"""
        self._not_finalized()
        if name not in self._host:
            self._host[name] = self._devt[name].cpu().numpy()
        return self._host[name]

    def device_tensors(self) -> dict:
        """the finalized arrays as device tensors: is_slave int8 [ndofs], slaves int32 (sorted), moff int32 [ndofs + 1],
        masters int32, coeffs float64, owners int32, c2s_off int32 [ncells + 1], c2s int32 (uploaded on first use when
        the host routine finalized)"""
        self._not_finalized()
        if self._devt is None:
            from . import _device as D

            dev = _native.require_gpu()
            self._devt = {k: D._to_dev(v, dev) for k, v in self._host.items()}
            D._settle()
        return self._devt

    # -- convenience builders (structured / matching meshes only) --------------
    def _periodic(self, V: FunctionSpace, blocks: np.ndarray, relation, bcs, scale: float, tol: float):
        """u(x_i) = scale * u(relation(x_i)) for the dof blocks ``blocks`` (cpp/PeriodicConstraint.h:30-222, serial
        branch): a slave block with any component under a Dirichlet condition is dropped as a whole
        (cpp/utils.h:1459-1496 marks blocks, cpp/PeriodicConstraint.h:563-567 filters with it); the mapped point is
        located in a cell and the slave is tied to that cell's dofs with the values of their basis functions there,
        |scale * phi_j| > tol kept (:170-196).  Mapped points that coincide with a dof (matching meshes: one
        basis function is 1, the others vanish) get that dof with coefficient ``scale`` exactly; points in no cell
        are skipped (in serial the reference finds no collision for them either)."""
        from scipy.spatial import cKDTree

        if V is not self.V:
            raise RuntimeError("The input space has to be a sub space (or the full space) of the MPC")
        blocks = np.unique(np.asarray(blocks, dtype=np.int64))
        if blocks.size == 0:
            return
        x = V.tabulate_dof_coordinates()
        bs = V.dofmap.bs
        is_bc = np.zeros(V.num_dofs, dtype=np.int8)
        for bc in bcs or []:
            if V.contains(bc.function_space):  # cpp/utils.h:1470-1476: only conditions living in V
                bc.mark_dofs(is_bc)
        blocks = blocks[~is_bc.reshape(-1, bs)[blocks].any(axis=1)]
        if blocks.size == 0:
            return
        xm = np.ascontiguousarray(np.asarray(relation(x[blocks].T)).T, dtype=np.float64)
        # dofs AT the mapped points (matching meshes): exact coefficient, no basis evaluation
        lo, hi = xm.min(axis=0) - 1e-8, xm.max(axis=0) + 1e-8
        cand = np.flatnonzero(np.all((x >= lo) & (x <= hi), axis=1))
        mblk = np.full(blocks.size, -1, dtype=np.int64)
        if cand.size:
            dist, loc = cKDTree(x[cand]).query(xm)
            hit = dist <= 1e-8
            mblk[hit] = cand[loc[hit]]
        # (a) matching dofs, vectorised: one master per slave component
        hb, hm = blocks[mblk >= 0], mblk[mblk >= 0]
        slaves = [(hb[:, None] * bs + np.arange(bs)[None, :]).reshape(-1)]
        masters = [(hm[:, None] * bs + np.arange(bs)[None, :]).reshape(-1)]
        coeffs = [np.full(hb.size * bs, scale, dtype=np.float64)]
        counts = [np.ones(hb.size * bs, dtype=np.int64)]
        # (b) the others: cell of the mapped point, basis values there
        rest = np.flatnonzero(mblk < 0)
        if rest.size:
            cells, basis = locate_points(V, xm[rest], tol=max(float(tol), 1e-12))
            for r, c, phi in zip(rest, cells, basis):
                if c < 0:
                    continue  # no collision: skipped
                val = scale * phi
                keep = np.abs(val) > tol
                mb = V.dofmap.list[c][keep].astype(np.int64)
                for k in range(bs):
                    slaves.append(np.array([blocks[r] * bs + k]))
                    masters.append(mb * bs + k)
                    coeffs.append(val[keep])
                    counts.append(np.array([mb.size]))
        slaves, masters, coeffs = np.concatenate(slaves), np.concatenate(masters), np.concatenate(coeffs)
        if slaves.size == 0:
            return
        offsets = np.concatenate([[0], np.cumsum(np.concatenate(counts))]).astype(np.int32)
        self.add_constraint(V, slaves.astype(np.int32), masters.astype(np.int64), coeffs.astype(np.float64),
                            np.zeros(masters.size, dtype=np.int32), offsets)

    def create_periodic_constraint_geometrical(
        self,
        V: FunctionSpace,
        indicator: Callable[[np.ndarray], np.ndarray],
        relation: Callable[[np.ndarray], np.ndarray],
        bcs: Sequence[DirichletBC],
        scale: float = 1.0,
        tol: float = 500 * np.finfo(np.float64).eps,
        num_threads: Optional[int] = 1,
    ):
        """u(x_i) = scale * u(relation(x_i)) for all dofs with indicator(x_i)
        (python/src/dolfinx_mpc/multipointconstraint.py:282-323, cpp/PeriodicConstraint.h:30-222 + :560-600)."""
        if isinstance(scale, np.generic):
            scale = scale.item()
        x = V.tabulate_dof_coordinates()
        blocks = np.flatnonzero(np.asarray(indicator(x.T), dtype=bool))
        self._periodic(V, blocks, relation, bcs, float(scale), float(tol))

    def create_periodic_constraint_topological(
        self,
        V: FunctionSpace,
        meshtag,
        tag: int,
        relation: Callable[[np.ndarray], np.ndarray],
        bcs: Sequence[DirichletBC],
        scale: float = 1.0,
        tol: float = 500 * np.finfo(np.float64).eps,
        num_threads: Optional[int] = 1,
    ):
        """periodic condition for all closure dofs of the entities of ``meshtag`` with value ``tag``
        (python/src/dolfinx_mpc/multipointconstraint.py:225-281, cpp/PeriodicConstraint.h:480-530:
        locate_dofs_topological, then the same construction as the geometrical variant)."""
        from .fem import locate_dofs_topological

        if isinstance(scale, np.generic):
            scale = scale.item()
        blocks = locate_dofs_topological(V, meshtag.dim, meshtag.find(tag))
        self._periodic(V, blocks, relation, bcs, float(scale), float(tol))

    @timed("~MPC: Create slip constraint")
    def create_slip_constraint(self, space: FunctionSpace, facet_marker, v, bcs: Sequence[DirichletBC] = ()):
        """u . v = 0 on the dofs in the closure of the facets ``facet_marker = (meshtags, marker)``, ``v`` a Function
        in ``space`` holding the direction, usually the facet normal (python/src/dolfinx_mpc/multipointconstraint.py:
        325-399, cpp/SlipConstraint.h:16-175): dof blocks touched by a Dirichlet condition are left out as a whole
        (:95-101), slave = the component with the largest |v_i| of the block, masters = the other components of the
        same block, c_i = -v_i / v_s (every one of them, no tolerance filter).

        Older call shape kept: ``create_slip_constraint(V, blocks, normals, bcs)`` with explicit dof blocks and an
        (n, bs) array of directions."""
        from .fem import Function, locate_dofs_topological

        if space is not self.V:
            raise ValueError("Input space has to be a sub space of the MPC space")
        V = self.V
        bs = V.dofmap.bs
        is_bc = np.zeros(V.num_dofs, dtype=np.int8)
        for bc in bcs or []:
            if V.contains(bc.function_space):
                bc.mark_dofs(is_bc)
        if isinstance(facet_marker, tuple) and len(facet_marker) == 2 and hasattr(facet_marker[0], "find"):
            if not isinstance(v, Function) or v.function_space.dofmap.bs != bs:
                raise ValueError("v has to be a Function in the (blocked) space of the constraint")
            meshtags, marker = facet_marker
            blocks = np.unique(locate_dofs_topological(V, meshtags.dim, meshtags.find(marker))).astype(np.int64)
            blocks = blocks[~is_bc.reshape(-1, bs)[blocks].any(axis=1)]  # cpp/SlipConstraint.h:95-101 (blocks, cpp/utils.h:1459-1496)
            normals = v.x._data.reshape(-1, bs)[blocks]
        else:
            blocks = np.asarray(facet_marker, dtype=np.int64)
            normals = np.asarray(v, dtype=np.float64).reshape(blocks.size, -1)[:, :bs]
            keep = ~is_bc.reshape(-1, bs)[blocks].any(axis=1)
            blocks, normals = blocks[keep], normals[keep]
        n = blocks.size
        if n == 0:
            return
        sidx = np.argmax(np.abs(normals), axis=1)  # first maximum, like std::ranges::max_element
        comp = np.arange(bs)[None, :].repeat(n, 0)
        others = comp[comp != sidx[:, None]].reshape(n, bs - 1)
        ns = normals[np.arange(n), sidx]
        slaves = blocks * bs + sidx
        masters = (blocks[:, None] * bs + others).reshape(-1)
        coeffs = (-normals[np.arange(n)[:, None], others] / ns[:, None]).reshape(-1)
        offsets = np.arange(n + 1, dtype=np.int32) * (bs - 1)
        self.add_constraint(V, slaves.astype(np.int32), masters.astype(np.int64), coeffs.astype(np.float64),
                            np.zeros(masters.size, dtype=np.int32), offsets)

    def create_general_constraint(self, slave_master_dict: Dict[bytes, Dict[bytes, float]],
                                  subspace_slave: Optional[int] = None, subspace_master: Optional[int] = None):
        """python/src/dolfinx_mpc/multipointconstraint.py:342-398 /
        dictcondition.py: {slave point bytes: {master point bytes: coeff}}."""
        from scipy.spatial import cKDTree

        V = self.V
        x = V.tabulate_dof_coordinates()
        tree = cKDTree(x)
        bs = V.dofmap.bs
        dim = len(np.frombuffer(next(iter(slave_master_dict)), dtype=np.float64))

        def find(pt_bytes):
            pt = np.zeros(3)
            pt[:dim] = np.frombuffer(pt_bytes, dtype=np.float64)
            d, i = tree.query(pt)
            if d > 1e-8:
                raise ValueError(f"no dof at point {pt}")
            return int(i)

        slaves, masters, coeffs, offsets = [], [], [], [0]
        for sp, md in slave_master_dict.items():
            sblk = find(sp)
            scomps = range(bs) if subspace_slave is None else [subspace_slave]
            for k in scomps:
                slaves.append(sblk * bs + k)
                for mp, c in md.items():
                    mcomp = k if subspace_master is None else subspace_master
                    masters.append(find(mp) * bs + mcomp)
                    coeffs.append(c)
                offsets.append(len(masters))
        self.add_constraint(V, np.array(slaves, dtype=np.int32), np.array(masters, dtype=np.int64),
                            np.array(coeffs, dtype=np.float64), np.zeros(len(masters), dtype=np.int32),
                            np.array(offsets, dtype=np.int32))

    def _contact_collisions(self, meshtags, slave_marker: int, master_marker: int, eps2: float, allow_missing_masters: bool):
        """collision search shared by the contact builders (cpp/ContactConstraint.h:454-475 / :1010-1030, serial
        branch): the dof blocks in the closure of the slave facets, for each the master-side cell that contains its
        point, and the values of that cell's basis functions there.  Returns (slave_blocks, cells, basis (n, nd)) of
        the blocks that collided, or None when there is nothing to tie."""
        from scipy.spatial import cKDTree

        from .fem import locate_dofs_topological
        from .mesh import facet_vertices

        self._already_finalized()
        V = self.V
        mesh = V.mesh
        bs = V.dofmap.bs
        tdim = mesh.tdim
        fdim = tdim - 1
        x = mesh.geometry.x
        slave_blocks = locate_dofs_topological(V, fdim, meshtags.find(slave_marker))
        if slave_blocks.size == 0:
            return None
        mfac = meshtags.find(master_marker)
        if mfac.shape[0] == 0:
            if allow_missing_masters:
                return None
            raise RuntimeError("No masters found on contact surface (when executed in serial). Please make sure "
                               "that the surfaces are in contact, or increase the tolerance eps2.")
        pts = V.tabulate_dof_coordinates()[slave_blocks]
        # candidate master cells per slave point: the cells of the nearest master facets (by centroid)
        mcells = mfac[:, 0].astype(np.int64)
        cent = x[facet_vertices(mesh, mfac)].mean(axis=1)
        k = int(min(16, mfac.shape[0]))
        _, near = cKDTree(cent).query(pts, k=k)
        near = near.reshape(pts.shape[0], k)
        tol = max(np.sqrt(eps2), 1e-12)
        found = np.full(pts.shape[0], -1, dtype=np.int64)
        lam_found = np.zeros((pts.shape[0], tdim + 1))

        def barycentric(cells, p):
            """barycentric coordinates of points p[i] in cells[i] (affine simplices)"""
            xv = x[mesh.geometry.dofmap[cells]]  # (n, tdim+1, 3)
            J = np.transpose(xv[:, 1:, :] - xv[:, :1, :], (0, 2, 1))[:, :, :tdim]  # (n, 3, tdim)
            rhs = (p - xv[:, 0, :])
            if tdim == 2:
                J, rhs = J[:, :2, :], rhs[:, :2]
            mu = np.linalg.solve(J, rhs[:, :, None])[:, :, 0]
            lam = np.concatenate([1.0 - mu.sum(axis=1, keepdims=True), mu], axis=1)
            return lam, xv

        for col in range(k):
            todo = np.flatnonzero(found < 0)
            if todo.size == 0:
                break
            cells = mcells[near[todo, col]]
            lam, xv = barycentric(cells, pts[todo])
            h = np.linalg.norm(xv[:, 1, :] - xv[:, 0, :], axis=1)
            inside = lam.min(axis=1) >= -tol / np.maximum(h, 1e-300) - 1e-10
            hit = todo[inside]
            found[hit] = cells[inside]
            lam_found[hit] = lam[inside]
        missing = np.flatnonzero(found < 0)
        if missing.size:  # points the nearest-centroid shortlist missed: try every master cell
            ucells = np.unique(mcells)
            for i in missing:
                lam, xv = barycentric(ucells, np.repeat(pts[i][None, :], ucells.size, axis=0))
                h = np.linalg.norm(xv[:, 1, :] - xv[:, 0, :], axis=1)
                ok = np.flatnonzero(lam.min(axis=1) >= -tol / np.maximum(h, 1e-300) - 1e-10)
                if ok.size:
                    found[i], lam_found[i] = ucells[ok[0]], lam[ok[0]]
        missing = np.flatnonzero(found < 0)
        if missing.size and not allow_missing_masters:
            raise RuntimeError("No masters found on contact surface (when executed in serial). Please make sure "
                               "that the surfaces are in contact, or increase the tolerance eps2.")
        keep = found >= 0
        slave_blocks, found, lam_found = slave_blocks[keep], found[keep], lam_found[keep]
        # basis functions of the master cell at the slave point (Lagrange P1 / P2 in barycentric form)
        if V.degree == 1:
            basis = lam_found
        else:
            from .mesh import TET_EDGES, TRI_EDGES

            le = TET_EDGES if tdim == 3 else TRI_EDGES
            basis = np.concatenate([lam_found * (2.0 * lam_found - 1.0),
                                    4.0 * lam_found[:, le[:, 0]] * lam_found[:, le[:, 1]]], axis=1)
        return slave_blocks, found, basis

    @timed("~MPC: Inelastic condition")
    def create_contact_inelastic_condition(self, meshtags, slave_marker: int, master_marker: int,
                                           eps2: float = 1e-20, allow_missing_masters: bool = False,
                                           num_threads: Optional[int] = 1):
        """u_s = u_m between two sets of tagged facets whose surfaces coincide; the vertices need not
        align (python/src/dolfinx_mpc/multipointconstraint.py:465-501, cpp/ContactConstraint.h:908-1174,
        serial branch).  Every dof block in the closure of the slave facets is tied, per component, to
        the dof blocks of the master-side cell it collides with, weighted by that cell's basis functions
        at the slave point; weights with |c| <= 1e-6 are dropped (cpp/ContactConstraint.h:1033).

        Output shape: slaves = block * bs + j for j < bs, masters = master block * bs + j, the same
        weights for every component.  A slave point that touches no master cell raises RuntimeError
        unless ``allow_missing_masters`` (then the block is skipped), as the reference does in serial
        (cpp/ContactConstraint.h:1086-1094)."""
        hit = self._contact_collisions(meshtags, slave_marker, master_marker, eps2, allow_missing_masters)
        if hit is None:
            return
        slave_blocks, found, basis = hit
        V = self.V
        bs = V.dofmap.bs
        cell_blocks = V.dofmap.list[found]  # (n, nd)
        nz = np.abs(basis) > 1e-6  # cpp/ContactConstraint.h:1033
        cnt = nz.sum(axis=1)
        mblk = cell_blocks[nz]  # row-major: per slave block, masters in cell-dof order
        coef = basis[nz]
        n = slave_blocks.size
        # per component j: slave = block*bs + j, masters = master block*bs + j (cpp/ContactConstraint.h:1054-1066)
        blk_off = np.concatenate([[0], np.cumsum(cnt)])
        slaves = (slave_blocks[:, None] * bs + np.arange(bs)[None, :]).reshape(-1)
        rep = np.repeat(cnt, bs)
        offsets = np.concatenate([[0], np.cumsum(rep)]).astype(np.int32)
        # output entry -> (slave block, component, position inside the block's master list)
        seg = np.repeat(np.arange(n * bs), rep)
        within = np.arange(int(rep.sum())) - np.repeat(offsets[:-1].astype(np.int64), rep)
        src = blk_off[seg // bs] + within
        comp = seg % bs
        masters = mblk[src].astype(np.int64) * bs + comp
        coeffs = coef[src]
        self.add_constraint(V, slaves.astype(np.int32), masters.astype(np.int64), coeffs.astype(np.float64),
                            np.zeros(masters.size, dtype=np.int32), offsets)

    def create_contact_slip_condition(self, meshtags, slave_marker: int, master_marker: int, normal, eps2: float = 1e-20,
                                      num_threads: Optional[int] = 1):
        """u_s . n_s = u_m . n_s between two sets of tagged facets whose surfaces coincide (vertices need not align):
        python/src/dolfinx_mpc/multipointconstraint.py:435-463, cpp/ContactConstraint.h:359-503 (serial branch).
        For every dof block b in the closure of the slave facets, with n = the values of ``normal`` in that block:
        slave = the component s with the largest |n_j|; masters (a) the other components j of the same block with
        |n_j| > 1e-6, c = -n_j / n_s (compute_block_contributions, :217-280), then (b) every component k of every dof
        block m of the master-side cell the block's point lies in with |n_k / n_s * phi_m(x_b)| > 1e-6, c = that value
        (compute_master_contributions, :59-160).  A slave point that touches no master cell raises RuntimeError
        (:500-508).  No Dirichlet filtering (the reference has none here)."""
        from .fem import Function

        if isinstance(eps2, np.generic):
            eps2 = eps2.item()
        V = self.V
        bs = V.dofmap.bs
        if bs != V.mesh.tdim:
            raise RuntimeError("create_contact_slip_condition needs a vector space with block size = geometric dimension")
        if not isinstance(normal, Function) or normal.function_space.dofmap.bs != bs:
            raise ValueError("normal has to be a Function in the space of the constraint")
        hit = self._contact_collisions(meshtags, slave_marker, master_marker, eps2, False)
        if hit is None:
            return
        slave_blocks, found, basis = hit
        nrm = normal.x._data.reshape(-1, bs)[slave_blocks]  # (n, bs); not normalised (cpp/ContactConstraint.h:424-431)
        n = slave_blocks.size
        sidx = np.argmax(np.abs(nrm), axis=1)
        ns = nrm[np.arange(n), sidx]
        # (a) same block
        comp = np.arange(bs)[None, :].repeat(n, 0)
        in_a = (comp != sidx[:, None]) & (np.abs(nrm) > 1e-6)
        cnt_a = in_a.sum(axis=1)
        m_a = (slave_blocks[:, None] * bs + comp)[in_a]
        c_a = (-nrm / ns[:, None])[in_a]
        # (b) other side: val[i, j, k] = n_k / n_s * phi_j
        val = basis[:, :, None] * (nrm / ns[:, None])[:, None, :]  # (n, nd, bs): cell dof j outer, component k inner
        in_b = np.abs(val) > 1e-6
        cnt_b = in_b.reshape(n, -1).sum(axis=1)
        cell_dofs = V.dofmap.list[found].astype(np.int64)  # (n, nd)
        m_b = (cell_dofs[:, :, None] * bs + np.arange(bs)[None, None, :])[in_b]
        c_b = val[in_b]
        # concatenate per slave: (a) then (b) (impl::concatenate, :283-340)
        cnt = cnt_a + cnt_b
        offsets = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
        masters = np.empty(int(cnt.sum()), dtype=np.int64)
        coeffs = np.empty(int(cnt.sum()), dtype=np.float64)
        oa = np.concatenate([[0], np.cumsum(cnt_a)])
        ob = np.concatenate([[0], np.cumsum(cnt_b)])
        pos_a = np.repeat(offsets[:-1].astype(np.int64), cnt_a) + (np.arange(int(cnt_a.sum())) - np.repeat(oa[:-1], cnt_a))
        pos_b = np.repeat(offsets[:-1].astype(np.int64) + cnt_a, cnt_b) + (np.arange(int(cnt_b.sum())) - np.repeat(ob[:-1], cnt_b))
        masters[pos_a], coeffs[pos_a] = m_a, c_a
        masters[pos_b], coeffs[pos_b] = m_b, c_b
        slaves = slave_blocks * bs + sidx
        self.add_constraint(V, slaves.astype(np.int32), masters, coeffs, np.zeros(masters.size, dtype=np.int32), offsets)

    # -- accessors (python/src/dolfinx_mpc/multipointconstraint.py:503-584) ----
    @property
    def is_slave(self) -> np.ndarray:
        return self._h("is_slave")

    @property
    def slaves(self) -> np.ndarray:
        return self._h("slaves")

    @property
    def num_slaves(self) -> int:
        """number of distinct slave dofs (``slaves.size`` without a download)"""
        self._not_finalized()
        return self._num_slaves

    @property
    def masters(self) -> AdjacencyList:
        return AdjacencyList(self._h("masters"), self._h("moff"))

    def coefficients(self):
        return self._h("coeffs"), self._h("moff")

    @property
    def owners(self) -> AdjacencyList:
        return AdjacencyList(self._h("owners"), self._h("moff"))

    @property
    def dtype(self):
        return self._dtype

    @property
    def num_local_slaves(self) -> int:
        self._not_finalized()
        return self._num_local_slaves

    @property
    def cell_to_slaves(self) -> AdjacencyList:
        return AdjacencyList(self._h("c2s"), self._h("c2s_off"))

    @property
    def function_space(self) -> FunctionSpace:
        self._not_finalized()
        return self.V

    # -- device mirror ----------------------------------------------------------
    def _device(self):
        """(MpcT struct, keep-alive tensors, slaves tensor) on the current HIP device."""
        self._not_finalized()
        if self._dev is None:
            t = self.device_tensors()
            s = _native.MpcT(t["is_slave"].data_ptr(), t["moff"].data_ptr(), t["masters"].data_ptr(),
                             t["coeffs"].data_ptr())
            self._dev = (s, t)
        return self._dev

    # -- post-solve (python/src/dolfinx_mpc/multipointconstraint.py:586-617) ----
    def _on_device(self, u, kernel):
        """run ``kernel(device fp64 tensor)`` on u: a device ``Vector`` / torch tensor in place, or a
        ``fem.Function`` (the reference's call shape ``mpc.backsubstitution(uh)``,
        python/src/dolfinx_mpc/multipointconstraint.py:586-617) through an upload and a copy back."""
        import torch

        from .fem import Function

        if isinstance(u, Function):
            if u.function_space is not self.V:
                raise ValueError("The input function has to be in the function space in the multi-point constraint")
            dev = _native.require_gpu()
            arr = torch.from_numpy(u.x._data.astype(self._dtype, copy=False)).to(dev)
            kernel(arr)
            u.x.array[:] = arr.cpu().numpy()
            return
        arr = u.array if hasattr(u, "array") else u
        from .la import _torch_dtype

        if not isinstance(arr, torch.Tensor) or arr.dtype != _torch_dtype(self._dtype) or not arr.is_cuda:
            raise TypeError("backsubstitution / homogenize need a fem.Function, a la.Vector or a device tensor of the constraint's "
                            "scalar type")
        kernel(arr)

    def backsubstitution(self, u) -> None:
        """u[slave] = sum_k c_k u[master_k] (cpp/MultiPointConstraint.h:129-145)."""
        import torch

        s, t = self._device()

        sid = _native.scalar_id(self._dtype)

        def run(arr):
            if sid == 0:
                rc = _native.lib().mpcx_backsubstitution(arr.data_ptr(), t["slaves"].data_ptr(), t["slaves"].numel(),
                                                         C.byref(s), torch.cuda.current_stream().cuda_stream)
            else:
                assert _native.scalar_id(str(arr.dtype).replace("torch.", "")) == sid, "vector and constraint of different scalar types"
                rc = _native.lib().mpcx_backsubstitution_scalar(sid, arr.data_ptr(), t["slaves"].data_ptr(), t["slaves"].numel(),
                                                                C.byref(s), torch.cuda.current_stream().cuda_stream)
            _native.check(rc, "mpcx_backsubstitution")

        self._on_device(u, run)

    def homogenize(self, u) -> None:
        """u[slave] = 0 (cpp/MultiPointConstraint.h:147-152)."""
        import torch

        _, t = self._device()

        def run(arr):
            sid = _native.scalar_id(str(arr.dtype).replace("torch.", ""))
            if sid == 0:
                rc = _native.lib().mpcx_homogenize(arr.data_ptr(), t["slaves"].data_ptr(), t["slaves"].numel(),
                                                   torch.cuda.current_stream().cuda_stream)
            else:
                rc = _native.lib().mpcx_homogenize_scalar(sid, arr.data_ptr(), t["slaves"].data_ptr(), t["slaves"].numel(),
                                                          torch.cuda.current_stream().cuda_stream)
            _native.check(rc, "mpcx_homogenize")

        self._on_device(u, run)

    def _already_finalized(self):
        if self.finalized:
            raise RuntimeError("MultiPointConstraint has already been finalized")

    def _not_finalized(self):
        if not self.finalized:
            raise RuntimeError("MultiPointConstraint has not been finalized")
