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


class MultiPointConstraint:
    """Hold data for multi point constraint relationships.

    Args:
        V: The function space
        dtype: scalar type; only float64 is built into the HIP backend
    """

    def __init__(self, V: FunctionSpace, dtype=np.float64):
        if np.dtype(dtype) != np.float64:
            raise NotImplementedError("the HIP backend is built for float64 only")
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
        raw = dict(slaves=np.ascontiguousarray(self._slaves, dtype=np.int32),
                   masters=np.ascontiguousarray(self._masters, dtype=np.int64),
                   coeffs=np.ascontiguousarray(self._coeffs, dtype=np.float64),
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
        return self._devt

    # -- convenience builders (structured / matching meshes only) --------------
    def create_periodic_constraint_geometrical(
        self,
        V: FunctionSpace,
        indicator: Callable[[np.ndarray], np.ndarray],
        relation: Callable[[np.ndarray], np.ndarray],
        bcs: Sequence[DirichletBC],
        scale: float = 1.0,
        tol: float = 1e-8,
    ):
        """u(x_i) = scale * u(relation(x_i)) for dofs with indicator(x_i)
        (python/src/dolfinx_mpc/multipointconstraint.py:282-340).  The general
        builder (cpp/PeriodicConstraint.h) does point location and basis
        evaluation; this backend only handles meshes whose mapped slave nodes
        coincide with master nodes (one master, coefficient ``scale``)."""
        from scipy.spatial import cKDTree

        assert V is self.V
        x = V.tabulate_dof_coordinates()
        bs = V.dofmap.bs
        blocks = np.flatnonzero(np.asarray(indicator(x.T), dtype=bool))
        is_bc = np.zeros(V.num_dofs, dtype=np.int8)
        for bc in bcs:
            if V.contains(bc.function_space):  # cpp/utils.h:1470-1476: only conditions living in V
                bc.mark_dofs(is_bc)
        xm = np.asarray(relation(x[blocks].T)).T
        if blocks.size == 0:
            return
        # only dofs inside the bounding box of the mapped points can be masters
        lo, hi = xm.min(axis=0) - tol, xm.max(axis=0) + tol
        cand = np.flatnonzero(np.all((x >= lo) & (x <= hi), axis=1))
        if cand.size == 0:
            raise NotImplementedError("no dof at the mapped slave coordinates (non-matching meshes are out of scope)")
        dist, loc = cKDTree(x[cand]).query(xm)
        mblk = cand[loc]
        if blocks.size and dist.max() > tol:
            raise NotImplementedError(
                "periodic constraint on non-matching nodes needs basis evaluation at the mapped "
                "point (cpp/PeriodicConstraint.h:170-222): out of scope of this backend"
            )
        slaves = (blocks[:, None] * bs + np.arange(bs)[None, :]).reshape(-1)
        masters = (mblk[:, None] * bs + np.arange(bs)[None, :]).reshape(-1)
        # a slave BLOCK with any component under a Dirichlet condition is dropped as a whole
        # (cpp/utils.h:1459-1496 marks blocks, cpp/PeriodicConstraint.h:563-567 filters with it)
        blk_bc = is_bc.reshape(-1, bs)[blocks].any(axis=1)
        keep = np.repeat(~blk_bc, bs)
        slaves, masters = slaves[keep], masters[keep]
        n = slaves.size
        self.add_constraint(V, slaves.astype(np.int32), masters.astype(np.int64), np.full(n, scale, dtype=np.float64),
                            np.zeros(n, dtype=np.int32), np.arange(n + 1, dtype=np.int32))

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

    def create_slip_constraint(self, V: FunctionSpace, blocks: np.ndarray, normals: np.ndarray,
                               bcs: Sequence[DirichletBC] = ()):
        """u.n = 0 on the given dof blocks (cpp/SlipConstraint.h:115-166): slave =
        component with the largest |n_i|, masters = the other components of the
        same block, c_i = -n_i / n_s (no tolerance filter on zero coefficients)."""
        assert V is self.V
        bs = V.dofmap.bs
        is_bc = np.zeros(V.num_dofs, dtype=np.int8)
        for bc in bcs:
            bc.mark_dofs(is_bc)
        slaves, masters, coeffs, offsets = [], [], [], [0]
        for blk, n in zip(np.asarray(blocks), np.asarray(normals)):
            s = int(np.argmax(np.abs(n[:bs])))
            sd = blk * bs + s
            if is_bc[sd]:
                continue
            slaves.append(sd)
            for k in range(bs):
                if k != s:
                    masters.append(blk * bs + k)
                    coeffs.append(-n[k] / n[s])
            offsets.append(len(masters))
        self.add_constraint(V, np.array(slaves, dtype=np.int32), np.array(masters, dtype=np.int64),
                            np.array(coeffs, dtype=np.float64), np.zeros(len(masters), dtype=np.int32),
                            np.array(offsets, dtype=np.int32))

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
            return
        mfac = meshtags.find(master_marker)
        if mfac.shape[0] == 0:
            if allow_missing_masters:
                return
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
            arr = torch.from_numpy(u.x._data).to(dev)
            kernel(arr)
            u.x.array[:] = arr.cpu().numpy()
            return
        arr = u.array if hasattr(u, "array") else u
        if not isinstance(arr, torch.Tensor) or arr.dtype != torch.float64 or not arr.is_cuda:
            raise TypeError("backsubstitution / homogenize need a fem.Function, a la.Vector or a float64 device tensor")
        kernel(arr)

    def backsubstitution(self, u) -> None:
        """u[slave] = sum_k c_k u[master_k] (cpp/MultiPointConstraint.h:129-145)."""
        import torch

        s, t = self._device()

        def run(arr):
            rc = _native.lib().mpcx_backsubstitution(arr.data_ptr(), t["slaves"].data_ptr(), t["slaves"].numel(),
                                                     C.byref(s), torch.cuda.current_stream().cuda_stream)
            _native.check(rc, "mpcx_backsubstitution")

        self._on_device(u, run)

    def homogenize(self, u) -> None:
        """u[slave] = 0 (cpp/MultiPointConstraint.h:147-152)."""
        import torch

        _, t = self._device()

        def run(arr):
            rc = _native.lib().mpcx_homogenize(arr.data_ptr(), t["slaves"].data_ptr(), t["slaves"].numel(),
                                               torch.cuda.current_stream().cuda_stream)
            _native.check(rc, "mpcx_homogenize")

        self._on_device(u, run)

    def _already_finalized(self):
        if self.finalized:
            raise RuntimeError("MultiPointConstraint has already been finalized")

    def _not_finalized(self):
        if not self.finalized:
            raise RuntimeError("MultiPointConstraint has not been finalized")
