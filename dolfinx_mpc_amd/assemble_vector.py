"""``assemble_vector`` / ``apply_lifting`` with the reference's signatures
(python/src/dolfinx_mpc/assemble_vector.py:25-147) on the HIP backend."""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

from .common import timed
from . import _device as D
from . import _native
from .fem import DirichletBC, Form
from .la import Vector, create_vector
from .multipointconstraint import MultiPointConstraint


_ALG = {"auto": 0, "atomic": 1, "rowblock": 2}
# rows of b one workgroup owns in LDS (8 B each); the entity lists follow the matrix plan's shape
VECTOR_BLOCK_ROWS = int(os.environ.get("MPCX_VECTOR_BLOCK_ROWS", 512))


# P2 with a tile-wise numbering and a many-point rule: large row blocks keep the halo (entities evaluated by more
# than one block) small -- 24-point source on 160^3: hash kernel 3.30 ms, row blocks of 512 rows 3.65, 2048 2.50,
# 4096 2.36, 8192 2.34, 12288 2.73
VECTOR_BLOCK_ROWS_P2 = int(os.environ.get("MPCX_VECTOR_BLOCK_ROWS_P2", 8192))


def _vector_plan(form: Form, i: int, V, rows: int = VECTOR_BLOCK_ROWS):
    """Row blocks of b and the entities touching each (mpcx_rowblock_plan_build on a
    one-entry-per-row pattern), cached per integral."""
    key = ("vplan", i, rows)
    if key not in form._device:
        from .assemble_matrix import _block_lists_device, _block_ranges

        integ = form.integrals[i]
        nrows = V.num_dofs
        rowptr = np.arange(nrows + 1, dtype=np.int64)
        hints = None
        if V.dof_tile_offsets is not None:
            hints = np.ascontiguousarray(V.dof_tile_offsets.astype(np.int32) * V.dofmap.bs)
        row0 = _block_ranges(nrows, rowptr, rows, rows, V.dofmap.bs, hints)
        nb = row0.size - 1
        dev = _native.require_gpu()
        t = _block_lists_device(row0, integ.num_entities, integ.estride, D.integral_device(form, i)["entities_ptr"],
                                D.space_device(V)["dofmap"], V.element_ndofs, V.dofmap.bs, dev)
        max_rows = int(np.diff(row0).max()) if nb > 0 else 0
        plan = _native.RowBlockPlanT(nb, max_rows, max_rows, 0, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(),
                                     None, None)
        form._device[key] = (plan, t)
    return form._device[key]


VECTOR_LDS_ROWS = 96 * 1024 // 8  # rows of b one workgroup can hold (own + halo)
VECTOR_OWNER_ROWS = int(os.environ.get("MPCX_VECTOR_OWNER_ROWS", 8192))  # own rows per block (3072: 8.0 ms, 4096-8192: 5.6-5.7 ms at config 5)


def _even_rows(V, cap: int) -> int:
    """rows per block <= cap that cut a tile of the numbering into equal parts (or take whole tiles): the block
    builder never crosses a tile start, so 8192 on tiles of 12288 rows gives blocks of 8192 and 4096 in turn
    (Stokes b0: 2.8 ms instead of 1.5 with 6144 + 6144)"""
    if V.dof_tile_offsets is None or V.dof_tile_offsets.size < 2:
        return cap
    tile = int(V.dof_tile_offsets[1] - V.dof_tile_offsets[0]) * V.dofmap.bs
    if tile <= 0:
        return cap
    if tile <= cap:
        return (cap // tile) * tile
    parts = -(-tile // cap)
    return -(-tile // parts)


def _vector_owner_plan(form: Form, i: int, V, md0, rows: int):
    """Owner-computes plan of the row-block vector kernel (include/mpcx.h, mpcx_vector_args_t::own_*), built on the
    device with torch (plumbing: gathers, searchsorted, sorts).  Every entity belongs to the block that holds the rows
    of its local dof 0; the dofs of other blocks its entities touch are the block's halo, appended to its LDS copy.
    ``md0``: the slave-masked dofmap (its flag bits move into the position table).  Returns None when a block with
    its halo does not fit the LDS budget."""
    return D.cached(form._device, "voplan", (md0,), (i, rows), lambda: _build_vector_owner_plan(form, i, V, md0, rows),
                    maxsize=4)


def _build_vector_owner_plan(form: Form, i: int, V, md0, rows: int):
    integ = form.integrals[i]
    n, nd = integ.num_entities, V.element_ndofs
    ents = D.integral_device(form, i)["entities"]
    cells = None if ents is None else ents.view(n, integ.estride)[:, 0].long()
    mrow = md0.view(-1, nd)[:n] if cells is None else md0.view(-1, nd)[cells]  # masked dofmap rows of the entities
    return _owner_plan_from_rows(mrow, V, rows)


def _owner_plan_from_rows(mrow, V, rows: int):
    """owner-computes plan for ``n`` work items whose masked dof rows are ``mrow`` (n, nd) (dof | flags << 28): the
    entities of an integral, or the cell clusters of the mesh with their eight vertices.  Built through the C ABI
    (include/mpcx.h mpcx_owner_plan_*: three fused passes over the table; rocPRIM scans / sorts between them); torch
    allocates.  ``MPCX_OWNER_PLAN=torch``: the same plan from torch gathers / searches (kept as the cross-check)."""
    import torch

    from . import _prims
    from .assemble_matrix import _block_ranges

    n, nd = mrow.shape
    if os.environ.get("MPCX_OWNER_PLAN", "") == "torch" or n * nd >= 2 ** 31 or n == 0:
        return _owner_plan_from_rows_torch(mrow, V, rows)
    bs = V.dofmap.bs
    nrows = V.num_dofs
    hints = None
    if V.dof_tile_offsets is not None:
        hints = np.ascontiguousarray(V.dof_tile_offsets.astype(np.int32) * bs)
    row0 = _block_ranges(nrows, None, rows, rows, bs, hints)
    nb = row0.size - 1
    dev = _native.require_gpu()
    L = _native.lib()
    st = D.stream_ptr()
    d_row0 = D._to_dev(row0, dev)
    mrow = mrow.contiguous()
    i32, i64 = torch.int32, torch.int64
    owner = torch.empty(n, dtype=i64, device=dev)
    item = torch.empty(n, dtype=i32, device=dev)
    fcount = torch.empty(n, dtype=i32, device=dev)
    _native.check(L.mpcx_owner_plan_count(n, nd, mrow.data_ptr(), bs, nb, d_row0.data_ptr(), owner.data_ptr(), item.data_ptr(),
                                          fcount.data_ptr(), st), "mpcx_owner_plan_count")
    foff = _prims.scan_i32_i64(fcount)
    nf = int(foff[-1].item())
    del fcount
    sorted_owner, order = _prims.sort_pairs(owner, item, max(int(nb).bit_length(), 1))
    off = _prims.segment_offsets(sorted_owner, 0, nb)
    del sorted_owner, item
    keys = torch.empty(max(nf, 1), dtype=i64, device=dev)
    src = torch.empty(max(nf, 1), dtype=i32, device=dev)
    lmap = torch.empty((n, nd), dtype=i32, device=dev)
    _native.check(L.mpcx_owner_plan_keys(n, nd, mrow.data_ptr(), bs, nb, d_row0.data_ptr(), owner.data_ptr(), foff.data_ptr(),
                                         keys.data_ptr(), src.data_ptr(), lmap.data_ptr(), st), "mpcx_owner_plan_keys")
    del owner, foff
    keys, src = _prims.sort_pairs(keys[:nf], src[:nf], 32 + max(int(nb).bit_length(), 1))
    ukey, _, heads, hscan = _prims.runs(keys, want_keys=True, want_marks=True)
    nu = ukey.numel()
    hoff = _prims.segment_offsets(ukey, 32, nb)
    max_rows_d = torch.zeros(1, dtype=i32, device=dev)
    _native.check(L.mpcx_owner_plan_halo(nf, D.ptr(keys), D.ptr(src), D.ptr(heads), D.ptr(hscan), hoff.data_ptr(), nb,
                                         d_row0.data_ptr(), bs, mrow.data_ptr(), lmap.data_ptr(), max_rows_d.data_ptr(), st),
                  "mpcx_owner_plan_halo")
    max_rows = int(max_rows_d.item())
    if max_rows > VECTOR_LDS_ROWS:
        return None
    del keys, src, heads, hscan
    # spill order: the halo entries sorted by dof (stable); runs of one dof are reduced into its row of b
    low = torch.empty(max(nu, 1), dtype=i64, device=dev)
    iota = torch.empty(max(nu, 1), dtype=i32, device=dev)
    _native.check(L.mpcx_low_word_iota(nu, D.ptr(ukey), low.data_ptr(), iota.data_ptr(), st), "mpcx_low_word_iota")
    sdof, sorder = _prims.sort_pairs(low[:nu], iota[:nu], max(int(nrows // bs).bit_length(), 1))
    urows64, seg = _prims.runs(sdof, want_keys=True)
    urows = urows64.to(i32).contiguous()
    spill = torch.zeros(max(nu, 1) * bs, dtype=torch.float64, device=dev)
    t = (d_row0, off, order, lmap, hoff, spill, sorder, urows, seg)
    plan = _native.RowBlockPlanT(nb, max_rows, max_rows, 0, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), None, None)
    return (plan, t, int(urows.numel()))


def _owner_plan_from_rows_torch(mrow, V, rows: int):
    """the owner-computes plan from torch gathers / searchsorted / unique (cross-check of the C-ABI builder; also taken
    when item * nd + i does not fit 32 bits)"""
    import torch

    from .assemble_matrix import _block_ranges

    n, nd = mrow.shape
    bs = V.dofmap.bs
    nrows = V.num_dofs
    ndof_blocks = nrows // bs
    hints = None
    if V.dof_tile_offsets is not None:
        hints = np.ascontiguousarray(V.dof_tile_offsets.astype(np.int32) * bs)
    row0 = _block_ranges(nrows, None, rows, rows, bs, hints)
    nb = row0.size - 1
    dev = _native.require_gpu()
    d_row0 = D._to_dev(row0, dev)
    dof = (mrow & ((1 << 28) - 1)).to(torch.int64)
    flags = (mrow >> 28) << 28
    blk = torch.searchsorted(d_row0[1:].contiguous(), (dof * bs).contiguous(), right=True)  # (n, nd)
    owner = blk[:, 0].contiguous()
    order = torch.argsort(owner, stable=True)
    off = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
    torch.cumsum(torch.bincount(owner, minlength=nb), 0, out=off[1:])
    foreign = blk != owner[:, None]
    del blk
    first = (d_row0[:-1].to(torch.int64) // bs)  # first dof of every block
    nown = (d_row0[1:] - d_row0[:-1]).to(torch.int64) // bs
    fe, fi = torch.nonzero(foreign, as_tuple=True)
    ukey, inv = torch.unique(owner[fe] * ndof_blocks + dof[fe, fi], return_inverse=True)  # (block, halo dof), sorted
    hblk, hdof = ukey // ndof_blocks, ukey % ndof_blocks
    hoff = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
    torch.cumsum(torch.bincount(hblk, minlength=nb), 0, out=hoff[1:])
    max_rows = int(((nown + (hoff[1:] - hoff[:-1])) * bs).max().item()) if nb > 0 else 0
    if max_rows > VECTOR_LDS_ROWS:
        return None
    lmap = dof - first[owner][:, None]
    lmap[fe, fi] = nown[owner[fe]] + (inv - hoff[hblk[inv]])
    lmap = (lmap.to(torch.int32) | flags.to(torch.int32)).contiguous()
    sdof, sorder = torch.sort(hdof, stable=True)
    urows, counts = torch.unique_consecutive(sdof, return_counts=True)
    seg = torch.zeros(urows.numel() + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=seg[1:])
    spill = torch.zeros(max(ukey.numel(), 1) * bs, dtype=torch.float64, device=dev)
    t = (d_row0, off, order.to(torch.int32).contiguous(), lmap, hoff, spill, sorder.to(torch.int32).contiguous(),
         urows.to(torch.int32).contiguous(), seg)
    plan = _native.RowBlockPlanT(nb, max_rows, max_rows, 0, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), None, None)
    return (plan, t, int(urows.numel()))


# own rows per block of the owner-computes cluster vector kernel (config 2, kernel ms: 1024 rows 2.73, 2048 2.75, 4096 2.81, 8192 4.40)
VCUBE_OWNER_ROWS = int(os.environ.get("MPCX_VCUBE_ROWS", 2048))
VCUBE_OWN_THREADS = 256  # threads per workgroup of vector_cube_own_kernel (csrc/mpcx_cubes.hip)


def _vcube_owner_rows(V) -> int:
    """own rows per block of the cluster vector kernel: MPCX_VCUBE_ROWS (default 2048), halved for small problems (the slab
    one rank holds at 8 GPUs) until there are row blocks for several rounds over the 256 CUs -- 128^3 cubes: 2048 / 1024 /
    512 rows 0.44 / 0.40 / 0.39 ms; 256^3 keeps 2048"""
    top = VCUBE_OWNER_ROWS
    if "MPCX_VCUBE_ROWS" not in os.environ:
        while top > 512 and V.num_dofs // top < 4096:
            top //= 2
    return top


def _vector_cube_owner_plan(mesh, V, d_verts, constraint, left: np.ndarray, slave_ents_h: np.ndarray):
    """owner-computes plan over the mesh's cell clusters (work item = cluster, its eight vertices = its dofs; slave
    flag folded into the vertex ids) + the slave CELLS the cluster call is responsible for (all but the leftover
    ones, which the per-cell call handles); cached per (clusters, constraint)"""
    import torch

    def build():
        _, t = constraint._device()
        flag = t["is_slave"][d_verts.long()].to(torch.int32) << 28
        mrow = (d_verts | flag).contiguous()
        top = _vcube_owner_rows(V)
        own = None
        if top < VCUBE_OWNER_ROWS:
            # a small problem: the cap that needs the fewest 256-thread passes over the blocks' clusters (a slab with a
            # ghost plane has tiles of 448 rows -- 273 clusters per 512-row block, two passes with 17 lanes in the second:
            # 512 / 1024 / 2048 rows 0.65 / 0.65 / 0.53 ms on rank 1 of the 8-way cut of config 2, 0.46 / 0.50 / 0.51 on rank 0);
            # ties go to the smaller blocks
            best = None
            for rows in (512, 1024, 2048):
                cand = _owner_plan_from_rows(mrow, V, _even_rows(V, rows))
                if cand is None:
                    continue
                per = cand[1][1][1:] - cand[1][1][:-1]  # clusters per block
                passes = int(((per + (VCUBE_OWN_THREADS - 1)) // VCUBE_OWN_THREADS).sum().item())
                if best is None or passes < best[0]:
                    best = (passes, rows, cand)
            if best is not None:
                own, V._vcube_rows = best[2], best[1]
        else:
            for rows in (top, top // 2, top // 4):
                own = _owner_plan_from_rows(mrow, V, _even_rows(V, rows))
                if own is not None:
                    V._vcube_rows = rows
                    break
        if own is None:
            return None
        sl = slave_ents_h if left.size == 0 else np.setdiff1d(slave_ents_h, left)
        return own + (D._to_dev(np.ascontiguousarray(sl, dtype=np.int32), d_verts.device),)

    return D.cached(mesh._device, "vcube_own", (d_verts, constraint, left), VCUBE_OWNER_ROWS, build, maxsize=2)


def _cluster_grid(mesh, d_verts):
    """The tensor grid under a mesh of axis-aligned box clusters (include/mpcx.h mpcx_vector_args_t::grid_*): per axis the
    distinct intervals (coordinate of corner 0, coordinate of corner 7) of the clusters and, per cluster, which interval it
    sits on along x, y, z.  Returns (fraction of clusters that are boxes, grid); the grid is None when a cluster is not a
    box with its eight vertices in corner order (compared exactly, as the kernel's own per-cluster check does), or when the
    intervals are not few against the clusters (no tensor structure: the tables would cost what they save).  Geometry only -- cached per (clusters, geometry version); every value of the
    right-hand side is still computed inside each launch."""
    import torch

    def build():
        x = D.mesh_device(mesh)["x"].view(-1, 3)
        v = d_verts.long()
        n = v.shape[0]
        X0, X7 = x[v[:, 0]], x[v[:, 7]]
        box = torch.ones(n, dtype=torch.bool, device=x.device)
        for c in range(1, 7):
            xc = x[v[:, c]]
            for d in range(3):
                box &= xc[:, d] == (X7[:, d] if (c >> d) & 1 else X0[:, d])
        frac = float(box.sum(dtype=torch.int64).item()) / max(n, 1)
        if frac < 1.0:
            return frac, None
        idx = torch.zeros((n, 4), dtype=torch.int32, device=x.device)
        ivs, ns = [], []
        for d in range(3):
            lo_u, lo_i = torch.unique(X0[:, d], return_inverse=True)
            hi_u, hi_i = torch.unique(X7[:, d], return_inverse=True)
            pair_u, pair_i = torch.unique(lo_i * hi_u.numel() + hi_i, return_inverse=True)
            idx[:, d] = pair_i.to(torch.int32)
            ivs.append(torch.stack([lo_u[pair_u // hi_u.numel()], hi_u[pair_u % hi_u.numel()]], dim=1))
            ns.append(int(pair_u.numel()))
        if sum(ns) > max(4096, n // 8):
            return frac, None
        return frac, (idx.contiguous(), torch.cat(ivs).contiguous(), tuple(ns))

    return D.cached(mesh._device, "vcube_grid", (d_verts,), (mesh.geometry.version,), build, maxsize=2)


GRID_BLOCK_ROWS = 128  # MPCX_GRID_BLOCK_ROWS of include/mpcx.h


def _block_rows(pk, idx, ns):
    """Per block of an owner plan the table rows its clusters need, and per cluster the positions of its three rows in the
    list of its block (mpcx_vector_args_t::grid_block_rows): (block_rows [nb][128] int32, local idx [n][4] int32), or None
    when a block needs more than 128 rows (no tiles in the numbering: the clusters then read the table itself)."""
    import torch

    off, ents = pk[1], pk[2]
    nb = off.numel() - 1
    if nb <= 0 or ents.numel() == 0:
        return None
    dev = ents.device
    blk = torch.repeat_interleave(torch.arange(nb, device=dev), (off[1:] - off[:-1]).long())
    of_cluster = torch.zeros(idx.shape[0], dtype=torch.int64, device=dev)
    of_cluster[ents.long()] = blk
    ntot = sum(ns)
    key = torch.cat([of_cluster * ntot + (idx[:, d].long() + sum(ns[:d])) for d in range(3)])  # (block, table row) pairs
    uniq, inv = torch.unique(key, return_inverse=True)  # sorted: the rows of a block are consecutive, x then y then z
    ublk = uniq // ntot
    counts = torch.bincount(ublk, minlength=nb)
    if int(counts.max().item()) > GRID_BLOCK_ROWS:
        return None
    starts = torch.cumsum(counts, 0) - counts
    rows = torch.full((nb, GRID_BLOCK_ROWS), -1, dtype=torch.int32, device=dev)
    rows[ublk, torch.arange(uniq.numel(), device=dev) - starts[ublk]] = (uniq % ntot).to(torch.int32)
    n = idx.shape[0]
    local = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    for d in range(3):
        local[:, d] = (inv[d * n:(d + 1) * n] - starts[of_cluster]).to(torch.int32)
    return rows.contiguous(), local.contiguous(), int(counts.max().item())


def rule_subset_table(qpts):
    """For a quadrature rule on the reference tetrahedron: the distinct sums of the barycentric coordinates of a point over a
    proper vertex subset (``eta``, ascending) and ``J[q][m]`` = which of them point q has for subset m (bit v = vertex v, vertex 0
    = the origin; m = 0 and 15 unused).  A cell of a box mesh has, along every axis, some of its vertices on the low and the
    others on the high side of its interval: the coordinate of point q is lo + (hi - lo) eta[J[q][m]], m = the high ones.
    None for rules with more than 250 distinct sums (one byte per entry)."""
    q = np.asarray(qpts, dtype=np.float64).reshape(-1, 3)
    lam = np.concatenate([1.0 - q.sum(axis=1, keepdims=True), q], axis=1)  # (nq, 4), vertex 0 first
    sub = np.array([[(m >> v) & 1 for v in range(4)] for m in range(16)], dtype=np.float64)  # (16, 4)
    inner = (lam @ sub.T)[:, 1:15]
    eta = np.unique(np.round(inner.ravel(), 13))
    if eta.size > 250:
        return None
    J = np.zeros((q.shape[0], 16), dtype=np.uint8)
    J[:, 1:15] = np.abs(inner[..., None] - eta).argmin(axis=-1)
    eta = np.array([inner[J[:, 1:15] == j].mean() for j in range(eta.size)])
    return eta, J


def _cell_grid(form, i, V, pk):
    """The per-cell twin of ``_cluster_grid`` (include/mpcx.h mpcx_vector_args_t::grid_eta / grid_J): for a scalar P1 / P2 source
    form over ALL cells of a tetrahedral mesh whose cells have their vertices on two values per axis (every cell of a box mesh),
    the tensor grid of intervals, per cell its three table rows (positions in the list of the block of the owner plan ``pk``
    that evaluates it) and which vertices lie on the high side per axis, the distinct sums of barycentric coordinates the rule
    produces (``eta``) and the table (point, vertex subset) -> eta.  None when a cell is not of that kind, the intervals are not
    few, or a block needs more than GRID_BLOCK_ROWS rows.  Geometry + rule only; cached per (plan, geometry version)."""
    import torch

    mesh = form.mesh
    k = form.integrals[i].kernel

    def build():
        md = D.mesh_device(mesh)
        x = md["x"].view(-1, 3)
        dm = md["x_dofmap"].view(-1, 4)
        n = dm.shape[0]
        table = rule_subset_table(k.qpts)
        if table is None:
            return None
        eta, J = table
        # axis by axis (a few n-vectors at a time: config 5 holds 89 M cells, and the counters of bench.py are taken by a second
        # process beside the first)
        idx = torch.zeros((n, 4), dtype=torch.int32, device=x.device)
        masks, ivs, ns = [], [], []
        vids = [dm[:, v].long() for v in range(4)]
        for d in range(3):
            xd = x[:, d].contiguous()
            c = [xd[vids[v]] for v in range(4)]
            lo = torch.minimum(torch.minimum(c[0], c[1]), torch.minimum(c[2], c[3]))
            hi = torch.maximum(torch.maximum(c[0], c[1]), torch.maximum(c[2], c[3]))
            m = torch.zeros(n, dtype=torch.int32, device=x.device)
            ok = hi > lo
            for v in range(4):
                on_hi = c[v] == hi
                ok &= on_hi | (c[v] == lo)
                m |= on_hi.to(torch.int32) << v
            if not bool(ok.all()):
                return None
            del c, ok
            masks.append(m)
            lo_u, lo_i = torch.unique(lo, return_inverse=True)
            hi_u, hi_i = torch.unique(hi, return_inverse=True)
            del lo, hi
            pair_u, pair_i = torch.unique(lo_i * hi_u.numel() + hi_i, return_inverse=True)
            del lo_i, hi_i
            idx[:, d] = pair_i.to(torch.int32)
            del pair_i
            ivs.append(torch.stack([lo_u[pair_u // hi_u.numel()], hi_u[pair_u % hi_u.numel()]], dim=1))
            ns.append(int(pair_u.numel()))
        del vids
        if sum(ns) > max(4096, n // 32):
            return None
        # |det J| / (h_x h_y h_z) = |det| of the 0 / 1 matrix "vertex v on the high side of axis d" (edges from vertex 0):
        # 1 for the tetrahedra of a Kuhn cut, 2 for the central one of a five-cell cut, 0 = flat
        def col(d):
            b = [((masks[d] >> v) & 1) for v in range(4)]
            return [b[v] - b[0] for v in (1, 2, 3)]

        a_, b_, c_ = col(0), col(1), col(2)  # a_[v]: entry (edge v, axis 0) ...
        det = (a_[0] * (b_[1] * c_[2] - b_[2] * c_[1]) - a_[1] * (b_[0] * c_[2] - b_[2] * c_[0]) + a_[2] * (b_[0] * c_[1] - b_[1] * c_[0])).abs()
        del a_, b_, c_
        if not bool((det >= 1).all()):
            return None
        cfac = det
        staged = _block_rows(pk, idx, tuple(ns))
        if staged is None:
            return None
        rows, local, longest = staged
        # cell types: which vertices lie on the high side per axis (+ the determinant factor); per type and point the three
        # indices into eta in one word
        if eta.size > 255:
            return None
        tkey = masks[0] | (masks[1] << 4) | (masks[2] << 8)
        types, tid = torch.unique(tkey, return_inverse=True)
        if types.numel() * J.shape[0] > 4096:
            return None
        local[:, 3] = tid.to(torch.int32) | (cfac << 16)
        th = types.cpu().numpy().astype(np.int64)
        Jw = (J[:, th & 15].astype(np.uint32) | (J[:, (th >> 4) & 15].astype(np.uint32) << 8)
              | (J[:, (th >> 8) & 15].astype(np.uint32) << 16)).T.copy()  # (ntypes, nq)
        ngp = (eta.size + 1) & ~1
        tab = torch.empty(sum(ns) * (2 * ngp + 2), dtype=torch.float64, device=x.device)
        return dict(rec=local.contiguous(), rows=rows, longest=longest, iv=torch.cat(ivs).contiguous(), ns=tuple(ns), tab=tab,
                    eta=D._to_dev(eta, x.device), J=D._to_dev(Jw.reshape(-1).view(np.int32), x.device), ng=int(eta.size),
                    ntypes=int(types.numel()))

    return D.cached(form._device, "cell_grid", (pk,), (i, mesh.geometry.version), build, maxsize=2)


def _grid_rule(k) -> bool:
    """does the kernel data hold the 14-point rule the tensor-grid tables are generated for (csrc/mpcx_box14.hpp)?"""
    from .quadrature import make_quadrature

    if k.qwts is None or k.qwts.size != 14:
        return False
    p, w = make_quadrature("tetrahedron", 5)
    return bool(np.array_equal(np.asarray(k.qpts).reshape(-1), p.reshape(-1)) and np.array_equal(np.asarray(k.qwts), w))


GRID_ROW = 40  # MPCX_GRID_ROW of include/mpcx.h


def vector_args(form: Form, i: int, b: Vector, constraint: MultiPointConstraint, alg: int, allow_cubes: bool = True):
    """Fill the C-ABI argument block of ``mpcx_assemble_vector`` for integral i; returns (args, keep-alive)."""
    V = form.function_spaces[0]
    integ = form.integrals[i]
    md = D.mesh_device(form.mesh)
    sd = D.space_device(V)
    m, mkeep = constraint._device()
    idv = D.integral_device(form, i)
    a = _native.VectorArgs()
    wt = getattr(b, "_write_through", None)  # (the locality twin of a caller's vector: mpcx_vector_args_t::row_map)
    if wt is None:
        a.b, a.num_dofs = b.array.data_ptr(), b.size
    else:
        a.b, a.num_dofs, a.row_map = wt[0].array.data_ptr(), b.size, wt[1].data_ptr()
    a.kernel = idv["kernel"]
    a.x, a.x_dofmap, a.nv = md["x"].data_ptr(), md["x_dofmap"].data_ptr(), form.mesh.geometry.dofmap.shape[1]
    a.estride, a.n_entities = integ.estride, integ.num_entities
    a.entities = a.entities0 = idv["entities_ptr"]
    a.coeffs = D.ptr(idv["coeffs"])
    a.cstride = integ.cstride
    a.constants = D.ptr(idv["constants"])
    a.dofmap, a.nd, a.bs = sd["dofmap"].data_ptr(), V.element_ndofs, V.dofmap.bs
    a.mpc = m
    a.cell_info0 = D.cell_info_ptr(V, integ.kernel)
    a.algorithm = 1
    keep = [md, sd, mkeep, idv]
    a.leftover = None  # (python attribute) cells the cluster kernel does not cover
    a.second = None  # (python attribute) a follow-up call (hexahedra: slave rows through the imported kernel)
    k = integ.kernel
    a.stream = D.stream_ptr()
    a.kernel_name = "atomic"  # (python attribute) the table entry that was taken
    if alg == 1 or integ.num_entities == 0:
        return a, keep  # thread-per-entity kernels: LDS hash + device atomics (built-in), plain atomics (imported)
    from . import dispatch
    from .assemble_matrix import _masked_dofmap, _slave_entities

    nq = int(k.qwts.size if integ.itype == "cell" else k.fqwts.size)
    tiled = V.dof_tile_offsets is not None
    ctx = dispatch.Ctx(form=k.form, tet=k.celltype == 2, d0=V.degree, bs0=V.dofmap.bs, d1=V.degree, bs1=V.dofmap.bs,
                       nd0=V.element_ndofs, nd1=V.element_ndofs, nq=nq, cell_integral=integ.itype == "cell",
                       has_coefficient=integ.coefficient is not None, coeff_degree=k.coeff_degree,
                       all_cells=idv["entities_ptr"] is None, p1_geometry=sd["dofmap"] is md["x_dofmap"], same=True, tiled=tiled,
                       builtin_form=(k.builtin.form if getattr(k, "builtin", None) is not None else -1),
                           has_transforms=getattr(k, "ufcx_transforms", None) is not None)
    # row-block shapes (rows of b one workgroup holds): blocked spaces get the same number of NODES per block (vector P1,
    # contact benchmark: 0.43 -> 0.29 ms); scalar P2 sources with the basis table on a tiled numbering and many-point
    # rules take large blocks (the halo is paid in arithmetic)
    nq_max = 8 if (V.degree == 2 and tiled) else 4
    p2_fast = V.degree == 2 and tiled and k.form == 2 and k.coeff_degree == 0 and integ.itype == "cell"
    ufcx = k.form == 100
    heavy = ufcx or nq > nq_max  # an imported kernel's cost is unknown: treated as expensive
    rows = VECTOR_BLOCK_ROWS if "MPCX_VECTOR_BLOCK_ROWS" in os.environ else VECTOR_BLOCK_ROWS * V.dofmap.bs
    nrows_blk = VECTOR_BLOCK_ROWS_P2 if (p2_fast and heavy) else rows
    if heavy and not p2_fast and (tiled or ufcx):
        nrows_blk = max(nrows_blk, 2048 * V.dofmap.bs)
    names = dispatch.candidates(dispatch.VECTOR, ctx, "vector", plan_only=(alg == 2))
    if _native.scalar_id(getattr(form, "dtype", np.float64)) != 0:
        if k.form == 100:
            raise NotImplementedError("imported (UFCx) kernels are fp64-real")
        names = ["rowblock"]  # the one row-block formulation csrc/mpcx_scalar.hip restates over a scalar type
    for name in names:
        if name == "hex_own":
            # hexahedra: one thread per cell through the built-in Q1 source kernel (MPCX_ALG_CUBE with the built-in twin of
            # the imported kernel, owner-computes row blocks over the cell dofmap), then the rows of slave dofs through
            # the imported kernel over the slave cells -- a second call without bulk entities; "auto" only
            if alg != 0 or not allow_cubes:
                continue
            d_verts = sd.get("cell_verts")  # (ONE view object per space: the plan cache is keyed by identity)
            if d_verts is None or d_verts.shape[0] != integ.num_entities:
                d_verts = sd["cell_verts"] = sd["dofmap"].view(-1, 8)[: integ.num_entities]
            left = getattr(form.mesh, "_no_leftover", None)
            if left is None:
                left = form.mesh._no_leftover = np.zeros(0, dtype=np.int32)
            slave_h, _ = _slave_entities(form, i, constraint, constraint)
            own = _vector_cube_owner_plan(form.mesh, V, d_verts, constraint, left, slave_h)
            if own is None:
                continue
            plan, pk, n_own, d_slaves = own
            t = _native.VectorArgs.from_buffer_copy(a)  # the imported kernel's call: slave rows only
            t.algorithm, t.n_entities = 2, 0
            t.slave_entities, t.n_slave_entities = d_slaves.data_ptr(), d_slaves.numel()
            t.leftover, t.kernel_name, t.second = None, name, None
            a.kernel = idv["kernel_builtin"]
            a.plan = plan
            a.own_lmap, a.own_hoff, a.own_spill = pk[3].data_ptr(), pk[4].data_ptr(), pk[5].data_ptr()
            a.own_src, a.own_rows, a.own_seg, a.n_own_rows = pk[6].data_ptr(), pk[7].data_ptr(), pk[8].data_ptr(), n_own
            a.algorithm = 3
            a.cube_verts, a.n_cubes = d_verts.data_ptr(), d_verts.shape[0]
            a.kernel_name = name
            a.second = t if d_slaves.numel() > 0 else None
            keep += [pk, d_slaves, d_verts]
            return a, keep
        if name in ("cube_own", "cube_hash", "ufcx_cube_own"):
            # scalar P1 source over all cells: one thread per cell cluster (MPCX_ALG_CUBE, csrc/mpcx_cubes.hip); "auto" only
            if alg != 0 or not allow_cubes:
                continue
            from .clusters import mesh_clusters_device, mesh_clusters_ordered_device

            if name == "ufcx_cube_own":
                # imported kernel: only clusters whose cells the mesh lists in the cluster kernels' own vertex order
                d_verts, left, d_cells = mesh_clusters_ordered_device(form.mesh, integ.num_entities)
                if integ.cstride > 0:
                    a.cube_cells = d_cells.data_ptr()
                    keep += [d_cells]
            else:
                d_verts, left = mesh_clusters_device(form.mesh, integ.num_entities)
            if d_verts.shape[0] * 6 < 0.5 * integ.num_entities:
                continue
            if name in ("cube_own", "ufcx_cube_own"):
                # owner-computes row blocks over the clusters: no hash table, no device atomics, deterministic
                slave_h, _ = _slave_entities(form, i, constraint, constraint)
                own = _vector_cube_owner_plan(form.mesh, V, d_verts, constraint, left, slave_h)
                if own is None:
                    continue
                plan, pk, n_own, d_slaves = own
                a.plan = plan
                a.own_lmap, a.own_hoff, a.own_spill = pk[3].data_ptr(), pk[4].data_ptr(), pk[5].data_ptr()
                a.own_src, a.own_rows, a.own_seg, a.n_own_rows = pk[6].data_ptr(), pk[7].data_ptr(), pk[8].data_ptr(), n_own
                a.slave_entities, a.n_slave_entities = d_slaves.data_ptr(), d_slaves.numel()
                keep += [pk, d_slaves]
                if (name == "cube_own" and k.form == 2 and k.fn_id == 1 and k.coeff_degree == 0 and integ.coefficient is None
                        and os.environ.get("MPCX_BOX_GRID", "1") != "0" and _grid_rule(k)):
                    # the benchmark's right-hand side on box clusters of a tensor grid: its univariate factors once per
                    # interval and launch instead of 84 sines and exponentials per cluster (csrc/mpcx_cubes.hip)
                    boxes, grid = _cluster_grid(form.mesh, d_verts)
                    # (a mesh with few boxes takes the lighter instance that evaluates every cluster point by point)
                    a.cube_boxes = int(boxes >= 0.25)
                    if grid is not None and os.environ.get("MPCX_TENSOR_GRID", "1") != "0":
                        import torch

                        tab = D.cached(form._device, "grid_tab", (grid[0],), i,
                                       lambda: torch.empty(sum(grid[2]) * GRID_ROW, dtype=torch.float64, device=grid[0].device))
                        a.grid_idx, a.grid_iv, a.grid_tab = grid[0].data_ptr(), grid[1].data_ptr(), tab.data_ptr()
                        a.grid_n[0], a.grid_n[1], a.grid_n[2] = grid[2]
                        # the rows of the table every block needs (kept in LDS by the blocks), the clusters numbered by them
                        staged = None
                        if os.environ.get("MPCX_GRID_STAGE", "1") != "0":
                            staged = D.cached(form._device, "grid_block_rows", (grid[0], pk), i,
                                              lambda: _block_rows(pk, grid[0], grid[2]))
                        if staged is not None:
                            a.grid_block_rows, a.grid_idx = staged[0].data_ptr(), staged[1].data_ptr()
                            a.grid_block_rows_max = staged[2]
                            keep += [staged]
                        a.tensor_grid = True  # (python attribute)
                        keep += [grid, tab]
            a.algorithm = 3
            a.cube_verts, a.n_cubes = d_verts.data_ptr(), d_verts.shape[0]
            a.leftover = left if left.size else None
            a.kernel_name = name
            keep += [d_verts]
            return a, keep
        if name == "hash":
            a.kernel_name = name
            return a, keep
        md0 = _masked_dofmap(form, V, None, constraint, 0)  # slave flag only: bcs do not touch b here
        if name in ("ownblock", "ufcx_ownblock"):
            # owner-computes lists; the halo rows share the LDS budget, so the own part of a block is smaller
            own = None
            for cap in (VECTOR_OWNER_ROWS, VECTOR_OWNER_ROWS * 3 // 4, VECTOR_OWNER_ROWS // 2):
                own = _vector_owner_plan(form, i, V, md0, _even_rows(V, min(nrows_blk, cap)))
                if own is not None:
                    break
            if own is None:
                continue
            plan, pk, n_own = own
            a.own_lmap, a.own_hoff, a.own_spill = pk[3].data_ptr(), pk[4].data_ptr(), pk[5].data_ptr()
            a.own_src, a.own_rows, a.own_seg, a.n_own_rows = pk[6].data_ptr(), pk[7].data_ptr(), pk[8].data_ptr(), n_own
            if (name == "ownblock" and k.form == 2 and k.fn_id == 1 and k.coeff_degree == 0 and integ.coefficient is None
                    and integ.itype == "cell" and k.celltype == 2 and V.dofmap.bs == 1 and V.degree in (1, 2)
                    and idv["entities_ptr"] is None and form.mesh.geometry.dofmap.shape[1] == 4
                    and (V.degree == 1 or bool(idv["kernel"].qphi))
                    and os.environ.get("MPCX_BOX_GRID", "1") != "0" and os.environ.get("MPCX_CELL_GRID", "1") != "0"):
                # the benchmark's right-hand side on the cells of a box mesh: its univariate factors once per interval and
                # launch instead of a sine and an exponential per quadrature point (csrc/mpcx_kernels.hip vector_cell_grid_kernel)
                cg = _cell_grid(form, i, V, pk)
                if cg is not None:
                    a.grid_idx, a.grid_iv, a.grid_tab = cg["rec"].data_ptr(), cg["iv"].data_ptr(), cg["tab"].data_ptr()
                    a.grid_n[0], a.grid_n[1], a.grid_n[2] = cg["ns"]
                    a.grid_block_rows, a.grid_block_rows_max = cg["rows"].data_ptr(), cg["longest"]
                    a.grid_eta, a.grid_J, a.grid_ng, a.grid_ntypes = cg["eta"].data_ptr(), cg["J"].data_ptr(), cg["ng"], cg["ntypes"]
                    keep += [cg]
        else:  # "rowblock" / "ufcx_rowblock": halo entities evaluated by every block they touch
            plan, pk = _vector_plan(form, i, V, nrows_blk)
        _, slave_ents = _slave_entities(form, i, constraint, constraint)
        a.algorithm = 2
        a.plan = plan
        a.mdofmap = md0.data_ptr()
        a.slave_entities, a.n_slave_entities = slave_ents.data_ptr(), slave_ents.numel()
        a.kernel_name = name
        keep += [pk, md0, slave_ents]
        break
    a.stream = D.stream_ptr()
    return a, keep


@timed("~MPC: Assemble vector (C++)")
def assemble_vector(form: Form, constraint: MultiPointConstraint, b: Optional[Vector] = None,
                    num_threads: Optional[int] = 1, algorithm: Optional[str] = None) -> Vector:
    """Assemble a linear form into ``b`` with the multi point constraint applied
    (python/src/dolfinx_mpc/assemble_vector.py:79-104): ``b`` is created on the
    MPC function space if None, zeroed, then accumulated into.

    ``algorithm``: "atomic" (LDS hash per workgroup + one device atomic per distinct dof),
    "rowblock" (LDS row blocks, no device atomics: faster for cheap integrands, slower when
    the quadrature dominates because entities on block borders are evaluated once per block)
    or "auto" (default: row blocks for rules of <= 4 points); env MPCX_VECTOR_ALG."""
    if form.rank != 1:
        raise RuntimeError("assemble_vector needs a linear form")
    constraint._not_finalized()
    _native.require_gpu()
    D.resolve_builtin_twins(form)  # imported kernels with a stated (and checked) built-in twin, fem.form_ufcx(builtin=...)
    L = _native.lib()
    sid = _native.scalar_id(form.dtype)
    if b is None:
        b = create_vector(constraint.function_space, dtype=form.dtype)
    alg = _ALG[(algorithm or os.environ.get("MPCX_VECTOR_ALG", "auto")).lower()]
    if sid != 0:
        # float32 / complex64 / complex128: the general per-entity kernel (csrc/mpcx_scalar.hip)
        if np.dtype(constraint.dtype) != form.dtype:
            raise ValueError(f"form of scalar type {form.dtype} assembled with a constraint of {np.dtype(constraint.dtype)}")
        if alg == 0:
            alg = 2  # (LDS row blocks; the per-entity kernel with algorithm="atomic")
    for integ in form.integrals:
        if integ.itype not in ("cell", "exterior_facet"):
            raise RuntimeError("Interior facet integrals currently not supported")
    if alg != 1 and sid == 0:
        from . import locality  # a numbering without locality: the spatially reordered twin (locality.py)

        tw = locality.twin_of(form.mesh)
        if tw is not None:
            try:
                return locality.assemble_vector(tw, form, constraint, b, alg)
            except _native.PlanNotRepresentable:
                pass
    D.mesh_device(form.mesh)  # a moved mesh is refreshed on the caller's stream, before any side stream reads it
    from .la import side_stream

    with side_stream("vector", b):  # the library's vector stream (la.side_stream); completion is awaited by b.array
        _assemble_vector_on_stream(form, constraint, b, alg)
    return b


def _assemble_vector_on_stream(form: Form, constraint: MultiPointConstraint, b: Vector, alg: int):
    """the body of ``assemble_vector``: zero, then every integral, enqueued on the CURRENT torch stream"""
    L = _native.lib()
    wt = getattr(b, "_write_through", None)
    (b if wt is None else wt[0]).set(0.0)
    from . import corun

    vfloor = corun.params()["vector_floor"] if corun.note_vector_call(b.device) else 0
    for i, integ in enumerate(form.integrals):
        try:
            a, keep = vector_args(form, i, b, constraint, alg)
        except _native.PlanNotRepresentable:
            if alg != 0:
                raise
            a, keep = vector_args(form, i, b, constraint, 1)  # 'auto': no plan fits (e.g. 64 nodes per cell) -> per-entity kernel
        a.lds_floor = vfloor  # (a matrix assembly is in flight on the other stream: dolfinx_mpc_amd/corun.py)
        _native.check(L.mpcx_assemble_vector(C.byref(a)), "mpcx_assemble_vector")
        if a.second is not None:
            _native.check(L.mpcx_assemble_vector(C.byref(a.second)), "mpcx_assemble_vector")
        if a.leftover is not None:  # cells outside any cluster: per-cell kernel
            from .assemble_matrix import _leftover_form

            al, kl = vector_args(_leftover_form(form, i, a.leftover), 0, b, constraint, alg, allow_cubes=False)
            _native.check(L.mpcx_assemble_vector(C.byref(al)), "mpcx_assemble_vector")
        del keep


def _lift_entities(form: Form, i: int, markers: np.ndarray, d_markers, V1):
    """compact list of entities with a bc-marked column dof (cpp/lifting.h:93-109), built on the device (a gather
    of the markers through the dofmap and a compaction: torch, plumbing -- the host version took 2.4 s for the
    10^8 cells of config 2); cached per marker array (itself cached per (space, bcs): its identity stands for the
    bc set)."""
    def build():
        import torch

        integ = form.integrals[i]
        idv = D.integral_device(form, i)
        dm = D.space_device(V1)["dofmap"].view(-1, V1.element_ndofs)  # (num_cells, nd) blocked
        if idv["entities"] is None:
            dofs = dm[: integ.num_entities]
        else:
            dofs = dm[idv["entities"].view(integ.num_entities, integ.estride)[:, 0].long()]
        bs = V1.dofmap.bs
        dofs = dofs.long()
        hit = torch.zeros(dofs.shape[0], dtype=torch.bool, device=dofs.device)
        for k in range(bs):
            hit |= (d_markers[dofs * bs + k] != 0).any(dim=1)
        idx = torch.nonzero(hit).reshape(-1).to(torch.int32).contiguous()
        return (None, idx)

    return D.cached(form._device, "lift_ents", (markers,), i, build)


@timed("~MPC: Apply lifting (C++)")
def apply_lifting(
    b: Vector,
    form: Sequence[Optional[Form]],
    bcs: Sequence[Sequence[DirichletBC]],
    constraint: MultiPointConstraint,
    x0: Optional[Sequence[Vector]] = None,
    scale: float = 1.0,
    num_threads: Optional[int] = 1,
):
    """b <- b - scale * K^T A_j (g_j - x0_j)
    (python/src/dolfinx_mpc/assemble_vector.py:25-76, cpp/lifting.h:441-483)."""
    import torch

    if isinstance(scale, np.generic):
        scale = scale.item()
    if isinstance(b, (list, tuple)):
        # nest: b = [b_i], form = [[a_ij]], constraint = [mpc_i]; bcs are grouped by the column space they
        # live in (dolfinx bcs_by_block), assemble_vector.py:51-62
        rows, cons = [list(r) for r in form], list(constraint)
        if len(rows) != len(b) or len(cons) != len(b):
            raise RuntimeError("Mismatch in size between b, a and the constraints in assembler.")
        blocks = list(bcs)
        if len(blocks) == 0 or isinstance(blocks[0], DirichletBC):
            blocks = []
            for j in range(len(rows[0])):
                Vj = next((r[j].function_spaces[1] for r in rows if r[j] is not None), None)
                blocks.append([bc for bc in bcs if Vj is not None and Vj.contains(bc.function_space)])
        for b_sub, a_sub, mpc_i in zip(b, rows, cons):
            apply_lifting(b_sub, a_sub, blocks, mpc_i, x0=x0, scale=scale, num_threads=num_threads)
        return
    x0 = [] if x0 is None else list(x0)
    form = list(form)
    if len(x0) > 0 and len(x0) != len(form):
        raise RuntimeError("Mismatch in size between x0 and bilinear form in assembler.")
    if len(form) != len(bcs):
        raise RuntimeError("Mismatch in size between a and bcs in assembler.")
    if all(f is None for f in form):
        return
    constraint._not_finalized()
    dev = _native.require_gpu()
    first = next(f for f in form if f is not None)
    for f in form:
        if f is not None:
            D.resolve_builtin_twins(f)
    from . import locality  # a numbering without locality: the spatially reordered twin (locality.py)

    tw = locality.twin_of(first.mesh) if _native.scalar_id(first.dtype) == 0 else None
    if tw is not None:
        try:
            return locality.apply_lifting(tw, b, form, bcs, constraint, x0, scale)
        except _native.PlanNotRepresentable:
            pass
    L = _native.lib()
    m, _keep = constraint._device()
    for j, aj in enumerate(form):
        if aj is None or len(bcs[j]) == 0:
            continue
        V0, V1 = aj.function_spaces
        # bc markers / values over the column space, cpp/lifting.h:166-180 (values read live)
        markers, d_markers, d_values = D.bc_values(V1, bcs[j], aj._device, aj.dtype)
        md = D.mesh_device(aj.mesh)
        s0, s1 = D.space_device(V0), D.space_device(V1)
        x0j = None
        if len(x0) > 0:
            x0j = x0[j].array if hasattr(x0[j], "array") else x0[j]
        for i, integ in enumerate(aj.integrals):
            if integ.itype not in ("cell", "exterior_facet"):
                raise RuntimeError("Interior facet integrals currently not supported")
            idv = D.integral_device(aj, i)
            _, lift = _lift_entities(aj, i, markers, d_markers, V1)
            a = _native.LiftingArgs()
            a.b, a.num_dofs = b.array.data_ptr(), b.size
            a.kernel = idv["kernel"]
            a.x, a.x_dofmap, a.nv = md["x"].data_ptr(), md["x_dofmap"].data_ptr(), aj.mesh.geometry.dofmap.shape[1]
            a.estride, a.n_entities = integ.estride, integ.num_entities
            a.entities = a.entities0 = a.entities1 = idv["entities_ptr"]
            a.coeffs = D.ptr(idv["coeffs"])
            a.cstride = integ.cstride
            a.constants = D.ptr(idv["constants"])
            a.dofmap0, a.nd0, a.bs0 = s0["dofmap"].data_ptr(), V0.element_ndofs, V0.dofmap.bs
            a.dofmap1, a.nd1, a.bs1 = s1["dofmap"].data_ptr(), V1.element_ndofs, V1.dofmap.bs
            a.bc_markers1, a.bc_values1 = d_markers.data_ptr(), d_values.data_ptr()
            a.x0 = None if x0j is None else x0j.data_ptr()
            a.scale = float(scale)
            a.lift_entities, a.n_lift_entities = lift.data_ptr(), lift.numel()
            a.mpc0 = m
            a.cell_info0, a.cell_info1 = D.cell_info_ptr(V0, integ.kernel), D.cell_info_ptr(V1, integ.kernel)
            a.stream = D.stream_ptr()
            _native.check(L.mpcx_apply_lifting(C.byref(a)), "mpcx_apply_lifting")


def set_bc(b: Vector, bcs: Sequence[DirichletBC], x0: Optional[Vector] = None, scale: float = 1.0):
    """dolfinx ``set_bc`` (bench_periodic.py:109): b[bc dofs] = scale * (g - x0)."""
    import torch

    for bc in bcs:
        dofs = bc.dof_indices()[0]
        idx = torch.from_numpy(dofs.astype(np.int64)).to(b.array.device)
        g = torch.from_numpy(np.ascontiguousarray(bc.values_at_dofs())).to(b.array.device)
        if x0 is not None:
            g = g - x0.array[idx]
        b.array[idx] = scale * g


def create_vector_nest(L: Sequence[Form], constraints: Sequence[MultiPointConstraint]):
    """python/src/dolfinx_mpc/assemble_vector.py:107-127"""
    assert len(constraints) == len(L)
    return [create_vector(c.function_space) for c in constraints]


def assemble_vector_nest(b, L: Sequence[Form], constraints: Sequence[MultiPointConstraint],
                         num_threads: Optional[int] = 1):
    """python/src/dolfinx_mpc/assemble_vector.py:130-147"""
    assert len(constraints) == len(L)
    for i, L_row in enumerate(L):
        if L_row is None:  # the reference passes ufl.ZeroBaseForm here: the sub-vector is zeroed
            b[i].set(0.0)
        else:
            assemble_vector(L_row, constraints[i], b=b[i], num_threads=num_threads)
