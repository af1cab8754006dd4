"""Multi-GPU partition of the assembly path: one process per GPU, cells split
into z-slabs, one exchange step after the local kernels.

The reference does no communication inside assembly; rows owned by another
rank are shipped by PETSc in ``A.assemble()`` (python/src/dolfinx_mpc/assemble_matrix.py:64)
and by ``b.ghostUpdate(ADD_VALUES, REVERSE)`` (python/benchmarks/bench_periodic.py:108).
Here the same step is a neighbour send/recv of the packed partial sums of the
interface-plane rows over ``torch.distributed`` (backend "nccl" = RCCL over
xGMI; "gloo" in the CPU tests), followed by a scatter-add on the owner.

Partition: the global mesh has ``(N, N, N*world)`` cubes;
rank r integrates the cubes with ``r*N <= kz < (r+1)*N``.  Node planes
``r*N .. (r+1)*N - 1`` are owned by rank r (the last rank also owns the top
plane); its local mesh additionally holds

* the top plane ``(r+1)*N`` as ghost nodes (rows receive partial sums that are
  sent to rank r+1), and
* for r > 0 one layer of ghost cubes below (``kz = r*N - 1``), never
  integrated, only there so that the owned bottom-plane rows have the columns
  rank r-1's contributions need (DOLFINx's shared-facet ghost layer).

Slabs are cut along z so a periodic slave (1, y, z) and its master (0, y, z)
always live on the same rank: the constraint adds no extra exchange.
"""

from __future__ import annotations

import numpy as np

from .mesh import _KUHN, Mesh, _tile_permutation


def slab_layers(n_axis: int, rank: int, world: int):
    """cube layers [l0, l1) of rank ``rank`` when ``n_axis`` layers are dealt out as evenly as possible"""
    return (rank * n_axis) // world, ((rank + 1) * n_axis) // world


def create_box_slab(p0, p1, n, rank: int, world: int, axis: int = 2, reorder=None, layers=None,
                    global_offset: int = 0, ghost_layers: int = 1) -> Mesh:
    """Rank ``rank``'s part of ONE global box of ``n = (nx, ny, nz)`` cubes (6 tets each) on [p0, p1],
    cut into ``world`` slabs of cube layers along ``axis`` (strong scaling: the global problem is fixed).
    The local mesh holds the owned cube layers ``[l0, l1)`` (``layers`` or an even split) plus, for
    rank > 0, ``ghost_layers`` ghost layers below (never integrated; they supply the columns the lower
    neighbour's contributions need: one layer for plain assembly, more when constraints carry
    contributions further, see create_stacked_cubes_slab).  Node planes l0 .. l1-1 are owned (the last rank also owns plane n_axis); the
    plane l1 above is held as ghost nodes whose partial sums go to rank + 1, the plane l0 - 1 below as
    column-only ghosts.  Nodes and cells are numbered owned-first (tile order inside each group).

    Sets ``num_owned_nodes``, ``num_owned_cells``, ``node_global`` (id in the global box, x fastest,
    + ``global_offset``), ``node_send_up`` (bool: ghost node of the upper interface plane)."""
    nx, ny, nz = (int(v) for v in n)
    nax = (nx, ny, nz)[axis]
    l0, l1 = slab_layers(nax, rank, world) if layers is None else layers
    if l1 <= l0:
        raise RuntimeError(f"create_box_slab: rank {rank} of {world} gets no cube layer of {nax}")
    c0 = max(0, l0 - ghost_layers) if rank > 0 else l0  # first cube layer held
    ncl = l1 - c0  # cube layers held
    # local grid: `ncl` cube layers along `axis`, full extent in the other two directions
    nloc = [nx, ny, nz]
    nloc[axis] = ncl
    lx, ly, lz = nloc
    nx1, ny1, nz1 = lx + 1, ly + 1, lz + 1
    ax = [np.linspace(p0[d], p1[d], (nx, ny, nz)[d] + 1) for d in range(3)]
    ax[axis] = ax[axis][c0 : l1 + 1]
    Z, Y, X = np.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
    x = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    k, j, i = np.meshgrid(np.arange(lz), np.arange(ly), np.arange(lx), indexing="ij")
    base = ((k * ny1 + j) * nx1 + i).ravel().astype(np.int64)
    corner = np.empty((base.size, 8), dtype=np.int64)
    for b in range(8):
        corner[:, b] = base + (b & 1) + ((b >> 1) & 1) * nx1 + ((b >> 2) & 1) * nx1 * ny1
    cells = corner[:, _KUHN]  # (ncubes, 6, 4) lexicographic local node ids
    cube_layer = c0 + (i, j, k)[axis].ravel()
    gk, gj, gi = np.meshgrid(np.arange(nz1), np.arange(ny1), np.arange(nx1), indexing="ij")
    loc = [gi.ravel(), gj.ravel(), gk.ravel()]
    node_plane = c0 + loc[axis]  # global plane index along the slab axis
    last = rank == world - 1
    owned_node = (node_plane >= l0) & ((node_plane < l1) | (last & (node_plane == l1)))
    owned_cube = cube_layer >= l0
    if reorder is not None:
        # the tiling starts at the first OWNED plane / cube layer: with the ghost layer below counted in, the owned nodes of a
        # rank > 0 fell into tiles of 7 + 1 planes (448-row tiles, 273 clusters per row block: two 256-thread passes with 17
        # lanes in the second)
        shift = [0, 0, 0]
        shift[axis] = l0 - c0
        shift = tuple(shift)
        tperm = _tile_permutation((nx1, ny1, nz1), reorder, shift)  # old -> tile position
        cperm = _tile_permutation((lx, ly, lz), reorder, shift)
    else:
        tperm = np.arange(x.shape[0])
        cperm = np.arange(base.size)
    order = np.lexsort((tperm, ~owned_node))  # new -> old: owned first, then by tile position
    perm = np.empty_like(order)
    perm[order] = np.arange(order.size)  # old -> new
    corder = np.lexsort((cperm, ~owned_cube))
    cells = perm[cells[corder]].reshape(-1, 4)
    glob = [loc[0].astype(np.int64), loc[1].astype(np.int64), loc[2].astype(np.int64)]
    glob[axis] = glob[axis] + c0
    node_global = ((glob[2] * (ny + 1) + glob[1]) * (nx + 1) + glob[0])[order] + int(global_offset)
    mesh = Mesh(x[order], cells.astype(np.int32), "tetrahedron")
    if reorder is not None:
        from .mesh import _tile_ids, _tile_starts

        # keep owned / ghost groups apart in the hint (ghost group gets its own tile ids)
        tid = _tile_ids((nx1, ny1, nz1), reorder, shift).astype(np.int64)
        tid = np.where(owned_node, tid, tid + tid.max() + 1)
        mesh.node_tile_offsets = _tile_starts(tid[order])
    mesh.num_owned_nodes = int(owned_node.sum())
    mesh.num_owned_cells = int(owned_cube.sum()) * 6
    mesh.node_global = node_global
    mesh.node_send_up = ((node_plane == l1) & ~owned_node)[order]
    mesh.node_plane = node_plane[order]
    mesh.partition = dict(axis=axis, layers=(l0, l1), n=(nx, ny, nz), rank=rank, world=world)
    return mesh


def create_slab_mesh(N: int, rank: int, world: int, reorder=None) -> Mesh:
    """Weak-scaling partition: rank ``rank``'s N^3 box of the (N, N, N*world) mesh on
    [0,1]^2 x [0,world] (every rank gets the same amount of work whatever ``world`` is)."""
    mesh = create_box_slab((0.0, 0.0, 0.0), (1.0, 1.0, float(world)), (N, N, N * world), rank, world, 2, reorder,
                           layers=(rank * N, (rank + 1) * N))
    mesh.slab = (N, rank, world)
    return mesh


def merge_slab_meshes(meshes) -> Mesh:
    """Disjoint union of slab meshes of several bodies held by ONE rank (the two cubes of the contact
    benchmark): nodes owned-first over all bodies, then the ghosts; cells likewise; the per-node
    partition data (global id, send-up flag) follows."""
    nn = [m.num_nodes for m in meshes]
    no = [m.num_owned_nodes for m in meshes]
    noff_owned = np.concatenate([[0], np.cumsum(no)])
    ng = [a - b for a, b in zip(nn, no)]
    goff = noff_owned[-1] + np.concatenate([[0], np.cumsum(ng)])
    new_of = []  # per body: old local node -> new local node
    for b, m in enumerate(meshes):
        t = np.empty(nn[b], dtype=np.int64)
        t[: no[b]] = noff_owned[b] + np.arange(no[b])
        t[no[b] :] = goff[b] + np.arange(ng[b])
        new_of.append(t)
    ntot = int(sum(nn))
    x = np.empty((ntot, 3))
    node_global = np.empty(ntot, dtype=np.int64)
    send_up = np.zeros(ntot, dtype=bool)
    body = np.empty(ntot, dtype=np.int32)
    for b, m in enumerate(meshes):
        x[new_of[b]] = m.geometry.x
        node_global[new_of[b]] = m.node_global
        send_up[new_of[b]] = m.node_send_up
        body[new_of[b]] = b
    owned_cells = [new_of[b][m.geometry.dofmap[: m.num_owned_cells]] for b, m in enumerate(meshes)]
    ghost_cells = [new_of[b][m.geometry.dofmap[m.num_owned_cells :]] for b, m in enumerate(meshes)]
    cells = np.concatenate(owned_cells + ghost_cells, axis=0)
    cell_body = np.concatenate([np.full(c.shape[0], b, dtype=np.int32) for b, c in enumerate(owned_cells)]
                               + [np.full(c.shape[0], b, dtype=np.int32) for b, c in enumerate(ghost_cells)])
    out = Mesh(x, cells.astype(np.int32), meshes[0].cell_name)
    out.num_owned_nodes = int(noff_owned[-1])
    out.num_owned_cells = int(sum(m.num_owned_cells for m in meshes))
    out.node_global = node_global
    out.node_send_up = send_up
    out.node_body = body
    out.cell_body = cell_body
    if all(m.node_tile_offsets is not None for m in meshes):
        hints = []
        for b, m in enumerate(meshes):
            h = m.node_tile_offsets.astype(np.int64)
            hints.append(new_of[b][h])
        out.node_tile_offsets = np.unique(np.concatenate(hints)).astype(np.int32)
    out.partition = dict(meshes[0].partition, bodies=len(meshes))
    return out


def create_stacked_cubes_slab(n_top: int, rank: int, world: int, theta: float = 0.0, reorder=None, axis: int = 1):
    """Rank ``rank``'s part of the two-body contact mesh (``mesh.create_stacked_cubes(n_top)``, bottom body
    2 n_top cubes per side): BOTH bodies are cut along ``axis`` (default y, a direction inside the contact
    plane), at the same physical positions, so that the two interface layers of a slab live on one GPU
    and every slave finds its masters locally.
    ``n_top`` must be divisible by ``world``.  Returns (mesh, facet_tags)."""
    from .mesh import (CONTACT_BOTTOM, CONTACT_BOTTOM_INTERFACE, CONTACT_TOP, CONTACT_TOP_INTERFACE, MeshTags,
                       _facets_on_plane, rotation_matrix)

    if axis == 2:
        raise RuntimeError("create_stacked_cubes_slab: cut along x or y (z is the contact normal)")
    if n_top % world != 0:
        raise RuntimeError("create_stacked_cubes_slab: n_top must be divisible by the number of ranks")
    lt = (rank * n_top // world, (rank + 1) * n_top // world)
    top = create_box_slab((0.0, 0.0, 1.0), (1.0, 1.0, 2.0), (n_top,) * 3, rank, world, axis, reorder, layers=lt)
    # two fine ghost layers below: a master row on the slab's upper plane collects the contributions of the
    # cells round its slaves (fine nodes one fine layer below the plane), which reach two fine layers down
    bot = create_box_slab((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), (2 * n_top,) * 3, rank, world, axis, reorder,
                          layers=(2 * lt[0], 2 * lt[1]), global_offset=(n_top + 1) ** 3, ghost_layers=2)
    mesh = merge_slab_meshes([top, bot])
    z = mesh.geometry.x[:, 2]
    cells = mesh.geometry.dofmap.astype(np.int64)
    is_top_node = mesh.node_body == 0

    def tagged(on_plane, body):
        ids = np.flatnonzero(mesh.cell_body == body)
        cand = ids[on_plane[cells[ids]].any(axis=1)]
        return _facets_on_plane(cells, cand, on_plane)

    f_top = tagged(np.isclose(z, 2.0), 0)
    f_tif = tagged(np.isclose(z, 1.0) & is_top_node, 0)
    f_bif = tagged(np.isclose(z, 1.0) & ~is_top_node, 1)
    f_bot = tagged(np.isclose(z, 0.0), 1)
    ents = np.concatenate([f_top, f_bif, f_tif, f_bot], axis=0)
    vals = np.concatenate([np.full(f.shape[0], v, dtype=np.int32) for f, v in
                           ((f_top, CONTACT_TOP), (f_bif, CONTACT_BOTTOM_INTERFACE), (f_tif, CONTACT_TOP_INTERFACE),
                            (f_bot, CONTACT_BOTTOM))])
    if theta != 0.0:
        R = rotation_matrix([1 / np.sqrt(2), 1 / np.sqrt(2), 0], -theta)
        mesh.geometry.x = mesh.geometry.x @ R.T
    return mesh, MeshTags(mesh, 2, ents, vals)


class SlabExchange:
    """Ghost-row reduction for a (blocked) P1 space on a slab mesh: rank r sends the
    partial sums of its top-plane (ghost) rows to rank r+1, which adds them to
    its owned bottom-plane rows.  The value buffers are packed / scattered with
    index tensors built once; positions are matched through global (row, col)
    keys exchanged at set-up."""

    def __init__(self, mesh: Mesh, rowptr, cols, rank: int, world: int, device=None,
                 bs: int = 1, space=None):
        """``space``: the (row == column) function space; default = P1 on ``mesh`` with block size ``bs``.
        P2 spaces carry their own global ids / planes (FunctionSpace._p2_on_slab).
        ``rowptr`` / ``cols`` None: vectors only (``reduce_vector`` / ``forward_vector``)."""
        import torch
        import torch.distributed as dist

        vector_only = rowptr is None
        if vector_only:
            rowptr, cols = np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int32)
        self.vector_only = vector_only

        self.rank, self.world = rank, world
        self.device = device
        if space is not None:
            bs = space.dofmap.bs
        if space is not None and space.degree == 2:
            # P2 on z-slabs: edge dofs carry their own global ids / planes
            blk_global = space.dof_global
            blk_send = space.dof_send_up
        else:
            # dofs numbered like the mesh nodes (P1, scalar or blocked)
            blk_global = mesh.node_global
            blk_send = mesh.node_send_up
        # unrolled dofs: dof = block * bs + component
        g = (blk_global[:, None] * bs + np.arange(bs)[None, :]).reshape(-1)
        send = np.repeat(blk_send, bs)
        self.send_to = rank + 1 if rank + 1 < world else None
        self.recv_from = rank - 1 if rank > 0 else None
        # ---- what I send: every entry of my upper-interface (ghost) rows -------
        top = np.flatnonzero(send) if self.send_to is not None else np.zeros(0, dtype=np.int64)
        top = top[np.argsort(g[top])]
        cnt = np.zeros(top.size, dtype=np.int64) if vector_only else rowptr[top + 1] - rowptr[top]
        pos = (np.repeat(rowptr[top].astype(np.int64) - np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt)
               + np.arange(int(cnt.sum()))) if (top.size and not vector_only) else np.zeros(0, dtype=np.int64)
        # entries ordered by (global row, global col)
        k_row = np.repeat(g[top], cnt).astype(np.int64) if pos.size else np.zeros(0, dtype=np.int64)
        k_col = g[cols[pos]].astype(np.int64) if pos.size else np.zeros(0, dtype=np.int64)
        o = np.lexsort((k_col, k_row))
        self.send_pos = pos[o]
        send_keys = np.concatenate([k_row[o], k_col[o]])  # [rows..., cols...]
        self.send_rows = top
        # ---- key exchange (set-up only) -------------------------------------
        # RCCL moves device buffers; gloo (CPU tests) moves host buffers
        comm_dev = device if (device is not None and dist.get_backend() != "gloo") else None

        def _t(a, dtype):
            t = torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
            return t.to(comm_dev) if comm_dev is not None else t

        n_send = _t(np.array([send_keys.size, top.size]), torch.int64)
        n_recv = torch.zeros_like(n_send)
        ops = []
        if self.send_to is not None:
            ops.append(dist.P2POp(dist.isend, n_send, self.send_to))
        if self.recv_from is not None:
            ops.append(dist.P2POp(dist.irecv, n_recv, self.recv_from))
        for w in dist.batch_isend_irecv(ops) if ops else []:
            w.wait()
        nk, nr = (int(n_recv[0]), int(n_recv[1])) if self.recv_from is not None else (0, 0)
        k_send = _t(send_keys, torch.int64)
        r_send = _t(g[top], torch.int64)
        k_recv = torch.zeros(nk, dtype=torch.int64, device=k_send.device)
        r_recv = torch.zeros(nr, dtype=torch.int64, device=k_send.device)
        ops = []
        if self.send_to is not None:
            ops += [dist.P2POp(dist.isend, k_send, self.send_to), dist.P2POp(dist.isend, r_send, self.send_to)]
        if self.recv_from is not None:
            ops += [dist.P2POp(dist.irecv, k_recv, self.recv_from), dist.P2POp(dist.irecv, r_recv, self.recv_from)]
        for w in dist.batch_isend_irecv(ops) if ops else []:
            w.wait()
        # ---- where received values go in my CSR / vector ----------------------
        if self.recv_from is not None:
            kr = k_recv.cpu().numpy()
            grow, gcol = kr[: kr.size // 2], kr[kr.size // 2 :]
            # global id -> local dof by sorted search (no dense inverse table)
            gorder = np.argsort(g, kind="stable")
            gsorted = g[gorder]

            def to_local(q):
                p = np.searchsorted(gsorted, q)
                p = np.minimum(p, gsorted.size - 1)
                if (gsorted[p] != q).any():
                    raise RuntimeError("SlabExchange: received a row/column this rank does not hold")
                return gorder[p]

            lrow, lcol = to_local(grow), to_local(gcol)
            # vectorised binary search per entry inside its row
            lo = rowptr[lrow].astype(np.int64) if lrow.size else np.zeros(0, dtype=np.int64)
            hi = rowptr[lrow + 1].astype(np.int64) if lrow.size else np.zeros(0, dtype=np.int64)
            while lrow.size:
                active = lo < hi
                if not active.any():
                    break
                mid = (lo + hi) // 2
                less = np.zeros_like(active)
                less[active] = cols[mid[active]] < lcol[active]
                lo = np.where(active & less, mid + 1, lo)
                hi = np.where(active & ~less, mid, hi)
            if lrow.size and (cols[np.minimum(lo, cols.size - 1)] != lcol).any():
                raise RuntimeError("SlabExchange: received an entry outside the local sparsity pattern")
            self.recv_pos = lo
            self.recv_rows = to_local(r_recv.cpu().numpy())
        else:
            self.recv_pos = np.zeros(0, dtype=np.int64)
            self.recv_rows = np.zeros(0, dtype=np.int64)
        self._tensors = {}

    def _idx(self, name, dev):
        import torch

        key = (name, str(dev))
        if key not in self._tensors:
            self._tensors[key] = torch.from_numpy(np.ascontiguousarray(getattr(self, name))).to(torch.int64).to(dev)
        return self._tensors[key]

    def _begin(self, values, send_idx, recv_idx):
        """Pack the send buffer and post the neighbour send / receive; returns a handle for ``finish``.
        Nothing waits here: work enqueued afterwards (the next assembly kernel) overlaps the transfer."""
        import torch
        import torch.distributed as dist

        dev = values.device
        stage_cpu = dist.get_backend() == "gloo" and dev.type == "cuda"
        on_gpu = dev.type == "cuda"
        sbuf = None
        if self.send_to is not None:
            idx = self._idx(send_idx, dev)
            if on_gpu:  # library kernels for the data path (64-bit indexing)
                from . import _device as D
                from . import _native

                sbuf = torch.empty(idx.numel(), dtype=values.dtype, device=dev)
                _native.check(_native.lib().mpcx_gather_f64(values.data_ptr(), idx.data_ptr(), idx.numel(),
                                                            sbuf.data_ptr(), D.stream_ptr()), "mpcx_gather_f64")
            else:
                sbuf = values.index_select(0, idx)
        rbuf = torch.empty(getattr(self, recv_idx).size, dtype=values.dtype, device=dev)
        if stage_cpu:
            sbuf = None if sbuf is None else sbuf.cpu()
            rbuf = rbuf.cpu()
        ops = []
        if self.send_to is not None:
            ops.append(dist.P2POp(dist.isend, sbuf, self.send_to))
        if self.recv_from is not None:
            ops.append(dist.P2POp(dist.irecv, rbuf, self.recv_from))
        works = dist.batch_isend_irecv(ops) if ops else []
        return (values, recv_idx, sbuf, rbuf, works)

    def finish(self, handle):
        """Wait for the transfer of ``handle`` and add the received partial sums into the owned rows."""
        if handle is None:
            return
        values, recv_idx, _sbuf, rbuf, works = handle
        for w in works:
            w.wait()
        if self.recv_from is not None:
            dev = values.device
            idx = self._idx(recv_idx, dev)
            if dev.type == "cuda":
                from . import _device as D
                from . import _native

                rbuf = rbuf.to(dev)
                _native.check(_native.lib().mpcx_scatter_add_f64(values.data_ptr(), idx.data_ptr(), idx.numel(),
                                                                 rbuf.data_ptr(), D.stream_ptr()), "mpcx_scatter_add_f64")
            else:
                values.index_add_(0, idx, rbuf.to(dev))

    def _exchange(self, values, send_idx, recv_idx):
        """values: 1-D fp64 tensor (CSR values or vector), modified in place."""
        self.finish(self._begin(values, send_idx, recv_idx))

    def reduce_matrix_begin(self, A):
        if self.vector_only:
            raise RuntimeError("SlabExchange: built without a sparsity pattern (vectors only)")
        return self._begin(A.vals if hasattr(A, "vals") else A, "send_pos", "recv_pos")

    def forward_vector(self, b):
        """``ghostUpdate(INSERT, FORWARD)`` / ``scatter_forward``: the owners' values of the interface rows are copied
        to the ghost rows of the rank below (the reverse direction of ``reduce_vector``, same index lists)."""
        import torch
        import torch.distributed as dist

        arr = b.array if hasattr(b, "array") else b
        dev = arr.device
        stage_cpu = dist.get_backend() == "gloo" and dev.type == "cuda"
        ops, sbuf, rbuf = [], None, None
        if self.recv_from is not None:  # I own rows my lower neighbour holds as ghosts
            sbuf = arr.index_select(0, self._idx("recv_rows", dev))
            sbuf = sbuf.cpu() if stage_cpu else sbuf
            ops.append(dist.P2POp(dist.isend, sbuf, self.recv_from))
        if self.send_to is not None:
            rbuf = torch.empty(self.send_rows.size, dtype=arr.dtype, device="cpu" if stage_cpu else dev)
            ops.append(dist.P2POp(dist.irecv, rbuf, self.send_to))
        for w in dist.batch_isend_irecv(ops) if ops else []:
            w.wait()
        if rbuf is not None:
            arr.index_copy_(0, self._idx("send_rows", dev), rbuf.to(dev))

    def reduce_vector_begin(self, b):
        return self._begin(b.array if hasattr(b, "array") else b, "send_rows", "recv_rows")

    def reduce_matrix(self, A):
        """A.assemble() analogue: add the neighbour's partial sums into the owned rows."""
        vals = A.vals if hasattr(A, "vals") else A
        self._exchange(vals, "send_pos", "recv_pos")

    def reduce_vector(self, b):
        """b.ghostUpdate(ADD_VALUES, REVERSE) analogue."""
        arr = b.array if hasattr(b, "array") else b
        self._exchange(arr, "send_rows", "recv_rows")


def exchange_for(space, A=None):
    """The interface exchange of ``space`` on a partitioned mesh (``mesh.partition`` with world > 1) when a process
    group is initialised, else None.  Without ``A``: vectors only, cached on the space; with the (square) matrix
    ``A`` over ``space``: its value positions too, cached on the matrix.  This is what ``create_vector`` /
    ``create_matrix`` attach, so that ``b.ghostUpdate`` / ``A.assemble()`` do the reduction
    (bench_periodic.py:108, python/src/dolfinx_mpc/assemble_matrix.py:64)."""
    import torch
    import torch.distributed as dist

    mesh = space.mesh
    part = getattr(mesh, "partition", None)
    if part is None or part.get("world", 1) <= 1 or not (dist.is_available() and dist.is_initialized()):
        return None
    dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    kw = dict(device=dev, bs=space.dofmap.bs, space=space if space.degree == 2 else None)
    if A is None:
        key = ("vec_exchange", str(dev))
        if key not in space._device:
            space._device[key] = SlabExchange(mesh, None, None, part["rank"], part["world"], **kw)
        return space._device[key]
    key = ("mat_exchange", id(space))
    if key not in A._plans:
        A._plans[key] = (space, SlabExchange(mesh, A.rowptr, A.cols, part["rank"], part["world"], **kw))
    return A._plans[key][1]
