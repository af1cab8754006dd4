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


def create_slab_mesh(N: int, rank: int, world: int, reorder=None) -> Mesh:
    """Local mesh of rank ``rank``: owned cubes + (r>0) one ghost cube layer below.
    Nodes are numbered owned-first (tile order inside each group), cells
    owned-first.  Sets ``mesh.num_owned_nodes``, ``mesh.node_global`` (global
    node id, x fastest over the (N+1, N+1, N*world+1) grid), ``mesh.num_owned_cells``."""
    kz0 = rank * N - (1 if rank > 0 else 0)  # first cube layer held
    kz1 = (rank + 1) * N  # one past the last cube layer
    nzc = kz1 - kz0
    nx1, ny1, nz1 = N + 1, N + 1, nzc + 1
    pz0 = kz0  # first node plane held
    xs = np.linspace(0.0, 1.0, N + 1)
    zs = (pz0 + np.arange(nz1)) / float(N)
    Z, Y, X = np.meshgrid(zs, xs, xs, indexing="ij")
    x = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    k, j, i = np.meshgrid(np.arange(nzc), np.arange(N), np.arange(N), indexing="ij")
    base = ((k * ny1 + j) * nx1 + i).ravel().astype(np.int64)
    corner = np.empty((base.size, 8), dtype=np.int64)
    for b in range(8):
        corner[:, b] = base + (b & 1) + ((b >> 1) & 1) * nx1 + ((b >> 2) & 1) * nx1 * ny1
    cells = corner[:, _KUHN]  # (ncubes, 6, 4) lexicographic local node ids
    cube_kz = (kz0 + k).ravel()
    node_plane = pz0 + np.repeat(np.arange(nz1), nx1 * ny1)
    last = rank == world - 1
    owned_node = (node_plane >= rank * N) & ((node_plane < (rank + 1) * N) | (last & (node_plane == (rank + 1) * N)))
    owned_cube = cube_kz >= rank * N
    # numbering: tile order (or lexicographic), owned first
    if reorder is not None:
        tperm = _tile_permutation((nx1, ny1, nz1), reorder)  # old -> tile position
        cperm = _tile_permutation((N, N, nzc), reorder)
    else:
        tperm = np.arange(x.shape[0])
        cperm = np.arange(base.size)
    order = np.lexsort((tperm, ~owned_node))  # new -> old: owned first, then by tile position
    perm = np.empty_like(order)
    perm[order] = np.arange(order.size)  # old -> new
    corder = np.lexsort((cperm, ~owned_cube))
    cells = perm[cells[corder]].reshape(-1, 4)
    gk, gj, gi = node_plane, np.tile(np.repeat(np.arange(ny1), nx1), nz1), np.tile(np.arange(nx1), ny1 * nz1)
    node_global = ((gk.astype(np.int64) * ny1 + gj) * nx1 + gi)[order]
    mesh = Mesh(x[order], cells.astype(np.int32), "tetrahedron")
    if reorder is not None:
        from .mesh import _tile_ids, _tile_starts

        # keep owned / ghost groups apart in the hint (ghost group gets its own tile ids)
        tid = _tile_ids((nx1, ny1, nz1), reorder).astype(np.int64)
        tid = np.where(owned_node, tid, tid + tid.max() + 1)
        mesh.node_tile_offsets = _tile_starts(tid[order])
    mesh.num_owned_nodes = int(owned_node.sum())
    mesh.num_owned_cells = int(owned_cube.sum()) * 6
    mesh.node_global = node_global
    mesh.slab = (N, rank, world)
    return mesh


class SlabExchange:
    """Ghost-row reduction for a (blocked) P1 space on a slab mesh: rank r sends the
    partial sums of its top-plane (ghost) rows to rank r+1, which adds them to
    its owned bottom-plane rows.  The value buffers are packed / scattered with
    index tensors built once; positions are matched through global (row, col)
    keys exchanged at set-up."""

    def __init__(self, mesh: Mesh, rowptr: np.ndarray, cols: np.ndarray, rank: int, world: int, device=None,
                 bs: int = 1, space=None):
        """``space``: the (row == column) function space; default = P1 on ``mesh`` with block size ``bs``.
        P2 spaces carry their own global ids / planes (FunctionSpace._p2_on_slab)."""
        import torch
        import torch.distributed as dist

        self.rank, self.world = rank, world
        self.device = device
        N = mesh.slab[0]
        if space is not None:
            bs = space.dofmap.bs
            blk_global, blk_plane = space.dof_global, space.dof_plane
        else:
            blk_global = mesh.node_global
            blk_plane = mesh.node_global // ((N + 1) * (N + 1))
        # unrolled dofs: dof = block * bs + component
        g = (blk_global[:, None] * bs + np.arange(bs)[None, :]).reshape(-1)
        plane = np.repeat(blk_plane, bs)
        self.send_to = rank + 1 if rank + 1 < world else None
        self.recv_from = rank - 1 if rank > 0 else None
        # ---- what I send: every entry of my top-plane rows -------------------
        top = np.flatnonzero(plane == (rank + 1) * N) if self.send_to is not None else np.zeros(0, dtype=np.int64)
        top = top[np.argsort(g[top])]
        cnt = rowptr[top + 1] - rowptr[top]
        pos = (np.repeat(rowptr[top].astype(np.int64) - np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt)
               + np.arange(int(cnt.sum()))) if top.size else np.zeros(0, dtype=np.int64)
        # entries ordered by (global row, global col)
        k_row = np.repeat(g[top], cnt).astype(np.int64) if top.size else np.zeros(0, dtype=np.int64)
        k_col = g[cols[pos]].astype(np.int64) if top.size else np.zeros(0, dtype=np.int64)
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
            lo = rowptr[lrow].astype(np.int64)
            hi = rowptr[lrow + 1].astype(np.int64)
            while True:
                active = lo < hi
                if not active.any():
                    break
                mid = (lo + hi) // 2
                less = np.zeros_like(active)
                less[active] = cols[mid[active]] < lcol[active]
                lo = np.where(active & less, mid + 1, lo)
                hi = np.where(active & ~less, mid, hi)
            if (cols[np.minimum(lo, cols.size - 1)] != lcol).any():
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
        return self._begin(A.vals if hasattr(A, "vals") else A, "send_pos", "recv_pos")

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
