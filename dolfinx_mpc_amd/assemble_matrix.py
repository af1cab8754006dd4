"""``assemble_matrix`` / ``create_matrix`` / ``create_sparsity_pattern`` with the
reference's signatures (python/src/dolfinx_mpc/assemble_matrix.py:21-146),
dispatching to the HIP kernels through the C ABI (include/mpcx.h)."""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Union

import numpy as np

from .common import timed
from . import _device as D
from . import _native
from .fem import DirichletBC, Form
from .la import MPCMatrix
from .multipointconstraint import MultiPointConstraint

_ALG = {"auto": 0, "atomic": 1, "rowblock": 2}
# cluster kernel (MPCX_ALG_CUBE): row blocks of the cluster path (LDS: max_nnz * 8 B + max_rows * 4 B)
CUBE_MAX_NNZ = int(os.environ.get("MPCX_CUBE_MAX_NNZ", 9216))
CUBE_MAX_ROWS = int(os.environ.get("MPCX_CUBE_MAX_ROWS", 512))
# hexahedra (27 entries per Q1 row): 256 rows = 55 KB, two workgroups per CU; a tile of 512 nodes is cut in two
HEX_MAX_ROWS = int(os.environ.get("MPCX_HEX_MAX_ROWS", 256))
HEX_MAX_NNZ = int(os.environ.get("MPCX_HEX_MAX_NNZ", 9216))

# LDS budget of one row block: max_nnz * (8 B value + 4 B column) + row offsets
# (measured on MI355X, tools/sweep_rowblock.py: 512 rows x 9216 nnz = 76 KB of LDS
# -> 2 workgroups of 512 threads per CU is the fastest shape for P1)
ROWBLOCK_MAX_NNZ = int(os.environ.get("MPCX_ROWBLOCK_MAX_NNZ", 9216))
ROWBLOCK_MAX_ROWS = int(os.environ.get("MPCX_ROWBLOCK_MAX_ROWS", 512))
# node blocks that the launch expands to scalar CSR values (MPCX_BLOCK_SCALAR=0, or no overlay to add into): slots per block
NODEBLOCK_CSR_MAX_SLOTS = int(os.environ.get("MPCX_NODEBLOCK_CSR_MAX_SLOTS", 8192))
# the lean scalar P1 kernel needs 64 VGPRs only: half-size blocks (half tiles of an 8x8x8 numbering) put
# four workgroups of 512 threads on a CU -- 1.75 ms against 1.82 ms at config 2 (sweep in DESIGN.md section 5)
ROWBLOCK_LIGHT_MAX_NNZ = int(os.environ.get("MPCX_ROWBLOCK_LIGHT_MAX_NNZ", 4608))
ROWBLOCK_LIGHT_MAX_ROWS = int(os.environ.get("MPCX_ROWBLOCK_LIGHT_MAX_ROWS", 256))
# slave entities x element-tensor entries above which the master contributions skip the host-built plan
MPC_PLAN_MAX_ENTRIES = int(float(os.environ.get("MPCX_MPC_PLAN_MAX_ENTRIES", 1e8)))
# scatter-offset rows are dictionary-compressed when at most this many are distinct (table stays cache resident)
MAX_OFFSET_PATTERNS = 4096


def _warn_no_locality(kind: str, slots: int, entities: int, limit: float = 4.0):
    """Row blocks are contiguous CSR row ranges held in LDS; every entity is listed under each block it touches.  On
    a numbering without locality (a mesh as a file may deliver it) nearly every entity touches as many blocks as it
    has dofs: correct, but 5-13 x slower.  Meshes of MPCX_AUTO_REORDER_MIN_CELLS cells and more never get
    here: they are assembled on the internal, spatially reordered twin (locality.py).  For the small ones, and with
    MPCX_AUTO_REORDER=0, say so once, with the remedy."""
    if entities > 4096 and slots > limit * entities:
        import warnings

        warnings.warn(f"dolfinx_mpc_amd: the {kind} plan lists {slots / entities:.1f} row blocks per entity -- the dof numbering "
                      "has no spatial locality, the row-block kernels will run several times slower than they can.  "
                      "Set MPCX_AUTO_REORDER=1 (the library then assembles on an internal, spatially reordered copy; automatic "
                      "from MPCX_AUTO_REORDER_MIN_CELLS = 50 000 cells), or renumber the mesh once with "
                      "dolfinx_mpc_amd.mesh.reorder_spatial(mesh) before creating function spaces.", RuntimeWarning, stacklevel=3)


def _pair(constraint):
    if isinstance(constraint, MultiPointConstraint):
        return constraint, constraint
    assert len(constraint) == 2
    return constraint[0], constraint[1]


def _pattern_on_device(V0, V1, mpc0, mpc1, keep_on_device: bool = False):
    """(rowptr int64, cols int32) built by the HIP kernels (include/mpcx.h, mpcx_pattern_device_*), as
    numpy arrays or -- ``keep_on_device`` -- as device tensors; None if a row block has more distinct
    column blocks than the kernel holds in LDS."""
    import torch

    L = _native.lib()
    dev = _native.require_gpu()
    st = D.stream_ptr()
    s0, s1 = D.space_device(V0), D.space_device(V1)
    dm0, dm1 = s0["dofmap"], s1["dofmap"]
    nc, nd0, nd1 = dm0.shape[0], dm0.shape[1], dm1.shape[1]
    bs0, bs1 = V0.dofmap.bs, V1.dofmap.bs
    nb0 = V0.num_dofs // bs0

    def dev_mpc(m):
        t = m.device_tensors()  # resident since finalize() when it ran on the device
        return (t["c2s_off"], t["c2s"], t["moff"], t["masters"])

    c0, c1 = dev_mpc(mpc0), dev_mpc(mpc1)
    counter = torch.zeros(nb0, dtype=torch.int32, device=dev)
    rc = L.mpcx_pattern_device_adjacency(nc, dm0.data_ptr(), nd0, bs0, c0[0].data_ptr(), c0[1].data_ptr(),
                                         c0[2].data_ptr(), c0[3].data_ptr(), None, counter.data_ptr(), None, st)
    _native.check(rc, "mpcx_pattern_device_adjacency")
    adj_off = torch.zeros(nb0 + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counter, 0, out=adj_off[1:])
    nadj = int(adj_off[-1].item())
    adj = torch.empty(nadj, dtype=torch.int32, device=dev)
    counter.zero_()
    rc = L.mpcx_pattern_device_adjacency(nc, dm0.data_ptr(), nd0, bs0, c0[0].data_ptr(), c0[1].data_ptr(),
                                         c0[2].data_ptr(), c0[3].data_ptr(), adj_off.data_ptr(), counter.data_ptr(),
                                         adj.data_ptr(), st)
    _native.check(rc, "mpcx_pattern_device_adjacency")
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    row_count = counter  # reuse
    args = (nb0, adj_off.data_ptr(), adj.data_ptr(), dm1.data_ptr(), nd1, bs1, c1[0].data_ptr(), c1[1].data_ptr(),
            c1[2].data_ptr(), c1[3].data_ptr(), row_count.data_ptr())
    _native.check(L.mpcx_pattern_device_rows(*args, None, bs0, None, flag.data_ptr(), st), "mpcx_pattern_device_rows")
    if int(flag.item()) != 0:
        return None
    per_row = (row_count.to(torch.int64) * bs1).repeat_interleave(bs0) if bs0 > 1 else row_count.to(torch.int64) * bs1
    rowptr64 = torch.zeros(nb0 * bs0 + 1, dtype=torch.int64, device=dev)
    torch.cumsum(per_row, 0, out=rowptr64[1:])
    nnz = int(rowptr64[-1].item())
    cols = torch.empty(nnz, dtype=torch.int32, device=dev)
    _native.check(L.mpcx_pattern_device_rows(*args, rowptr64.data_ptr(), bs0, cols.data_ptr(), flag.data_ptr(), st),
                  "mpcx_pattern_device_rows")
    if keep_on_device:
        return rowptr64, cols
    return rowptr64.cpu().numpy(), cols.cpu().numpy()


@timed("~MPC: Create sparsity pattern")
def create_sparsity_pattern(form: Form, mpc: Union[MultiPointConstraint, Sequence[MultiPointConstraint]],
                            num_threads: int = 0, where: Optional[str] = None, keep_on_device: bool = False):
    """MPC sparsity pattern as scalar CSR ``(rowptr int64, cols int32)`` with sorted columns:
    the pattern cpp/utils.h:381-496 inserts into a dolfinx SparsityPattern,
    after ``finalize()`` (python/src/dolfinx_mpc/assemble_matrix.py:68-88).

    ``where``: "device" (HIP kernels, 0.5 s at config 2 including the transfers), "host"
    (threaded C++ builder, 2.6 s) or None = env MPCX_PATTERN, default: device when a GPU is
    present; both give the same arrays.  ``keep_on_device``: return device tensors when the device
    builder ran (create_matrix does: the pattern is consumed there)."""
    mpc0, mpc1 = _pair(mpc)
    mpc0._not_finalized()
    mpc1._not_finalized()
    if form.rank != 2:
        raise RuntimeError("Cannot create sparsity pattern. Form is not a bilinear form")
    V0, V1 = mpc0.function_space, mpc1.function_space
    L = _native.lib()
    p = _native._ptr
    dm0, dm1 = V0.dofmap.list, V1.dofmap.list
    assert dm0.shape[0] == dm1.shape[0]
    if where is None:
        where = os.environ.get("MPCX_PATTERN")
    if where is None:
        import torch

        where = "device" if torch.cuda.is_available() else "host"
    if where.lower() == "device":
        out = _pattern_on_device(V0, V1, mpc0, mpc1, keep_on_device)
        if out is not None:
            return out
    if num_threads <= 0:
        num_threads = min(os.cpu_count() or 1, 16)
    h = L.mpcx_pattern_build(
        dm0.shape[0], p(dm0), dm0.shape[1], V0.dofmap.bs, V0.num_dofs // V0.dofmap.bs, p(dm1), dm1.shape[1],
        V1.dofmap.bs, V1.num_dofs // V1.dofmap.bs,
        p(mpc0.cell_to_slaves.offsets), p(mpc0.cell_to_slaves.array), p(mpc0.masters.offsets), p(mpc0.masters.array),
        p(mpc1.cell_to_slaves.offsets), p(mpc1.cell_to_slaves.array), p(mpc1.masters.offsets), p(mpc1.masters.array),
        num_threads,
    )
    if not h:
        raise RuntimeError("mpcx_pattern_build failed: " + L.mpcx_last_error().decode())
    try:
        rowptr = np.empty(L.mpcx_pattern_nrows(h) + 1, dtype=np.int64)
        cols = np.empty(L.mpcx_pattern_nnz(h), dtype=np.int32)
        L.mpcx_pattern_copy(h, p(rowptr), p(cols))
    finally:
        L.mpcx_pattern_free(h)
    return rowptr, cols


@timed("~MPC: Create Matrix")
def create_matrix(form: Form, mpc0: MultiPointConstraint, mpc1: Optional[MultiPointConstraint] = None) -> MPCMatrix:
    """python/src/dolfinx_mpc/mpc.cpp:321-344 ``cpp.mpc.create_matrix``."""
    mpc1 = mpc0 if mpc1 is None else mpc1
    rowptr, cols = create_sparsity_pattern(form, (mpc0, mpc1), keep_on_device=True)
    A = MPCMatrix(rowptr, cols, mpc1.function_space.num_dofs, dtype=getattr(form, "dtype", None))
    # partitioned mesh: A.assemble() ships the interface rows to their owner (assemble_matrix.py:64)
    from .distributed import exchange_for

    V0, V1 = mpc0.function_space, mpc1.function_space
    part = getattr(V0.mesh, "partition", None)
    if part is not None and part.get("world", 1) > 1:
        if V0 is V1:
            ex = exchange_for(V0, A)
            if ex is not None:
                A.attach_exchange(ex)
        else:
            import torch.distributed as dist

            if dist.is_available() and dist.is_initialized():
                raise NotImplementedError("partitioned meshes: square blocks (test space == trial space) only")
    return A


def _torch_dtype_of(dtype):
    from .la import _torch_dtype

    return _torch_dtype(dtype)


def _slave_entities(form: Form, i: int, mpc0, mpc1):
    """entity indices of integral i whose cell holds a slave of mpc0 or mpc1."""
    def build():
        import torch

        integ = form.integrals[i]
        idv = D.integral_device(form, i)
        off0 = mpc0.device_tensors()["c2s_off"]
        has = (off0[1:] - off0[:-1]) > 0  # per cell, on the device (100 M cells at config 2: no host pass, no download)
        if mpc1 is not mpc0:
            off1 = mpc1.device_tensors()["c2s_off"]
            has = has | ((off1[1:] - off1[:-1]) > 0)
        if idv["entities"] is None:  # cells 0..n-1 in order (the usual domain): no gather through the entity list
            has = has[: integ.num_entities]
        else:
            has = has[idv["entities"].view(integ.num_entities, integ.estride)[:, 0].long()]
        idx = torch.nonzero(has).reshape(-1).to(torch.int32).contiguous()
        return (idx.cpu().numpy(), idx)

    return D.cached(form._device, "slave_ents", (mpc0, mpc1), i, build)


def _block_ranges(nrows: int, rowptr: np.ndarray, max_rows: int, max_nnz: int, bs: int, hints) -> np.ndarray:
    """contiguous row ranges of a row-block plan (host: one greedy pass over rowptr), block_row0 [nb + 1];
    ``rowptr=None``: one entry per row (the vector plans)"""
    L = _native.lib()
    p = _native._ptr
    hp, hn = (None, 0) if hints is None else (p(hints), hints.size)
    rp = None if rowptr is None else p(rowptr)
    nb = L.mpcx_block_ranges(nrows, rp, max_rows, max_nnz, bs, hp, hn, None, 0)
    if nb < 0:
        msg = L.mpcx_last_error().decode()
        if "exceeds the block capacity" in msg:
            # one row longer than a block's LDS (a master with thousands of slaves): 'auto' falls back to the
            # thread-per-entity kernels with device atomics
            raise _native.PlanNotRepresentable("row-block algorithm: " + msg)
        raise RuntimeError("mpcx_block_ranges failed: " + msg)
    row0 = np.empty(nb + 1, dtype=np.int32)
    if L.mpcx_block_ranges(nrows, rp, max_rows, max_nnz, bs, hp, hn, p(row0), row0.size) != nb:
        raise RuntimeError("mpcx_block_ranges failed: " + L.mpcx_last_error().decode())
    return row0


def _block_lists_device(row0: np.ndarray, n_entities: int, estride: int, entities_ptr, dofmap_dev, nd: int, bs: int, dev,
                        group_rows: bool = False, rotate: bool = False):
    """entities touching every row block, built on the device (mpcx_rowblock_pairs_device: count -> scan -> fill
    in entity order, then a stable sort by block: torch, plumbing).  Returns (block_row0, block_ent_off, block_ents)
    device tensors; the lists are ordered by entity inside each block, like the host builder's, or (group_rows) by
    the set of local rows the entity has inside the block, then by entity."""
    import torch

    L = _native.lib()
    st = D.stream_ptr()
    nb = row0.size - 1
    if nd > 32:
        # (Q3 hexahedra: 64 nodes per cell) the list builder keeps an entity's blocks in a 32-entry register array
        raise _native.PlanNotRepresentable("row-block plan: more than 32 dof blocks per entity; the per-entity kernels take it")
    d_row0 = D._to_dev(row0, dev)
    counts = torch.empty(max(n_entities, 1), dtype=torch.int32, device=dev)
    args = (n_entities, estride, entities_ptr, dofmap_dev.data_ptr(), nd, bs, nb, d_row0.data_ptr(), counts.data_ptr())
    _native.check(L.mpcx_rowblock_pairs_device(*args, None, None, None, None, 0, st), "mpcx_rowblock_pairs_device")
    from . import _prims

    # scans, the stable sort by block and the per-block offsets are rocPRIM behind the C ABI (mpcx_scan_exclusive_*,
    # mpcx_sort_pairs_*, mpcx_segment_offsets); torch only allocates and forms the sort key
    scan = _prims.scan_i32_i64(counts[:n_entities])
    offsets = scan[:-1]
    total = int(scan[-1].item()) if n_entities else 0
    pair_block = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
    pair_ent = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
    pair_rows = torch.empty(max(total, 1), dtype=torch.int32, device=dev) if group_rows else None
    if total:
        _native.check(L.mpcx_rowblock_pairs_device(*args, offsets.data_ptr(), pair_block.data_ptr(), pair_ent.data_ptr(),
                                                   D.ptr(pair_rows), int(rotate), st), "mpcx_rowblock_pairs_device")
    pair_block, pair_ent = pair_block[:total], pair_ent[:total]
    if total:
        if group_rows:
            # entities that keep the same local rows next to each other: a wave then skips the rows of dofs outside
            # the block as a whole instead of issuing their scatter-adds with most lanes masked off
            shift = nd
            key = (pair_block.to(torch.int64) << nd) | (pair_rows[:total].to(torch.int64) & ((1 << nd) - 1))
        else:
            shift = 0
            key = pair_block.to(torch.int64)
        key, ents = _prims.sort_pairs(key, pair_ent.contiguous(), shift + max(int(nb).bit_length(), 1))
        off = _prims.segment_offsets(key, shift, nb)
        del key
    else:
        ents = pair_ent
        off = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
    return d_row0, off, ents


def _block_pairs_device(row0: np.ndarray, n_entities: int, estride: int, entities_dev, dofmap_dev, nd: int, bs: int, dev):
    """(entity, local row dof) pairs of every row block for the row-pair kernel (include/mpcx.h,
    mpcx_rowblock_plan_t::row_pairs), built on the device with torch (plumbing: gather, searchsorted, two sorts).
    Pair (e, i) belongs to the block that holds the rows of dof i of entity e, so every pair appears exactly once.
    Inside a block the pairs are ordered by local row i (a wave runs one unrolled row body), then round-robin over
    the row dofs (rank of the pair among the pairs of its dof, then dof), so neighbouring lanes add into different
    CSR rows.  Returns (block_row0, block_pair_off, pair ids = e * nd + i as int32)."""
    import torch

    nb = row0.size - 1
    d_row0 = D._to_dev(row0, dev)
    if n_entities * nd >= 2 ** 31:
        raise _native.PlanNotRepresentable("row-pair plan: entity * nd + i does not fit 32 bits")
    if entities_dev is None:
        dof = dofmap_dev[:n_entities].reshape(-1)
    else:
        dof = dofmap_dev[entities_dev.view(n_entities, estride)[:, 0].long()].reshape(-1)
    M = dof.numel()
    if M == 0:
        return (d_row0, torch.zeros(nb + 1, dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.int32, device=dev))
    blk = torch.searchsorted(d_row0[1:].contiguous(), (dof * bs).contiguous(), right=True).to(torch.int64)
    loc = dof.to(torch.int64) - (d_row0.to(torch.int64)[blk] // bs)  # dof inside its block: < 2^16 rows
    li = torch.arange(M, device=dev, dtype=torch.int64) % nd
    assert nd <= 32
    key = (blk << 41) | (li << 36) | loc  # bits: loc 0-23, rank 24-35 (below), local index 36-40, block 41-
    del li
    key, order = torch.sort(key, stable=True)
    _, counts = torch.unique_consecutive(key, return_counts=True)
    starts = torch.cumsum(counts, 0) - counts
    rank = torch.arange(M, device=dev, dtype=torch.int64) - torch.repeat_interleave(starts, counts)
    del starts, counts
    if int(rank.max().item()) >= 2 ** 12 or int(loc.max().item()) >= 2 ** 24:
        raise _native.PlanNotRepresentable("row-pair plan: more than 4096 entities round one dof")
    del loc
    key = (key & ~((1 << 36) - 1)) | (rank << 24) | (key & ((1 << 24) - 1))
    del rank
    _, order2 = torch.sort(key)
    del key
    ids = order[order2].to(torch.int32).contiguous()
    del order, order2
    per_block = torch.bincount(blk, minlength=nb)
    off = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
    torch.cumsum(per_block, 0, out=off[1:])
    return d_row0, off, ids


def _diag_blocked(form: Form, i: int, V0, V1) -> bool:
    """component-diagonal operator on a blocked space (S (x) I: stiffness, mass, facet mass)"""
    return form.integrals[i].kernel.form in (0, 1, 4) and V0.dofmap.bs > 1 and V1.dofmap.bs == V0.dofmap.bs


def _slot_mask(A: MPCMatrix, form: Form, V0, V1, bc0, bc1, mpc0, mpc1):
    """mpcx_matrix_args_t::slot_mask of the node-block kernel: one byte per bs x bs block of A, bit k = the (k, k)
    entry lies in a Dirichlet / slave row or column.  None if A is not made of whole blocks."""
    import torch

    def build():
        bs = V0.dofmap.bs
        L = _native.lib()
        if A.shape[0] % bs or A.nnz % (bs * bs):
            return None
        out = torch.zeros(max(A.nnz // (bs * bs), 1), dtype=torch.uint8, device=A.device)
        bad = torch.zeros(1, dtype=torch.int32, device=A.device)
        _, k0 = mpc0._device()
        _, k1 = mpc1._device()
        rc = L.mpcx_diag_slot_mask(A.shape[0] // bs, A.d_rowptr.data_ptr(), A.d_cols.data_ptr(), bs, D.ptr(bc0),
                                   k0["is_slave"].data_ptr(), D.ptr(bc1), k1["is_slave"].data_ptr(), out.data_ptr(),
                                   bad.data_ptr(), D.stream_ptr())
        _native.check(rc, "mpcx_diag_slot_mask")
        return None if int(bad.item()) else out

    return D.cached(A._plans, "slot_mask", (V0, V1, mpc0, mpc1, bc0, bc1), 0, build, maxsize=4)


def _rowblock_plan(A: MPCMatrix, form: Form, i: int, V0, lean: bool = False, pairs: bool = False, nodeblock: bool = False,
                   csr_valued: bool = False):
    light = lean and V0.dofmap.bs == 1 and V0.element_ndofs <= 4
    max_rows_cap, max_nnz_cap = ((ROWBLOCK_LIGHT_MAX_ROWS, ROWBLOCK_LIGHT_MAX_NNZ) if light
                                 else (ROWBLOCK_MAX_ROWS, ROWBLOCK_MAX_NNZ))
    if np.dtype(getattr(form, "dtype", np.float64)).kind == "c":
        max_nnz_cap //= 2  # complex: 16 bytes per LDS value (complex64 accumulates in fp64 pairs as well, csrc/mpcx_scalar.hip)
    kf = form.integrals[i].kernel
    # entities of a block ordered by which of their local rows lie inside it (measured: P1 elasticity 1.82 -> 1.49 ms,
    # Taylor-Hood coupling blocks 2.0 -> 1.8, P2 stiffness +1.5 %; the light P1 kernel loses its coordinate locality,
    # 2.07 -> 2.81 ms, and the compact component-diagonal layout 2.5 %: both keep entity order)
    group_rows = not light and not os.environ.get("MPCX_NO_GROUP_ROWS")
    if nodeblock:
        # component-diagonal forms, node-block kernel: one LDS value per bs x bs block (matrix_nodeblock_kernel)
        group_rows = True
        if csr_valued:
            # expanded to scalar CSR values by the launch: the write-out reads the masks from an LDS copy (one more byte per
            # slot) and is the long phase.  Taylor-Hood a00 128^3, 1024 threads: 9216 slots 9.8 ms, 8192 9.0, 6144 9.4, 4608 9.7
            max_nnz_cap = min(max_nnz_cap, NODEBLOCK_CSR_MAX_SLOTS)
        max_rows_cap, max_nnz_cap = max_rows_cap * V0.dofmap.bs, max_nnz_cap * V0.dofmap.bs ** 2
    elif (kf.form in (0, 1, 4) and V0.dofmap.bs > 1 and not os.environ.get("MPCX_NO_DIAG_COMPACT")
          and _native.scalar_id(getattr(form, "dtype", np.float64)) == 0):
        group_rows = False
        # component-diagonal forms on blocked spaces: the kernel keeps one LDS value per column block, so a
        # workgroup owns bs times more rows (include/mpcx.h, matrix_rowblock_kernel)
        max_rows_cap, max_nnz_cap = max_rows_cap * V0.dofmap.bs, max_nnz_cap * V0.dofmap.bs
    def build():
        L = _native.lib()
        p = _native._ptr
        integ = form.integrals[i]
        dm = V0.dofmap.list
        # numbering hint: first row of every tile of a tiled P1 numbering
        hints = None
        if V0.dof_tile_offsets is not None:
            hints = np.ascontiguousarray(V0.dof_tile_offsets.astype(np.int32) * V0.dofmap.bs)
        dev = A.device
        if pairs:
            row0 = _block_ranges(A.shape[0], A.rowptr, max_rows_cap, max_nnz_cap, V0.dofmap.bs, hints)
            nb = row0.size - 1
            lists = _block_pairs_device(row0, integ.num_entities, integ.estride, D.integral_device(form, i)["entities"],
                                        D.space_device(V0)["dofmap"], V0.element_ndofs, V0.dofmap.bs, dev)
        elif os.environ.get("MPCX_PLAN_LISTS", "device") == "host":
            ents = np.ascontiguousarray(integ.entities.astype(np.int32).reshape(-1))
            h = L.mpcx_rowblock_plan_build(A.shape[0], p(A.rowptr), max_rows_cap, max_nnz_cap,
                                           integ.num_entities, integ.estride, p(ents), p(dm), dm.shape[1], V0.dofmap.bs,
                                           None if hints is None else p(hints), 0 if hints is None else hints.size, 1)
            if not h:
                msg = L.mpcx_last_error().decode()
                if "exceeds the block capacity" in msg:
                    raise _native.PlanNotRepresentable("row-block algorithm: " + msg)
                raise RuntimeError("mpcx_rowblock_plan_build failed: " + msg)
            try:
                nb = L.mpcx_rowblock_plan_num_blocks(h)
                row0 = np.empty(nb + 1, dtype=np.int32)
                off = np.empty(nb + 1, dtype=np.int64)
                ents_b = np.empty(L.mpcx_rowblock_plan_num_ents(h), dtype=np.int32)
                L.mpcx_rowblock_plan_copy(h, p(row0), p(off), p(ents_b))
            finally:
                L.mpcx_rowblock_plan_free(h)
            lists = (D._to_dev(row0, dev), D._to_dev(off, dev), D._to_dev(ents_b, dev))
        else:
            row0 = _block_ranges(A.shape[0], A.rowptr, max_rows_cap, max_nnz_cap, V0.dofmap.bs, hints)
            nb = row0.size - 1
            lists = _block_lists_device(row0, integ.num_entities, integ.estride, D.integral_device(form, i)["entities_ptr"],
                                        D.space_device(V0)["dofmap"], V0.element_ndofs, V0.dofmap.bs, dev,
                                        group_rows=group_rows, rotate=lean)
        # 8-bit scatter offsets of every (entity, local row, local col), built on the device
        import torch

        V1 = form.function_spaces[1]
        s0, s1 = D.space_device(V0), D.space_device(V1)
        idv = D.integral_device(form, i)
        offs = torch.empty(integ.num_entities * V0.element_ndofs * V1.element_ndofs, dtype=torch.uint8, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        rc = L.mpcx_scatter_offsets(A.d_rowptr.data_ptr(), A.d_cols.data_ptr(), integ.estride, integ.num_entities,
                                    idv["entities_ptr"], idv["entities_ptr"], s0["dofmap"].data_ptr(),
                                    V0.element_ndofs, V0.dofmap.bs, s1["dofmap"].data_ptr(), V1.element_ndofs,
                                    V1.dofmap.bs, int(lean), offs.data_ptr(), flag.data_ptr(), D.stream_ptr())
        _native.check(rc, "mpcx_scatter_offsets")
        if int(flag.item()) != 0:
            raise _native.PlanNotRepresentable(
                "row-block algorithm: a CSR row holds more than 255 column blocks before one of the "
                "entity's columns (or a column is missing from the pattern); use algorithm='atomic'")
        # dictionary compression: structured / tiled meshes have few distinct offset rows, so the
        # kernel reads a 2-byte id per entity plus a cache-resident table instead of nd0*nd1 bytes
        noff = V0.element_ndofs * V1.element_ndofs
        pattern = None
        npat = -1
        if os.environ.get("MPCX_OFFSET_DICT") and integ.num_entities > 0:
            offs_h = offs.cpu().numpy()
            ids = np.empty(integ.num_entities, dtype=np.uint16)
            table = np.empty(MAX_OFFSET_PATTERNS * noff, dtype=np.uint8)
            npat = L.mpcx_compress_offsets(p(offs_h), integ.num_entities, noff, MAX_OFFSET_PATTERNS, p(ids), p(table))
            if npat > 0:
                offs = D._to_dev(table[: npat * noff].copy(), dev)
                pattern = D._to_dev(ids.view(np.int16), dev)  # torch has no uint16 on every build: same bits
        t = lists + (offs, pattern)
        if not pairs:
            _warn_no_locality("row-block", int(t[2].numel()), integ.num_entities)
        max_rows = int(np.diff(row0).max())
        max_nnz = int(np.diff(A.rowptr[row0]).max())
        s = _native.RowBlockPlanT(nb, max_rows, max_nnz, int(pairs), t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(),
                                  t[3].data_ptr(), D.ptr(pattern))
        return (s, t, {"num_blocks": nb, "num_ents": int(t[2].numel()), "max_rows": max_rows,
                       "max_nnz": max_nnz, "offset_patterns": npat,
                       "bytes": int(sum(x.numel() * x.element_size() for x in t if x is not None))})

    return D.cached(A._plans, "rowblock", (form,), (i, max_nnz_cap, max_rows_cap, lean, group_rows, pairs, nodeblock), build)


# row blocks of the pair-record kernel (csrc/mpcx_pairs.hip): LDS bytes = 8 * nnz (+ 4 * rows for blocked spaces)
# Measured (MI355X): scalar P2 stiffness 246^3: 2304 entries per block 10.0 ms, 4608 (37 KB, four workgroups of 512 threads per
# CU) 9.5, 9216 10.2; Taylor-Hood a01 / a10 128^3: 4608 1.94 / 1.58 ms, 2304 1.89 / 1.38; contact elasticity: 4608 1.14,
# 2304 0.94 -- the short pair lists of a block are a chain of dependent phases (zero, records, contexts, write-out), so
# the CU wants many small resident blocks, but a block whose (local row) segments are shorter than a wave loses lanes
def _pairs_caps(V0, V1):
    """(max rows, max entries) of a row block; MPCX_PAIRS_MAX_NNZ / MPCX_PAIRS_MAX_ROWS override"""
    rows = int(os.environ.get("MPCX_PAIRS_MAX_ROWS", 512))
    nnz = int(os.environ.get("MPCX_PAIRS_MAX_NNZ", 0))
    if nnz > 0:
        return rows, nnz
    PAIRS_MAX_ROWS = rows
    scalar_p2 = V0.dofmap.bs == 1 and V1.dofmap.bs == 1 and V0.element_ndofs >= 6 and V1.element_ndofs >= 6
    return PAIRS_MAX_ROWS, (4608 if scalar_p2 else 2304)


def _pairs_plan(A: MPCMatrix, form: Form, i: int, V0, V1, bc0, bc1, mpc0, mpc1):
    """Plan of ``matrix_pairs_kernel`` (include/mpcx.h: plan.row_pairs == 2, pair_recs): row blocks, the (entity, local
    row) pairs of every block ordered by local row, ONE record per pair.  Cached per (form, constraints, Dirichlet
    markers): the records carry the row / column masks.  Returns (plan struct, keep-alive, info)."""
    import torch

    caps = _pairs_caps(V0, V1)

    def build():
        L = _native.lib()
        integ = form.integrals[i]
        dev = A.device
        bs0 = V0.dofmap.bs
        hints = None
        if V0.dof_tile_offsets is not None:
            hints = np.ascontiguousarray(V0.dof_tile_offsets.astype(np.int32) * bs0)
        row0 = _block_ranges(A.shape[0], A.rowptr, caps[0], caps[1], bs0, hints)
        nb = row0.size - 1
        idv = D.integral_device(form, i)
        s0, s1 = D.space_device(V0), D.space_device(V1)
        d_row0, d_off, d_ids = _block_pairs_device(row0, integ.num_entities, integ.estride, idv["entities"], s0["dofmap"],
                                                   V0.element_ndofs, bs0, dev)
        npairs = integ.num_entities * V0.element_ndofs
        W = L.mpcx_pair_words(V1.element_ndofs)
        recs = torch.empty(max(npairs, 1) * W, dtype=torch.int32, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        _, k0 = mpc0._device()
        _, k1 = mpc1._device()
        rc = L.mpcx_pair_records(npairs, d_ids.data_ptr(), integ.estride, idv["entities_ptr"], idv["entities_ptr"],
                                 s0["dofmap"].data_ptr(), V0.element_ndofs, bs0, s1["dofmap"].data_ptr(), V1.element_ndofs,
                                 V1.dofmap.bs, D.ptr(bc0), k0["is_slave"].data_ptr(), D.ptr(bc1), k1["is_slave"].data_ptr(),
                                 A.d_rowptr.data_ptr(), A.d_cols.data_ptr(), nb, d_row0.data_ptr(), recs.data_ptr(),
                                 flag.data_ptr(), D.stream_ptr())
        _native.check(rc, "mpcx_pair_records")
        bad = int(flag.item())
        if bad:
            raise _native.PlanNotRepresentable(
                "pair records: " + ", ".join(m for b, m in ((1, "a scatter offset beyond 8 bits or a column missing from the pattern"),
                                                            (2, "a row slot beyond its field"),
                                                            (4, "more than 2^27 entities (shard the mesh)")) if bad & b))
        del d_ids
        # dictionary of offset patterns (opt-in, MPCX_PAIRS_DICT=1): a structured mesh has a few thousand whatever its size
        # (P2 on a tiled Kuhn mesh: 1 500), so a record shrinks to two words -- half the plan memory, but the dependent
        # table load costs more than the bytes it saves: P2 Poisson 246^3 9.5 -> 11.3 ms.  More than 65535 patterns
        # (unstructured meshes): full records
        table, npat = None, -1
        if npairs > 0 and os.environ.get("MPCX_PAIRS_DICT", "0") == "1":
            ws = torch.empty(L.mpcx_pair_compress_workspace(V1.element_ndofs), dtype=torch.uint8, device=dev)
            recs2 = torch.empty(npairs * 2, dtype=torch.int32, device=dev)
            ds = L.mpcx_pair_dict_stride(V1.element_ndofs)
            table = torch.empty(65535 * ds, dtype=torch.int32, device=dev)
            n = C.c_int32(0)
            _native.check(L.mpcx_pair_compress(npairs, recs.data_ptr(), V1.element_ndofs, recs2.data_ptr(), table.data_ptr(),
                                               C.byref(n), ws.data_ptr(), D.stream_ptr()), "mpcx_pair_compress")
            npat = int(n.value)
            if npat >= 0:
                recs, table = recs2, table[: max(npat, 1) * ds].clone()
            else:
                table = None
            del ws
        max_rows = int(np.diff(row0).max())
        max_nnz = int(np.diff(A.rowptr[row0]).max())
        plan = _native.RowBlockPlanT(nb, max_rows, max_nnz, 2, d_row0.data_ptr(), d_off.data_ptr(), None, None, None)
        keep = (d_row0, d_off, recs, table)
        return (plan, keep, {"num_blocks": nb, "num_ents": npairs, "max_rows": max_rows, "max_nnz": max_nnz,
                             "offset_patterns": npat,
                             "bytes": int(recs.numel() * 4 + d_off.numel() * 8 + d_row0.numel() * 4)})

    return D.cached(A._plans, "pairs", (form, mpc0, mpc1, bc0, bc1), (i,) + caps + (os.environ.get("MPCX_PAIRS_DICT", "0"),), build)


def _pair_context(form: Form, i: int):
    """per-entity constant-free context of the pair-record kernel (mpcx_pair_context), cached per geometry version;
    None with MPCX_PAIRS_CONTEXT=recompute (the kernel then computes it per pair from the coordinates)"""
    import torch

    if os.environ.get("MPCX_PAIRS_CONTEXT", "cached") == "recompute":
        return None

    def build():
        L = _native.lib()
        integ = form.integrals[i]
        idv = D.integral_device(form, i)
        md = D.mesh_device(form.mesh)
        cn = L.mpcx_pair_context_size(C.byref(idv["kernel"]))
        if cn <= 0:
            raise _native.PlanNotRepresentable("pair records: the operator has no compact context")
        ctx = torch.empty(max(integ.num_entities, 1) * cn, dtype=torch.float64, device=md["x"].device)
        rc = L.mpcx_pair_context(C.byref(idv["kernel"]), integ.num_entities, integ.estride, idv["entities_ptr"],
                                 md["x"].data_ptr(), md["x_dofmap"].data_ptr(), form.mesh.geometry.dofmap.shape[1],
                                 ctx.data_ptr(), D.stream_ptr())
        _native.check(rc, "mpcx_pair_context")
        return ctx

    return D.cached(form._device, "pair_ctx", (form.mesh,), (i, form.mesh.geometry.version), build, maxsize=4)


def _rowpair_eligible(form: Form, i: int, V0, V1) -> bool:
    """Row-pair kernel (include/mpcx.h, mpcx_rowblock_plan_t::row_pairs): operators with a compact per-entity context
    (csrc/mpcx_elements.hpp, ElementOp::LAZY / lazy_applies): cell integrals of stiffness without coefficient,
    elasticity and the Taylor-Hood coupling blocks.  Chosen for vector-valued P1 only: there a 74 KB row block
    holds ~70 nodes, two thirds of the lanes of a thread-per-cell block are masked off and the context is cheap
    (contact elasticity 1.45 -> 0.96 ms).  P2 pays the context ten times per cell (P2 stiffness 246^3:
    14.3 -> 22.6 ms; Taylor-Hood a00 unchanged, a01 1.9 -> 2.7 ms) and keeps thread-per-cell blocks;
    MPCX_ROWPAIR=all forces it wherever the operator allows."""
    mode = os.environ.get("MPCX_ROWPAIR", "auto")
    if os.environ.get("MPCX_NO_ROWPAIR") or mode == "none":
        return False
    integ = form.integrals[i]
    kf = integ.kernel
    if integ.itype != "cell" or integ.coefficient is not None:
        return False
    d0, d1 = V0.degree, V1.degree
    if mode != "all" and not (d0 == 1 and d1 == 1 and V0.dofmap.bs > 1):
        return False
    if kf.form == 0:
        return kf.coeff_degree == 0 and d0 == d1 and d0 in (1, 2)
    if kf.form == 3:
        return d0 == d1 and d0 in (1, 2)
    if kf.form == 6:
        return kf.coeff_degree == 0 and d0 == 2 and d1 == 1
    if kf.form == 7:
        return kf.coeff_degree == 0 and d0 == 1 and d1 == 2
    return False


def _cube_eligible(form: Form, i: int, V0) -> bool:
    """scalar P1 stiffness on tetrahedra without coefficient, over cells 0..n-1 (MPCX_ALG_CUBE)"""
    integ = form.integrals[i]
    k = integ.kernel
    shape_ok = k.form == 0 and k.bs == 1
    return (shape_ok and k.celltype == 2 and k.degree == 1 and (k.degree1 or 1) == 1 and (k.bs1 or k.bs) == k.bs
            and k.coeff_degree == 0 and integ.coefficient is None and integ.itype == "cell"
            and not os.environ.get("MPCX_NO_CUBE"))


def _cube_plan(A: MPCMatrix, form: Form, i: int, V0, bc_dev, mpc, hexa: bool = False, closed_form_only: bool = False,
               ordered: bool = False):
    """Row blocks over the mesh's cell clusters + one 96-byte record per (block, cluster) slot
    (mpcx_cube_records); cached per (form, constraint, Dirichlet markers).  Returns
    (plan struct, records tensor, keep-alive, info, leftover cells) or None when the mesh has no clusters.
    ``hexa``: the clusters are the hexahedra themselves (their Q1 dofmap; every vertex pair coupled: mpcx_hex_records).
    ``ordered`` (imported kernels): only clusters whose cells the mesh lists in the cluster kernels' own vertex order
    (clusters.mesh_clusters_ordered_device); every part then carries the cluster of each of its slots (block_ents) and the
    keep-alive tuple ends with the clusters' cells (mpcx_matrix_args_t::cube_cells)."""
    import torch

    from .clusters import mesh_clusters_device, mesh_clusters_ordered_device

    integ = form.integrals[i]
    d_cells = None
    if ordered:
        d_verts, left, d_cells = mesh_clusters_ordered_device(form.mesh, integ.num_entities)
        if d_verts.shape[0] == 0 or d_verts.shape[0] * 6 < 0.5 * integ.num_entities:
            return None
        max_rows_cfg, max_nnz_cfg = CUBE_MAX_ROWS, CUBE_MAX_NNZ
    elif hexa:
        d_verts = D.space_device(V0)["dofmap"].view(-1, 8)[: integ.num_entities]
        left = np.zeros(0, dtype=np.int32)
        max_rows_cfg, max_nnz_cfg = HEX_MAX_ROWS, HEX_MAX_NNZ
    else:
        # closed_form_only (vector elasticity): the kernel knows parallelepiped clusters only; the cells of every other
        # cluster are leftover cells like those outside any cluster
        d_verts, left = mesh_clusters_device(form.mesh, integ.num_entities, parallelepipeds_only=closed_form_only)
        if d_verts.shape[0] == 0 or d_verts.shape[0] * 6 < 0.5 * integ.num_entities:
            return None
        max_rows_cfg, max_nnz_cfg = CUBE_MAX_ROWS, CUBE_MAX_NNZ

    def build():
        L = _native.lib()
        dev = A.device
        nc = d_verts.shape[0]
        hints = None
        if V0.dof_tile_offsets is not None:
            hints = np.ascontiguousarray(V0.dof_tile_offsets.astype(np.int32))
        bs = V0.dofmap.bs
        if hints is not None:
            hints = np.ascontiguousarray(hints * bs)
        row0 = _block_ranges(A.shape[0], A.rowptr, max_rows_cfg, max_nnz_cfg, bs, hints)
        nb = row0.size - 1
        if closed_form_only:
            # vector elasticity: ONE record per cluster and a row-pair plan over the clusters -- the unit of work is a
            # (cluster, local row vertex) pair whose rows lie in the block (pair id = cluster * 8 + vertex)
            d_row0, d_off, d_pairs = _block_pairs_device(row0, nc, 1, None, d_verts, 8, bs, dev)
            d_ents = torch.arange(nc, dtype=torch.int32, device=dev)
        else:
            d_row0, d_off, d_ents = _block_lists_device(row0, nc, 1, None, d_verts, 8, bs, dev)
        nslots = d_ents.numel()
        if not closed_form_only:
            _warn_no_locality("cluster", int(nslots), nc, limit=5.0)
        # One record per (row block, cluster) slot, streamed.  MPCX_CUBE_CLUSTER_RECORDS=1: ONE record per cluster (nothing in
        # a record depends on the row block) and the slots carry the index of their cluster's record
        # (mpcx_matrix_args_t::cube_rec_index) -- 6 % less traffic (the L2 does not catch all the re-reads of clusters that
        # touch several blocks) but one more dependent load: 1.01 against 0.94 ms at 256^3 cubes, 1.53 against 1.35 ms on
        # hexahedra (round 4), so not the default.
        per_cluster = not closed_form_only and os.environ.get("MPCX_CUBE_CLUSTER_RECORDS", "0") == "1"
        nrec = nc if (per_cluster or closed_form_only) else nslots
        rec_ents = torch.arange(nc, dtype=torch.int32, device=dev) if per_cluster else d_ents
        recs = torch.empty(nrec * 96, dtype=torch.uint8, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        _, t = mpc._device()
        records = L.mpcx_hex_records if hexa else L.mpcx_cube_records
        rc = records(nrec, rec_ents.data_ptr(), d_verts.data_ptr(), bs, D.ptr(bc_dev), t["is_slave"].data_ptr(),
                     A.d_rowptr.data_ptr(), A.d_cols.data_ptr(), recs.data_ptr(), flag.data_ptr(), D.stream_ptr())
        _native.check(rc, "mpcx_hex_records" if hexa else "mpcx_cube_records")
        if int(flag.item()) != 0:
            raise _native.PlanNotRepresentable("cluster algorithm: a scatter offset does not fit 8 bits (or a column is missing)")
        max_rows = int(np.diff(row0).max())
        max_nnz = int(np.diff(A.rowptr[row0]).max())
        if closed_form_only:
            plan = _native.RowBlockPlanT(nb, max_rows, max_nnz, 1, d_row0.data_ptr(), d_off.data_ptr(), d_pairs.data_ptr(), None, None)
            parts = [(plan, recs, 96, None, d_off, 1)]
            keep = (d_row0, parts, d_verts, d_pairs)
            info = {"num_blocks": nb, "num_ents": int(d_pairs.numel()), "max_rows": max_rows, "max_nnz": max_nnz, "clusters": int(nc),
                    "narrow_blocks": 0, "closed_form_blocks": nb,
                    "bytes": int(d_row0.numel() * 4 + recs.numel() + d_off.numel() * 8 + d_pairs.numel() * 4)}
            return (parts, keep, info)
        # Row blocks are launched by KIND: record format (tetrahedral clusters: 64-byte records with 4-bit offsets for the
        # row blocks all of whose slots allow it, 96 bytes for blocks that hold fat rows -- master rows of a constraint) and
        # cell shape (row blocks all of whose clusters / hexahedra are parallelepipeds go to the kernel instance that
        # carries only the closed form of the integral: flag bit 0).  One part, one launch per kind that occurs.
        want_narrow = bs == 1 and not hexa and os.environ.get("MPCX_CUBE_NARROW", "1") != "0"
        want_shape = bs == 1 and not ordered and os.environ.get("MPCX_CUBE_SHAPES", os.environ.get("MPCX_HEX_SPLIT", "1")) != "0"
        kind = torch.zeros(nb, dtype=torch.int64, device=dev)  # per block: bit 0 = a wide slot, bit 1 = a general cell
        slot_rec = d_ents.long() if per_cluster else None  # record of every slot

        def any_per_block(rec_flag):
            slot_flag = rec_flag[slot_rec] if per_cluster else rec_flag
            cs = torch.zeros(nslots + 1, dtype=torch.int64, device=dev)
            torch.cumsum(slot_flag.to(torch.int64), 0, out=cs[1:])
            return (cs[d_off[1:]] - cs[d_off[:-1]]) > 0

        if nslots > 0 and want_narrow:
            wide = torch.empty(nrec, dtype=torch.uint8, device=dev)
            _native.check(L.mpcx_cube_slot_width(nrec, recs.data_ptr(), wide.data_ptr(), D.stream_ptr()), "mpcx_cube_slot_width")
            kind += any_per_block(wide).to(torch.int64)
            del wide
        else:
            kind += 1
        if nslots > 0 and want_shape:
            general = torch.empty(nrec, dtype=torch.uint8, device=dev)
            _native.check(L.mpcx_hex_slot_shapes(nrec, recs.data_ptr(), D.mesh_device(form.mesh)["x"].data_ptr(),
                                                 general.data_ptr(), D.stream_ptr()), "mpcx_hex_slot_shapes")
            kind += 2 * any_per_block(general).to(torch.int64)
            del general
        else:
            kind += 2
        kinds = [int(k) for k in torch.unique(kind).tolist()] if nb > 0 else [3]
        parts = []
        narrow_all = None  # per-cluster narrow records, shared by the narrow launches
        for kd in kinds:
            nbytes, flags = (96 if kd & 1 else 64), (0 if kd & 2 else 1)
            if len(kinds) == 1:
                sel, ids, off_c, src = None, None, d_off, None
                nblk = nb
            else:
                sel = torch.nonzero(kind == kd).reshape(-1)
                cnt = d_off[sel + 1] - d_off[sel]
                off_c = torch.zeros(sel.numel() + 1, dtype=torch.int64, device=dev)
                torch.cumsum(cnt, 0, out=off_c[1:])
                tot = int(off_c[-1].item())
                src = torch.repeat_interleave(d_off[sel] - off_c[:-1], cnt) + torch.arange(tot, dtype=torch.int64, device=dev)
                ids = sel.to(torch.int32).contiguous()
                nblk = int(sel.numel())
            ridx = None
            if per_cluster:
                ridx = d_ents if src is None else d_ents[src].contiguous()  # cluster of every slot of this launch
                if nbytes == 64:
                    if narrow_all is None:
                        allc = torch.arange(nc, dtype=torch.int64, device=dev)
                        narrow_all = torch.empty(nc * 64, dtype=torch.uint8, device=dev)
                        _native.check(L.mpcx_cube_pack_narrow(nc, allc.data_ptr(), recs.data_ptr(), narrow_all.data_ptr(), D.stream_ptr()),
                                      "mpcx_cube_pack_narrow")
                        del allc
                    out = narrow_all
                elif len(kinds) == 1 or not want_narrow:
                    out = recs
                else:
                    # the wide launches of a mostly narrow mesh (row blocks with master rows) cover few slots: their records
                    # are gathered per slot, so that the per-cluster wide records need not be kept
                    out = recs.view(nc, 96)[ridx.long()].contiguous().view(-1)
                    ridx = None
            elif nbytes == 64:
                if src is None:
                    src = torch.arange(nslots, dtype=torch.int64, device=dev)
                out = torch.empty(src.numel() * 64, dtype=torch.uint8, device=dev)
                _native.check(L.mpcx_cube_pack_narrow(src.numel(), src.data_ptr(), recs.data_ptr(), out.data_ptr(), D.stream_ptr()),
                              "mpcx_cube_pack_narrow")
            else:
                out = recs if src is None else recs.view(nslots, 96)[src].contiguous().view(-1)
            slot_cluster = None
            if ordered:  # the cluster of every slot of this launch: forms with coefficients index cube_cells with it
                slot_cluster = d_ents if src is None or src.numel() == nslots else d_ents[src].contiguous()
            parts.append((_native.RowBlockPlanT(nblk, max_rows, max_nnz, 0, d_row0.data_ptr(), off_c.data_ptr(),
                                                D.ptr(slot_cluster), None, None),
                          out, nbytes, ids, off_c, flags, ridx, slot_cluster))
        del recs
        keep = (d_row0, parts, d_verts, d_cells)
        info = {"num_blocks": nb, "num_ents": int(nslots), "max_rows": max_rows, "max_nnz": max_nnz, "clusters": int(nc),
                "narrow_blocks": sum(int(p[0].num_blocks) for p in parts if p[2] == 64),
                "closed_form_blocks": sum(int(p[0].num_blocks) for p in parts if p[5] == 1),
                "bytes": int(d_row0.numel() * 4 + sum(p[4].numel() * 8 + (0 if p[3] is None else p[3].numel() * 4)
                                                      + (0 if p[6] is None else p[6].numel() * 4) for p in parts)
                             + sum(t_.numel() for t_ in {id(p[1]): p[1] for p in parts}.values())),
                "records": "per cluster" if per_cluster else "per slot"}
        return (parts, keep, info)

    try:
        # (the split by cell shape depends on the coordinates -- a moved mesh gets a new plan)
        plan, keep, info = D.cached(A._plans, "cubes", (form, mpc, bc_dev),
                                    (i, max_rows_cfg, max_nnz_cfg, hexa, closed_form_only, ordered, form.mesh.geometry.version), build)
    except _native.PlanNotRepresentable:
        return None
    return plan, keep, info, left


P2CUBE_REC = 640  # bytes per cluster record of the P2 cluster kernel (include/mpcx.h mpcx_p2_cluster_records)


def _p2_cube_plan(A: MPCMatrix, form: Form, i: int, V0, bc_dev, mpc):
    """Plan of the P2 cluster kernel (scalar P2 stiffness on parallelepiped clusters, closed form): the 27 dofs of every
    cluster, one 528-byte record per cluster, and a row-pair plan over the clusters ((cluster, local dof) pairs by row
    block).  Cells of other clusters and cells in no cluster are leftover cells (per-cell kernel).  Same return shape as
    ``_cube_plan``; None when fewer than half of the cells sit in parallelepiped clusters."""
    import torch

    from .clusters import mesh_clusters_device

    integ = form.integrals[i]
    if os.environ.get("MPCX_CLUSTER_DETECT", "topology") == "consecutive":
        return None
    d_verts, left, fan_cells = mesh_clusters_device(form.mesh, integ.num_entities, parallelepipeds_only=True, with_cells=True)
    if d_verts.shape[0] == 0 or d_verts.shape[0] * 6 < 0.5 * integ.num_entities:
        return None

    def build():
        L = _native.lib()
        dev = A.device
        nc = d_verts.shape[0]
        if nc * 27 >= 2 ** 31:
            raise _native.PlanNotRepresentable("P2 cluster plan: cluster * 27 + dof does not fit 32 bits; shard the mesh")
        md = D.mesh_device(form.mesh)
        sd = D.space_device(V0)
        st = D.stream_ptr()
        dofs27 = torch.empty((nc, 27), dtype=torch.int32, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        _native.check(L.mpcx_p2_cluster_dofs(nc, d_verts.data_ptr(), fan_cells.data_ptr(), md["x_dofmap"].data_ptr(),
                                             sd["dofmap"].data_ptr(), dofs27.data_ptr(), flag.data_ptr(), st), "mpcx_p2_cluster_dofs")
        if int(flag.item()) != 0:
            raise _native.PlanNotRepresentable("P2 cluster plan: a cluster dof was not found in the cells' dofmaps")
        hints = None
        if V0.dof_tile_offsets is not None:
            hints = np.ascontiguousarray(V0.dof_tile_offsets.astype(np.int32))
        row0 = _block_ranges(A.shape[0], A.rowptr, ROWBLOCK_MAX_ROWS, ROWBLOCK_MAX_NNZ, 1, hints)
        nb = row0.size - 1
        d_row0, d_off, d_pairs = _block_pairs_device(row0, nc, 1, None, dofs27, 27, 1, dev)
        recs = torch.empty(nc * P2CUBE_REC, dtype=torch.uint8, device=dev)
        _, t = mpc._device()
        _native.check(L.mpcx_p2_cluster_records(nc, d_verts.data_ptr(), dofs27.data_ptr(), md["x"].data_ptr(), D.ptr(bc_dev),
                                                t["is_slave"].data_ptr(),
                                                A.d_rowptr.data_ptr(), A.d_cols.data_ptr(), recs.data_ptr(), flag.data_ptr(), st),
                      "mpcx_p2_cluster_records")
        if int(flag.item()) != 0:
            raise _native.PlanNotRepresentable("P2 cluster plan: a scatter offset does not fit 8 bits (or a column is missing)")
        del dofs27
        max_rows = int(np.diff(row0).max())
        max_nnz = int(np.diff(A.rowptr[row0]).max())
        plan = _native.RowBlockPlanT(nb, max_rows, max_nnz, 1, d_row0.data_ptr(), d_off.data_ptr(), d_pairs.data_ptr(), None, None)
        parts = [(plan, recs, P2CUBE_REC, None, d_off, 1)]
        keep = (d_row0, parts, d_verts, d_pairs)
        info = {"num_blocks": nb, "num_ents": int(d_pairs.numel()), "max_rows": max_rows, "max_nnz": max_nnz, "clusters": int(nc),
                "narrow_blocks": 0, "closed_form_blocks": nb,
                "bytes": int(d_row0.numel() * 4 + recs.numel() + d_off.numel() * 8 + d_pairs.numel() * 4)}
        return (parts, keep, info)

    try:
        plan, keep, info = D.cached(A._plans, "cubes", (form, mpc, bc_dev),
                                    (i, ROWBLOCK_MAX_ROWS, ROWBLOCK_MAX_NNZ, "p2", form.mesh.geometry.version), build)
    except _native.PlanNotRepresentable:
        return None
    return plan, keep, info, left


def _leftover_form(form: Form, i: int, left: np.ndarray) -> Form:
    """integral i restricted to the cells outside any cluster (per-cell kernels)"""
    from .fem import Integral

    integ = form.integrals[i]
    cells = np.ascontiguousarray(left, dtype=np.int32)

    def build():
        # the coefficient travels with the cells (an imported kernel dereferences w[] for every leftover cell as well;
        # the built-in cluster kernels exclude forms with coefficients): Functions are packed per call for the new
        # entity list like for any integral (cpp/assemble_matrix.cpp:583-589)
        co = integ.coefficient
        return Form(form.function_spaces, [Integral("cell", cells, integ.kernel, None if isinstance(co, np.ndarray) else co,
                                                    integ.constant)])

    fl = D.cached(form._device, "leftover", (left,), i, build)
    if isinstance(integ.coefficient, np.ndarray):
        # a caller-packed array [n_entities][cstride]: the rows of the leftover cells, taken on every call (the caller may
        # have rewritten the array in place)
        ents = np.asarray(integ.entities).reshape(integ.num_entities, -1)[:, 0]
        if ents.size == cells.size or np.array_equal(ents[cells], cells):
            rows = cells
        else:
            order = np.argsort(ents, kind="stable")
            rows = order[np.searchsorted(ents[order], cells)]
        fl.integrals[0].coefficient = np.ascontiguousarray(integ.coefficient[rows])  # (compared / uploaded by _device)
    return fl


def _mpc_plan(A: MPCMatrix, form: Form, i: int, mpc0, mpc1, bc0_h, bc1_h, slave_ents_h):
    """Plan of the master contributions of integral i's slave entities (mpcx_mpc_plan_build: the index
    logic of modify_mpc_cell evaluated once, gathered by target position), as device tensors
    (tgt, off, ent, pq, coef) + a has-targets flag; cached on the matrix per (form, integral,
    constraints, Dirichlet markers)."""
    def build():
        L = _native.lib()
        p = _native._ptr
        integ = form.integrals[i]
        V0, V1 = form.function_spaces
        ents = np.ascontiguousarray(integ.entities.astype(np.int32).reshape(-1))
        c0, c1 = mpc0.coefficients()[0], mpc1.coefficients()[0]
        h = L.mpcx_mpc_plan_build(
            slave_ents_h.size, p(slave_ents_h), integ.estride, p(ents), p(ents), p(V0.dofmap.list), V0.element_ndofs,
            V0.dofmap.bs, p(V1.dofmap.list), V1.element_ndofs, V1.dofmap.bs,
            None if bc0_h is None else p(bc0_h), None if bc1_h is None else p(bc1_h),
            p(mpc0.is_slave), p(mpc0.masters.offsets), p(mpc0.masters.array), p(c0),
            p(mpc1.is_slave), p(mpc1.masters.offsets), p(mpc1.masters.array), p(c1), p(A.rowptr), p(A.cols))
        if not h:
            raise RuntimeError("mpcx_mpc_plan_build failed: " + L.mpcx_last_error().decode())
        try:
            n, nt = L.mpcx_mpc_plan_size(h), L.mpcx_mpc_plan_num_targets(h)
            tgt = np.zeros(max(nt, 1), dtype=np.int64)
            off = np.zeros(max(nt, 1) + 1, dtype=np.int64)
            ent = np.zeros(max(n, 1), dtype=np.int32)
            pq = np.zeros(max(n, 1), dtype=np.int32)
            coef = np.zeros(max(n, 1), dtype=np.float64)
            L.mpcx_mpc_plan_copy(h, p(tgt), p(off), p(ent), p(pq), p(coef))
        finally:
            L.mpcx_mpc_plan_free(h)
        dev = A.device
        return tuple(D._to_dev(t, dev) for t in (tgt, off, ent, pq, coef)) + (nt > 0,)

    # bc0_h / bc1_h are the marker arrays cached per (space, bcs): their identity stands for the bc set
    return D.cached(A._plans, "mpc_plan", (form, mpc0, mpc1, bc0_h, bc1_h), i, build)


# tuples above which the device plan is not built (its sort / gathers run through torch, whose indexing is not
# trusted beyond 2^31 bytes on this stack): such layers go through matrix_mpc_kernel
MPC_PLAN_MAX_TUPLES = int(float(os.environ.get("MPCX_MPC_PLAN_MAX_TUPLES", 2.5e8)))


# scratch for the element tensors of the slave entities of an imported kernel (one per matrix; above this the tuples
# re-tabulate their entity)
SLAVE_TENSOR_BYTES = int(float(os.environ.get("MPCX_SLAVE_TENSOR_BYTES", 4e9)))


def _empty_f64(n: int, dev):
    import torch

    return torch.empty(max(int(n), 1), dtype=torch.float64, device=dev)


def _mpc_plan_device(A: MPCMatrix, form: Form, i: int, mpc0, mpc1, bc0_dev, bc1_dev, slave_ents_dev):
    """The same plan built by the HIP kernel ``mpc_plan_device_kernel`` (include/mpcx.h mpcx_mpc_plan_device):
    count -> scan -> fill, then a stable sort by target position and a run-length pass (torch: plumbing).
    Nothing visits the host: 0.02 s instead of 0.5 s at config 2, 1 s instead of minutes for the slip walls of
    config 3.  Returns the tuple (tgt, off, ent, pq, coef, has_targets) or None if there are too many tuples."""
    import torch

    def build():
        L = _native.lib()
        integ = form.integrals[i]
        V0, V1 = form.function_spaces
        s0, s1 = D.space_device(V0), D.space_device(V1)
        idv = D.integral_device(form, i)
        m0, _k0 = mpc0._device()
        m1, _k1 = mpc1._device()
        dev = A.device
        n = slave_ents_dev.numel()
        k = integ.kernel
        diag = int(k.form in (0, 1, 4) and V0.dofmap.bs > 1)  # stiffness / mass / facet mass on blocked spaces
        ents = idv["entities_ptr"]
        st = D.stream_ptr()
        counts = torch.empty(n, dtype=torch.int64, device=dev)

        def call(offsets, pos, ent, pq, coef):
            rc = L.mpcx_mpc_plan_device(n, slave_ents_dev.data_ptr(), integ.estride, ents, ents, s0["dofmap"].data_ptr(),
                                        V0.element_ndofs, V0.dofmap.bs, s1["dofmap"].data_ptr(), V1.element_ndofs,
                                        V1.dofmap.bs, D.ptr(bc0_dev), D.ptr(bc1_dev), C.byref(m0), C.byref(m1),
                                        A.d_rowptr.data_ptr(), A.d_cols.data_ptr(), diag, counts.data_ptr(),
                                        D.ptr(offsets), D.ptr(pos), D.ptr(ent), D.ptr(pq), D.ptr(coef), st)
            _native.check(rc, "mpcx_mpc_plan_device")

        from . import _prims

        call(None, None, None, None, None)
        scan = _prims.scan_i64(counts)  # (rocPRIM behind the C ABI, like the sort and the run-length step below)
        total = int(scan[-1].item())
        if total > MPC_PLAN_MAX_TUPLES:
            return None
        z64 = torch.zeros(1, dtype=torch.int64, device=dev)
        if total == 0:
            return (z64, torch.zeros(2, dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.int32, device=dev),
                    torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.float64, device=dev), False)
        offsets = scan[:-1]
        pos = torch.empty(total, dtype=torch.int64, device=dev)
        ent = torch.empty(total, dtype=torch.int32, device=dev)
        pq = torch.empty(total, dtype=torch.int32, device=dev)
        coef = torch.empty(total, dtype=torch.float64, device=dev)
        call(offsets, pos, ent, pq, coef)
        # stable sort by target position (signed keys: the -1 of tuples outside the pattern come first)
        pos, order = _prims.sort_pairs(pos, torch.arange(total, dtype=torch.int64, device=dev), 64)
        nneg = int((pos < 0).sum().item())  # tuples outside the pattern (none for a pattern built from the same constraint)
        pos, order = pos[nneg:].contiguous(), order[nneg:]
        if pos.numel() == 0:
            return (z64, torch.zeros(2, dtype=torch.int64, device=dev), ent[:1], pq[:1], coef[:1], False)
        tgt, off = _prims.runs(pos)
        return (tgt, off, ent[order].contiguous(), pq[order].contiguous(), coef[order].contiguous(), True)

    return D.cached(A._plans, "mpc_plan_dev", (form, mpc0, mpc1, bc0_dev, bc1_dev), i, build)


def _masked_dofmap(form: Form, V, bc_dev, mpc, which: int, rotate: bool = False):
    """dofmap with the Dirichlet/slave mask folded into bits 28.. (device, cached
    per (space, bcs, constraint)): replaces the marker gathers of
    cpp/assemble_matrix.cpp:511-533 and the is_slave look-ups in the bulk kernel."""
    import torch

    def build():
        if V.num_dofs // V.dofmap.bs >= (1 << 28):
            raise _native.PlanNotRepresentable("row-block algorithm: more than 2^28 dof blocks per GPU; shard the mesh")
        sd = D.space_device(V)
        _, t = mpc._device()
        out = torch.empty_like(sd["dofmap"])
        rc = _native.lib().mpcx_mask_dofmap(sd["dofmap"].data_ptr(), sd["dofmap"].shape[0], sd["dofmap"].shape[1],
                                            V.dofmap.bs, D.ptr(bc_dev), t["is_slave"].data_ptr(), int(rotate),
                                            out.data_ptr(), D.stream_ptr())
        _native.check(rc, "mpcx_mask_dofmap")
        return out

    # bc_dev is the marker tensor cached per (space, bcs): its identity stands for the bc set
    return D.cached(form._device, "mdof", (V, mpc, bc_dev), (which, rotate), build, maxsize=4)


def matrix_args(form: Form, i: int, A: MPCMatrix, mpc0, mpc1, bcs, alg: int, store_mode: int = 0,
                with_mpc_kernel: bool = True, allow_cubes: bool = True, allow_block_scalar: bool = False):
    """Fill the C-ABI argument block of ``mpcx_assemble_matrix`` for integral i.  ``allow_block_scalar``: the
    node-block kernel may leave its result in block-scalar storage (one value per bs x bs block, la.MPCMatrix)."""
    V0, V1 = form.function_spaces
    integ = form.integrals[i]
    md = D.mesh_device(form.mesh)
    s0, s1 = D.space_device(V0), D.space_device(V1)
    bc0_h, bc0 = D.bc_markers(V0, bcs, form._device)
    bc1_h, bc1 = D.bc_markers(V1, bcs, form._device)
    m0, k0 = mpc0._device()
    m1, k1 = mpc1._device()
    idv = D.integral_device(form, i)
    slave_ents_h, slave_ents = _slave_entities(form, i, mpc0, mpc1)
    a = _native.MatrixArgs()
    a.nrows = A.shape[0]
    a.rowptr, a.cols = A.d_rowptr.data_ptr(), A.d_cols.data_ptr()  # (a.vals: at the end, unless block-scalar)
    a.kernel = idv["kernel"]
    a.x, a.x_dofmap, a.nv = md["x"].data_ptr(), md["x_dofmap"].data_ptr(), form.mesh.geometry.dofmap.shape[1]
    a.estride, a.n_entities = integ.estride, integ.num_entities
    a.entities = a.entities0 = a.entities1 = idv["entities_ptr"]
    a.coeffs = D.ptr(idv["coeffs"])
    a.cstride = integ.cstride
    a.constants = D.ptr(idv["constants"])
    a.dofmap0, a.nd0, a.bs0 = s0["dofmap"].data_ptr(), V0.element_ndofs, V0.dofmap.bs
    a.dofmap1, a.nd1, a.bs1 = s1["dofmap"].data_ptr(), V1.element_ndofs, V1.dofmap.bs
    a.bc0, a.bc1 = D.ptr(bc0), D.ptr(bc1)
    a.mpc0, a.mpc1 = m0, m1
    a.slave_entities = slave_ents.data_ptr()
    a.n_slave_entities = slave_ents.numel() if with_mpc_kernel else 0
    a.cell_info0, a.cell_info1 = D.cell_info_ptr(V0, integ.kernel), D.cell_info_ptr(V1, integ.kernel)
    mplan = None
    # master contributions from a plan gathered by target position (no device atomics, no CSR searches in the
    # timed path).  MPCX_MPC_PLAN = device (default: built by a HIP kernel) | host (mpcx_mpc_plan_build; only for
    # thin layers: a large layer of big elements would take the host minutes) | none (matrix_mpc_kernel: what a
    # caller of the bare C ABI gets with mpc_plan_off == NULL); MPCX_NO_MPC_PLAN=1 is the old spelling of none
    mode = "none" if os.environ.get("MPCX_NO_MPC_PLAN") else os.environ.get("MPCX_MPC_PLAN", "device").lower()
    if _native.scalar_id(getattr(form, "dtype", np.float64)) != 0:
        mode = "none"  # (the plan carries fp64 coefficients; the scalar-type kernels eliminate inside the entity loop)
    if integ.kernel.form == 100 and _native.lib().mpcx_ufcx_big_tensor(idv["kernel"].ufcx):
        # an imported kernel whose element tensor does not fit a thread's private memory (vector-valued Q3 hexahedra:
        # 192 x 192): per-entity kernels with the tensor in a global scratch slab only (include/mpcx.h mpcx_ufcx_big_tensor)
        if alg != 1:
            raise _native.PlanNotRepresentable("imported kernel with an element tensor of more than 12288 entries: "
                                               "algorithm='atomic' (or 'auto')")
        mode = "none"
    n0n1 = V0.element_ndofs * V0.dofmap.bs * V1.element_ndofs * V1.dofmap.bs
    if a.n_slave_entities > 0 and mode == "device" and max(V0.element_ndofs * V0.dofmap.bs,
                                                           V1.element_ndofs * V1.dofmap.bs) <= 32:
        mplan = _mpc_plan_device(A, form, i, mpc0, mpc1, bc0, bc1, slave_ents)
    elif a.n_slave_entities > 0 and mode == "host" and a.n_slave_entities * n0n1 <= MPC_PLAN_MAX_ENTRIES:
        mplan = _mpc_plan(A, form, i, mpc0, mpc1, bc0_h, bc1_h, slave_ents_h)
    if mplan is not None:
        a.mpc_plan_targets = mplan[0].numel() if mplan[5] else 0
        (a.mpc_plan_tgt, a.mpc_plan_off, a.mpc_plan_ent, a.mpc_plan_pq,
         a.mpc_plan_coef) = (t.data_ptr() for t in mplan[:5])
        mean = mplan[2].numel() / max(mplan[0].numel(), 1)  # tuples per target position
        a.mpc_plan_group = 16 if mean > 10 else (4 if mean > 2.5 else 1)
        if integ.kernel.form == 100 and a.mpc_plan_targets > 0 and os.environ.get("MPCX_SLAVE_TENSORS", "1") != "0" \
                and a.n_slave_entities * n0n1 * 8 <= SLAVE_TENSOR_BYTES:
            # imported kernel: every slave entity's tensor once per call into a scratch array, the plan's tuples read
            # their entry from it (include/mpcx.h mpcx_matrix_args_t::slave_tensors)
            def slots():
                import torch

                return torch.searchsorted(slave_ents.to(torch.int64), mplan[2].to(torch.int64)).to(torch.int32).contiguous()

            slot = D.cached(A._plans, "mpc_plan_slot", (mplan[2], slave_ents), i, slots)
            scratch = D.cached(A._plans, "slave_tensors", (), (a.n_slave_entities, n0n1),
                               lambda: _empty_f64(a.n_slave_entities * n0n1, A.device), maxsize=2)
            a.mpc_plan_slot, a.slave_tensors = slot.data_ptr(), scratch.data_ptr()
            mplan = tuple(mplan) + (slot, scratch)
    a.algorithm = alg
    a.store_mode = store_mode
    a.stream = D.stream_ptr()
    keep = [md, s0, s1, bc0, bc1, k0, k1, idv, slave_ents, mplan]
    a.leftover = None  # (python attribute) cells the cluster kernel does not cover
    a.block_scalar = False  # (python attribute) the result goes to A._compact, not to A.vals
    if alg == 2:
        from . import dispatch

        same = V1 is V0 and mpc1 is mpc0 and bc1 is bc0
        kf = integ.kernel
        ctx = dispatch.Ctx(form=kf.form, tet=kf.celltype == 2, d0=V0.degree, bs0=V0.dofmap.bs, d1=V1.degree, bs1=V1.dofmap.bs,
                           nd0=V0.element_ndofs, nd1=V1.element_ndofs,
                           nq=int(kf.qwts.size if integ.itype == "cell" else kf.fqwts.size), cell_integral=integ.itype == "cell",
                           has_coefficient=integ.coefficient is not None, coeff_degree=kf.coeff_degree,
                           all_cells=idv["entities_ptr"] is None and integ.estride == 1,
                           p1_geometry=s0["dofmap"] is md["x_dofmap"], same=same, tiled=V0.dof_tile_offsets is not None,
                           builtin_form=(kf.builtin.form if getattr(kf, "builtin", None) is not None else -1),
                           has_transforms=getattr(kf, "ufcx_transforms", None) is not None)
        a.kernel_name = None  # (python attribute) the table entry that was taken
        names = dispatch.candidates(dispatch.MATRIX, ctx, "matrix")
        if _native.scalar_id(getattr(form, "dtype", np.float64)) != 0:
            if kf.form == 100:
                raise NotImplementedError("imported (UFCx) kernels are fp64-real")
            names = ["rowblock"]  # the one formulation csrc/mpcx_scalar.hip restates over a scalar type
        for name in names:
            lean = pairs = False
            smask = None
            if name == "hex_cube":
                # hexahedra: the bulk through the built-in Q1 kernel over (row block, cell) slots (MPCX_ALG_CUBE with the
                # built-in twin of the imported kernel), then the master contributions of the slave cells through the
                # imported kernel -- a second call without bulk entities
                cp = _cube_plan(A, form, i, V0, bc0, mpc0, hexa=True) if allow_cubes else None
                if cp is None:
                    continue
                parts, ck, _info, _left = cp
                t = _native.MatrixArgs.from_buffer_copy(a)  # the imported kernel's call: master contributions only
                t.algorithm, t.n_entities, t.store_mode = 2, 0, 0
                _set_vals(t, A)
                t.leftover, t.kernel_name, t.block_scalar, t.second = None, name, False, None
                a.kernel = idv["kernel_builtin"]
                a.algorithm = 3
                a.n_slave_entities = 0
                a.kernel_name = name
                _set_vals(a, A)
                # one launch per kind of row block (all cells parallelepipeds / not), then the master contributions: a chain
                # of follow-up calls (python attribute ``second``)
                chain = []
                for n_part, part in enumerate(parts):
                    plan, recs, nbytes, ids, _off = part[:5]
                    u = a if n_part == 0 else _native.MatrixArgs.from_buffer_copy(a)
                    u.plan = plan
                    u.cube_recs, u.cube_rec_bytes, u.cube_block_ids = recs.data_ptr(), nbytes, D.ptr(ids)
                    u.cube_flags = part[5] if len(part) > 5 else 0
                    u.cube_rec_index = D.ptr(part[6]) if len(part) > 6 else None
                    if n_part > 0:
                        u.leftover, u.kernel_name, u.block_scalar = None, name, False
                    chain.append(u)
                if t.n_slave_entities > 0:
                    chain.append(t)
                for u, v in zip(chain, chain[1:] + [None]):
                    u.second = v
                A._compact_stale = False
                keep += [ck]
                return a, keep
            if name in ("cube", "cube_el", "p2_cube", "ufcx_cube"):
                # cell clusters (MPCX_ALG_CUBE); None when the mesh has no clean six-tet fans or an offset overflows
                if not allow_cubes:
                    cp = None
                elif name == "p2_cube":
                    cp = _p2_cube_plan(A, form, i, V0, bc0, mpc0)
                else:
                    cp = _cube_plan(A, form, i, V0, bc0, mpc0, closed_form_only=(name == "cube_el"), ordered=(name == "ufcx_cube"))
                if cp is None:
                    continue
                parts, ck, _info, left = cp
                if name == "ufcx_cube" and integ.cstride > 0:
                    a.cube_cells = ck[3].data_ptr()  # the cell of every cluster tet: index of its packed coefficients
                a.algorithm = 3
                a.leftover = left if left.size else None
                a.kernel_name = name
                _set_vals(a, A)
                # one launch per kind of row block (record format, cell shape): a chain of follow-up calls (python attribute
                # ``second``).  The master contributions ride on the LAST launch: the row blocks are written in store mode, so
                # they must all be in place before anything is added to them
                chain = []
                n_slave = a.n_slave_entities
                for n_part, part in enumerate(parts):
                    plan, recs, nbytes, ids, _off = part[:5]
                    t = a if n_part == 0 else _native.MatrixArgs.from_buffer_copy(a)
                    t.plan = plan
                    t.cube_recs, t.cube_rec_bytes, t.cube_block_ids = recs.data_ptr(), nbytes, D.ptr(ids)
                    t.cube_flags = part[5] if len(part) > 5 else 0
                    t.cube_rec_index = D.ptr(part[6]) if len(part) > 6 else None
                    t.n_slave_entities = n_slave if n_part == len(parts) - 1 else 0
                    if n_part > 0:
                        t.leftover, t.kernel_name, t.block_scalar = None, name, False
                    chain.append(t)
                for u, v in zip(chain, chain[1:] + [None]):
                    u.second = v
                keep += [ck]
                return a, keep
            if name == "ufcx_pairs":
                # imported text with row-wise copies (csrc/mpcx_ufcx.cpp): pair records, no cached context
                if not _native.lib().mpcx_ufcx_rowwise(idv["kernel"].ufcx) or os.environ.get("MPCX_UFCX_PAIRS", "1") == "0":
                    continue
                try:
                    plan, pk, _info = _pairs_plan(A, form, i, V0, V1, bc0, bc1, mpc0, mpc1)
                except _native.PlanNotRepresentable:
                    continue
                if pk[3] is not None:  # (dictionary-compressed records, MPCX_PAIRS_DICT=1: the imported-text kernel reads full ones)
                    continue
                md1 = _masked_dofmap(form, V1, bc1, mpc1, 1)
                a.plan = plan
                a.pair_recs, a.pair_ctx, a.pair_dict = pk[2].data_ptr(), None, None
                a.mdofmap1 = md1.data_ptr()
                a.kernel_name = name
                keep += [pk, md1]
                break
            if name == "pairs":
                # pair records + cached contexts (csrc/mpcx_pairs.hip): nothing else is read per entity
                try:
                    plan, pk, _info = _pairs_plan(A, form, i, V0, V1, bc0, bc1, mpc0, mpc1)
                    pctx = _pair_context(form, i)
                except _native.PlanNotRepresentable:
                    continue
                md1 = _masked_dofmap(form, V1, bc1, mpc1, 1)  # column masks of the few entities that have any
                a.plan = plan
                a.pair_recs, a.pair_ctx, a.pair_dict = pk[2].data_ptr(), D.ptr(pctx), D.ptr(pk[3])
                a.mdofmap1 = md1.data_ptr()
                a.kernel_name = name
                keep += [pk, pctx, md1]
                break
            if name == "rowpair":
                pairs = True  # (reads the plain, unrotated masked dofmaps)
            elif name == "nodeblock":
                smask = _slot_mask(A, form, V0, V1, bc0, bc1, mpc0, mpc1)
                if smask is None:  # the pattern is not made of whole bs x bs blocks
                    continue
                a.slot_mask = smask.data_ptr()
                keep += [smask]
                if (allow_block_scalar and os.environ.get("MPCX_BLOCK_SCALAR", "1") != "0"
                        and (a.n_slave_entities == 0 or (mplan is not None and mplan[5]))):
                    # (a plan without a target inside the pattern has no overlay to add into: scalar CSR values then)
                    # block-scalar storage: 8 B per bs x bs block instead of 8 bs^2; couplings and diagonals in an overlay
                    import torch

                    bs = V0.dofmap.bs
                    c = A._compact
                    if c is None or c["bs"] != bs or c["mask"] is not smask:
                        c = dict(bs=bs, svals=torch.empty(A.nnz // (bs * bs), dtype=torch.float64, device=A.device), mask=smask,
                                 ov_pos=None, ov_val=None, ov_plan=None, diag_pos=None, diagval=0.0, diag_key=None, ov_rc=None)
                        A._compact = c
                    if mplan is not None and mplan[5] and c["ov_plan"] is not mplan:
                        c["ov_pos"], c["ov_plan"], c["ov_rc"] = mplan[0], mplan, None
                        c["ov_val"] = torch.zeros(mplan[0].numel(), dtype=torch.float64, device=A.device)
                    elif mplan is None or not mplan[5]:
                        c["ov_pos"] = c["ov_val"] = c["ov_plan"] = c["ov_rc"] = None
                    a.block_vals = c["svals"].data_ptr()
                    a.mpc_plan_out = D.ptr(c["ov_val"])
                    a.block_scalar = True
            elif name == "rowblock_lean":
                lean = True
            plan, pk, _info = _rowblock_plan(A, form, i, V0, lean, pairs, smask is not None,
                                             csr_valued=smask is not None and not a.block_scalar)
            a.plan = plan
            a.lean = int(lean)
            md0 = _masked_dofmap(form, V0, bc0, mpc0, 0, lean)
            md1 = md0 if same else _masked_dofmap(form, V1, bc1, mpc1, 1)
            a.mdofmap0, a.mdofmap1 = md0.data_ptr(), md1.data_ptr()
            a.kernel_name = name
            keep += [pk, md0, md1]
            break
    if not a.block_scalar:
        A._compact_stale = False  # every value is (re)written below: nothing to expand first
        _set_vals(a, A)
    return a, keep


def _set_vals(a, A: MPCMatrix):
    """the value array of a launch: A's own, or -- for the locality twin of a caller's matrix (``A._write_through``, set by
    dolfinx_mpc_amd/locality.py) -- the CALLER's array through the permutation of the two CSRs (mpcx_matrix_args_t::val_map)"""
    wt = getattr(A, "_write_through", None)
    if wt is None:
        a.vals = A.vals.data_ptr()
    else:
        target, vmap, wide, omap, odelta = wt
        a.vals = target.vals.data_ptr()
        a.val_map, a.val_map_wide = vmap.data_ptr(), int(wide)
        if odelta is not None:
            a.out_map, a.out_delta = omap.data_ptr(), odelta.data_ptr()


@timed("~MPC: Assemble matrix (C++)")
def assemble_matrix(
    form: Form,
    constraint: Union[MultiPointConstraint, Sequence[MultiPointConstraint]],
    bcs: Optional[Sequence[DirichletBC]] = None,
    diagval: float = 1,
    A: Optional[MPCMatrix] = None,
    num_threads: Optional[int] = 1,
    algorithm: Optional[str] = None,
) -> MPCMatrix:
    """Assemble a bilinear form into a CSR matrix with multi point constraints
    and Dirichlet conditions (python/src/dolfinx_mpc/assemble_matrix.py:21-65).

    Args:
        form: the bilinear form
        constraint: the multi point constraint (or a (row, col) pair)
        bcs: Dirichlet boundary conditions
        diagval: value set on the diagonal for slave and Dirichlet dofs
        A: matrix to assemble into (created with the MPC pattern if None)
        num_threads: accepted for API compatibility (host set-up threads)
        algorithm: "atomic" | "rowblock" | None (= env MPCX_MATRIX_ALG or "auto")
    """
    import torch

    bcs = [] if bcs is None else list(bcs)
    if isinstance(constraint, MultiPointConstraint):
        assert form.function_spaces[0] is form.function_spaces[1]
    mpc0, mpc1 = _pair(constraint)
    if form.rank != 2:
        raise RuntimeError("assemble_matrix needs a bilinear form")
    _native.require_gpu()
    D.resolve_builtin_twins(form)  # imported kernels with a stated (and checked) built-in twin, fem.form_ufcx(builtin=...)
    if A is None:
        A = create_matrix(form, mpc0, mpc1)
    alg = _ALG[(algorithm or os.environ.get("MPCX_MATRIX_ALG", "auto")).lower()]
    sid = _native.scalar_id(form.dtype)
    if sid != 0:
        # float32 / complex64 / complex128: the general kernels of csrc/mpcx_scalar.hip -- LDS row blocks over a plain
        # entity plan ("rowblock", and "auto" when the plan can be built) or per-entity device atomics ("atomic"); the
        # cluster, pair, node-block and lean kernels are fp64-real
        for m in (mpc0, mpc1):
            if np.dtype(m.dtype) != form.dtype:
                raise ValueError(f"form of scalar type {form.dtype} assembled with a constraint of {np.dtype(m.dtype)}")
        if A.dtype != _torch_dtype_of(form.dtype):
            raise ValueError("matrix and form of different scalar types")
    for integ in form.integrals:
        if integ.itype not in ("cell", "exterior_facet"):
            raise RuntimeError("Not implemented yet")  # cpp/assemble_matrix.cpp:658-659
    if alg != 1 and sid == 0:
        # a numbering without locality: assemble on the spatially reordered twin, hand the values back in the caller's
        # numbering (dolfinx_mpc_amd/locality.py)
        from . import locality

        tw = locality.twin_of(form.mesh)
        if tw is not None:
            try:
                return locality.assemble_matrix(tw, form, mpc0, mpc1, bcs, diagval, A, alg)
            except _native.PlanNotRepresentable:
                tw.remember_failure(A, form)  # (pattern or plan of the twin not representable: not retried on every call)
    D.mesh_device(form.mesh)  # a moved mesh is refreshed on the caller's stream, before any side stream reads it
    from .la import side_stream

    A._twin_stale = False
    with side_stream("matrix", A):  # the library's matrix stream (la.side_stream); completion is awaited by A.vals
        _assemble_matrix_on_stream(form, mpc0, mpc1, bcs, diagval, A, alg)
    return A


def _assemble_matrix_on_stream(form: Form, mpc0, mpc1, bcs, diagval, A: MPCMatrix, alg: int):
    """the body of ``assemble_matrix``: everything is enqueued on the CURRENT torch stream"""
    import torch  # noqa: F401

    L = _native.lib()
    auto = alg == 0
    if auto:
        alg = 2  # LDS row blocks (clusters where they apply); device atomics if a plan cannot be built

    V0, V1 = form.function_spaces
    stream = D.stream_ptr()
    wt = getattr(A, "_write_through", None)
    target = A if wt is None else wt[0]  # whose value array the launches write (``_set_vals``)

    def prepare(alg):
        """argument blocks of every integral (plans are built / fetched here, nothing is launched)"""
        calls, zeroed = [], False
        for i, integ in enumerate(form.integrals):
            if alg == 2 and (zeroed or integ.num_entities > 0):
                # the first non-empty integral's row blocks overwrite every value: no memset pass (an empty
                # integral launches nothing, so it must not count as having cleared a reused matrix)
                store_mode = 0 if zeroed else 1
                zeroed = True
                memset = False
            else:
                store_mode = 0
                memset = not zeroed  # python/src/dolfinx_mpc/assemble_matrix.py:51
                zeroed = True
            a, keep = matrix_args(form, i, A, mpc0, mpc1, bcs, alg, store_mode,
                                  allow_block_scalar=(len(form.integrals) == 1 and integ.num_entities > 0
                                                      and A._exchange is None and alg == 2))
            calls.append((memset, a, keep))
            nxt = getattr(a, "second", None)  # cluster path: the row blocks of the other record format / cell shape, ...
            while nxt is not None:
                calls.append((False, nxt, keep))
                nxt = getattr(nxt, "second", None)
            if a.leftover is not None:
                # cells outside any cluster: per-cell row-block kernel, ADDed; their master contributions are part
                # of the call above (its plan covers every slave entity of the integral)
                fl = _leftover_form(form, i, a.leftover)
                al, kl = matrix_args(fl, 0, A, mpc0, mpc1, bcs, alg, 0, with_mpc_kernel=False, allow_cubes=False)
                calls.append((False, al, kl))
        return calls, zeroed

    try:
        calls, zeroed = prepare(alg)
    except _native.PlanNotRepresentable:
        if not auto:
            raise
        # rows with more than 255 column blocks before an entity's column, tiny LDS ...: thread-per-entity atomics
        alg = 1
        calls, zeroed = prepare(alg)
    block_scalar = len(calls) == 1 and calls[0][1].block_scalar
    from . import corun

    if corun.note_matrix_call(A) and alg == 2:
        # a vector assembly runs beside this call (time loop, benchmark step): the first part of every row-block launch is
        # capped so that the vector kernel is co-resident on every CU, the rest runs uncapped (dolfinx_mpc_amd/corun.py)
        calls = corun.split_calls(calls, corun.params())
    for memset, a, _keep in calls:
        if memset:
            target.zeroEntries()
        if a.block_scalar and A._compact["ov_val"] is not None:
            A._compact["ov_val"].zero_()
        _native.check(L.mpcx_assemble_matrix(C.byref(a)), "mpcx_assemble_matrix")
    if not zeroed:
        target.zeroEntries()
    if block_scalar:
        _block_scalar_diagonals(A, form, mpc0, mpc1, bcs, diagval)
        A._compact_stale = True
        A.assemble()
        return

    # slave diagonal, cpp/assemble_matrix.cpp:711-724 (only when mpc0.V == mpc1.V)
    if mpc0.function_space is mpc1.function_space:
        _, t = mpc0._device()
        ns = mpc0.num_local_slaves
        _add_diagonal(L, A, t["slaves"].data_ptr(), ns, diagval, stream, form)
    # Dirichlet diagonal: dolfinx insert_diagonal, python/src/dolfinx_mpc/assemble_matrix.py:59-62
    if form.function_spaces[0] is form.function_spaces[1]:
        for bc in bcs:
            if not V0.contains(bc.function_space):
                continue
            def owned_dofs(bc=bc):
                dofs_h, nowned = bc.dof_indices()  # owned dofs only, like dolfinx insert_diagonal
                return D._to_dev(dofs_h[:nowned], A.device)

            dofs = D.cached(form._device, "bcdofs", (bc,), str(A.device), owned_dofs, maxsize=16)
            _add_diagonal(L, A, dofs.data_ptr(), dofs.numel(), diagval, stream, form)
    A.assemble()


def _add_diagonal(L, A: MPCMatrix, dofs_ptr, n: int, diagval, stream, form: Form):
    """A[d, d] += diagval for the listed dofs, in the matrix's scalar type"""
    sid = _native.scalar_id(getattr(form, "dtype", np.float64))
    wt = getattr(A, "_write_through", None)
    if wt is not None:
        target, vmap, wide = wt[:3]
        _native.check(L.mpcx_add_diagonal_mapped(A.shape[0], A.d_rowptr.data_ptr(), A.d_cols.data_ptr(), target.vals.data_ptr(),
                                                 dofs_ptr, n, float(diagval), vmap.data_ptr(), int(wide), stream),
                      "mpcx_add_diagonal_mapped")
    elif sid == 0:
        _native.check(L.mpcx_add_diagonal(A.shape[0], A.d_rowptr.data_ptr(), A.d_cols.data_ptr(), A.vals.data_ptr(), dofs_ptr, n,
                                          float(diagval), stream), "mpcx_add_diagonal")
    else:
        dv = complex(diagval)
        _native.check(L.mpcx_add_diagonal_scalar(sid, A.d_rowptr.data_ptr(), A.d_cols.data_ptr(), A.vals.data_ptr(), dofs_ptr, n,
                                                 dv.real, dv.imag, stream), "mpcx_add_diagonal_scalar")


def _block_scalar_diagonals(A: MPCMatrix, form: Form, mpc0, mpc1, bcs, diagval):
    """slave diagonal (cpp/assemble_matrix.cpp:711-724) and Dirichlet diagonal (insert_diagonal,
    python/src/dolfinx_mpc/assemble_matrix.py:59-62) of a matrix kept in block-scalar storage: their positions in the
    CSR, found once per (constraint, bcs), go to the overlay"""
    import torch

    c = A._compact
    V0 = form.function_spaces[0]
    key = (mpc0, mpc1, tuple(bcs))
    if c["diag_key"] is None or len(c["diag_key"]) != 3 or c["diag_key"][0] is not mpc0 or c["diag_key"][1] is not mpc1 \
            or len(c["diag_key"][2]) != len(bcs) or any(x is not y for x, y in zip(c["diag_key"][2], bcs)):
        dofs = []
        if mpc0.function_space is mpc1.function_space:
            dofs.append(mpc0.device_tensors()["slaves"][: mpc0.num_local_slaves].to(torch.int32))
        if form.function_spaces[0] is form.function_spaces[1]:
            for bc in bcs:
                if V0.contains(bc.function_space):
                    d, nowned = bc.dof_indices()
                    dofs.append(D._to_dev(np.ascontiguousarray(d[:nowned], dtype=np.int32), A.device))
        if dofs:
            d = torch.cat(dofs).contiguous()  # (a dof listed twice gets the diagonal twice, like the full path)
            pos = torch.empty(d.numel(), dtype=torch.int64, device=A.device)
            _native.check(_native.lib().mpcx_csr_positions(A.d_rowptr.data_ptr(), A.d_cols.data_ptr(), d.data_ptr(), d.data_ptr(),
                                                           d.numel(), pos.data_ptr(), D.stream_ptr()), "mpcx_csr_positions")
            c["diag_pos"], c["diag_dofs"] = pos[pos >= 0].contiguous(), d[pos >= 0].contiguous()
        else:
            c["diag_pos"] = c["diag_dofs"] = None
        c["diag_key"] = key
    c["diagval"] = float(diagval)


def create_matrix_nest(a: Sequence[Sequence[Optional[Form]]], constraints: Sequence[MultiPointConstraint]):
    """python/src/dolfinx_mpc/assemble_matrix.py:91-117: block (i, j) uses
    (constraints[i], constraints[j]); ``None`` blocks are skipped."""
    assert len(constraints) == len(a)
    return [[None if blk is None else create_matrix(blk, constraints[i], constraints[j]) for j, blk in enumerate(row)]
            for i, row in enumerate(a)]


def assemble_matrix_nest(A, a, constraints, bcs: Sequence[DirichletBC] = (), diagval: float = 1,
                         num_threads: Optional[int] = 1):
    """python/src/dolfinx_mpc/assemble_matrix.py:120-146"""
    for i, a_row in enumerate(a):
        for j, a_block in enumerate(a_row):
            if a_block is not None:
                assemble_matrix(a_block, (constraints[i], constraints[j]), bcs=bcs, diagval=diagval, A=A[i][j],
                                num_threads=num_threads)
