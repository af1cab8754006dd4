"""Device-resident linear algebra containers (stand-ins for the PETSc ``Mat`` /
``Vec`` the reference assembles into, python/src/dolfinx_mpc/mpc.cpp:273-297).

Partitioned meshes (dolfinx_mpc_amd.distributed): the objects carry the interface exchange of their
space, so that the reference's own calls do the reduction -- ``A.assemble()``
(python/src/dolfinx_mpc/assemble_matrix.py:64) and ``b.ghostUpdate(addv=ADD, mode=REVERSE)``
(python/benchmarks/bench_periodic.py:108) -- and a reference-style driver runs unchanged on N GPUs."""

from __future__ import annotations

import contextlib
import enum
import os

import numpy as np

from . import _native


class InsertMode(enum.IntEnum):
    """petsc4py.PETSc.InsertMode values used by the reference's drivers"""
    INSERT = 1
    ADD = 2
    INSERT_VALUES = 1
    ADD_VALUES = 2


class ScatterMode(enum.IntEnum):
    """petsc4py.PETSc.ScatterMode"""
    FORWARD = 0
    REVERSE = 1


# ---------------------------------------------------------------------------------------------------------
# Side streams.  assemble_matrix and assemble_vector enqueue their kernels on two library-owned HIP streams instead of
# the caller's: the matrix kernels are HBM / latency-bound, the vector kernels VALU-bound, and back-to-back calls of a
# reference-style driver (bench_periodic.py:97-103) then overlap tail against head (config 2: 3.87 ms for both kernels
# instead of 4.18, profiles/r03_overlap_probe.txt).  Ordering is kept with events: a side stream first waits for
# everything the caller's stream holds at call time, and whoever reads the result next (``A.vals``, ``b.array``,
# ``to_scipy`` ...) makes ITS stream wait for the event recorded at the end of the assembly -- the same deferral the
# interface exchange uses.  Matrices take turns on MPCX_MATRIX_STREAMS streams, one per object.  Default 1 since round 4:
# the blocks of a nest (a00, a01, a10 are independent matrices) gain nothing from running side by side (the row-block
# kernels fill every CU: Taylor-Hood 128^3 7.58 / 7.54 / 7.51 ms per step with 3 / 2 / 1 streams), and a THIRD
# high-priority stream turned out to share a hardware queue with the vector stream -- a matrix that landed on it lost
# the matrix / vector overlap (slab of config 2: 0.78 instead of 0.53 ms per step; GPU_MAX_HW_QUEUES does not help).
# MPCX_ASYNC_STREAMS=0 keeps everything on the caller's stream.
# ---------------------------------------------------------------------------------------------------------
_side = {}
_next_slot = [0]


def _async_enabled() -> bool:
    return os.environ.get("MPCX_ASYNC_STREAMS", "1") != "0"


# A/B switch (round 4's ordering: the caller's stream waits for the previous assembly into the same object)
_CHAIN_CALLER = os.environ.get("MPCX_SIDE_CHAIN_CALLER", "0") == "1"


@contextlib.contextmanager
def side_stream(kind: str, obj):
    """run the body on the library's ``kind`` ("matrix" / "vector") stream; ``obj`` (an MPCMatrix / Vector) gets the
    completion event its accessors wait for"""
    import torch

    if not (_async_enabled() and torch.cuda.is_available()) or getattr(obj, "device", None) is None or obj.device.type != "cuda":
        yield
        return
    cur = torch.cuda.current_stream(obj.device)
    slot = 0
    if kind == "matrix":
        slot = getattr(obj, "_side_slot", None)
        if slot is None:
            slot = obj._side_slot = _next_slot[0] % max(int(os.environ.get("MPCX_MATRIX_STREAMS", 1)), 1)
            _next_slot[0] += 1
    key = (kind, obj.device.index, slot)
    if key not in _side:
        # HIP stream priority (-1 = high).  The matrix streams are high-priority ones: the short, memory-bound matrix
        # kernels get the slots that free up while the long, VALU-bound vector kernel of the same step runs (config 2:
        # 3.56 -> 3.44 ms per step; vector high: 3.61).  MPCX_MATRIX_STREAM_PRIORITY / MPCX_VECTOR_STREAM_PRIORITY
        prio = int(os.environ.get("MPCX_%s_STREAM_PRIORITY" % kind.upper(), -1 if kind == "matrix" else 0))
        _side[key] = torch.cuda.Stream(device=obj.device, priority=prio)
    side = _side[key]
    raw = cur.cuda_stream
    if any(raw == st.cuda_stream for st in _side.values()):  # nested call from inside another assembly
        yield
        return
    # An earlier assembly into the same object that ran on THIS side stream is ordered by the stream itself.  The caller's
    # stream must not wait for it here: the next call's ``side.wait_stream(cur)`` would inherit that wait and chain the
    # vector assembly of step i + 1 behind the matrix hand-back of step i (locality twin, config 2: 5.2 ms per step with the
    # wait, the two streams in lockstep).  Results written by any other stream are waited for as before.
    if getattr(obj, "_ready_raw", None) != side.cuda_stream or _CHAIN_CALLER:
        obj._wait_ready()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        yield
        obj._ready = side.record_event()
        obj._ready_raw = side.cuda_stream


def wait_assembly():
    """make the current stream wait for everything the side streams hold (timing code, hand-written consumers that
    bypass the accessors)"""
    import torch

    if torch.cuda.is_available():
        cur = torch.cuda.current_stream()
        for st in _side.values():
            cur.wait_stream(st)


class _Pending:
    """An interface exchange that has been posted but not yet added into the owner's rows.  The packed send
    buffer travels while later kernels run (the matrix rows during the vector assembly); whoever reads the
    values next completes it first."""

    __slots__ = ("exchange", "handle")

    def __init__(self, exchange, handle):
        self.exchange, self.handle = exchange, handle

    def finish(self):
        self.exchange.finish(self.handle)


def _torch_dtype(dtype):
    """numpy-style dtype (None = float64) as a torch dtype"""
    import torch

    if dtype is None:
        return torch.float64
    if isinstance(dtype, torch.dtype):
        return dtype
    return {"float64": torch.float64, "float32": torch.float32, "complex128": torch.complex128,
            "complex64": torch.complex64}[np.dtype(dtype).name]


class Vector:
    """vector over the local (owned + ghost) dofs, resident in HBM (fp64 unless ``dtype`` says otherwise)."""

    def __init__(self, n: int, device=None, dtype=None):
        import torch

        self.device = device if device is not None else _native.require_gpu()
        self._array = torch.zeros(n, dtype=_torch_dtype(dtype), device=self.device)
        self._exchange = None  # distributed.SlabExchange of the space (partitioned meshes)
        self._pending = None
        self._ready = None  # event recorded at the end of an assembly on a side stream

    def _wait_ready(self):
        ev = self._ready
        if ev is not None:
            import torch

            # (nothing to wait for on the stream that recorded the event: the accessors are read several times per call
            # from inside the side stream itself)
            if torch._C._cuda_getCurrentRawStream(self.device.index) != getattr(self, "_ready_raw", None):
                torch.cuda.current_stream(self.device).wait_event(ev)

    @property
    def array(self):
        """the device tensor; the current stream first waits for an assembly still running on a side stream, and a
        posted ghost update is completed"""
        self._wait_ready()
        if self._pending is not None:
            p, self._pending = self._pending, None
            p.finish()
        return self._array

    def set(self, value: float):
        self.array.fill_(value)

    def numpy(self) -> np.ndarray:
        return self.array.detach().cpu().numpy()

    # PETSc-flavoured calls so reference-style drivers read the same
    @contextlib.contextmanager
    def localForm(self):
        yield self

    def attach_exchange(self, exchange):
        self._exchange = exchange

    def ghostUpdate(self, addv=None, mode=None):
        """``b.ghostUpdate(addv=ADD, mode=REVERSE)`` (bench_periodic.py:108): the partial sums of the ghost
        (interface-plane) rows are sent to their owner and added there; ``(INSERT, FORWARD)`` -- petsc4py's
        default when called without arguments: the owners' values are copied to the ghosts.  The other two
        combinations are not what the path uses and raise.  Single process / unpartitioned mesh: nothing to
        exchange."""
        addv = InsertMode.INSERT if addv is None else addv
        mode = ScatterMode.FORWARD if mode is None else mode
        add = int(addv) == int(InsertMode.ADD)
        reverse = int(mode) == int(ScatterMode.REVERSE)
        if add != reverse:
            raise NotImplementedError("ghostUpdate: (ADD, REVERSE) and (INSERT, FORWARD) are supported")
        if self._exchange is None:
            return None
        arr = self.array  # completes anything posted earlier
        if reverse:
            self._pending = _Pending(self._exchange, self._exchange.reduce_vector_begin(arr))
        else:
            self._exchange.forward_vector(arr)
        return None

    @property
    def size(self) -> int:
        return self._array.numel()


class NullSpace:
    """a set of vectors spanning a (near) null space -- the role of ``PETSc.NullSpace`` in
    ``A.setNearNullSpace(rigid_motions_nullspace(V))`` (python/benchmarks/bench_contact_3D.py:287,320).
    ``vectors``: list of host arrays over the local dofs; ``basis()``: (n, dim) array."""

    def __init__(self, vectors):
        self.vectors = [np.ascontiguousarray(v, dtype=np.float64) for v in vectors]

    @property
    def dim(self) -> int:
        return len(self.vectors)

    def basis(self) -> np.ndarray:
        return np.stack(self.vectors, axis=1)

    def getVecs(self):
        return self.vectors


def create_vector(V, dtype=None) -> Vector:
    """a vector over the dofs of ``V``; on a partitioned mesh with an initialised process group it carries the
    space's interface exchange (distributed.exchange_for)"""
    b = Vector(V.num_dofs, dtype=dtype)
    from .distributed import exchange_for

    ex = exchange_for(V)
    if ex is not None:
        b.attach_exchange(ex)
    return b


class MPCMatrix:
    """CSR matrix with the MPC sparsity pattern; pattern and values live on the GPU.
    ``rowptr`` is 64-bit (include/mpcx.h ``mpcx_nnz_t``): a matrix may hold more than 2^31 - 1
    entries on one GPU; column indices are 32-bit.  ``rowptr`` / ``cols`` may be given as numpy
    arrays or as device tensors (the device pattern builder hands over tensors, so a 17 GB column
    array never visits the host unless somebody asks for ``A.cols``)."""

    def __init__(self, rowptr, cols, ncols: int, device=None, dtype=None):
        import torch

        self.device = device if device is not None else _native.require_gpu()
        self.dtype = _torch_dtype(dtype)  # scalar type of the values (fp64: the tuned kernels)
        if isinstance(rowptr, torch.Tensor):
            self.d_rowptr = rowptr.to(self.device, torch.int64)
            self._rowptr = None
        else:
            self._rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
            self.d_rowptr = torch.from_numpy(self._rowptr).to(self.device)
        if isinstance(cols, torch.Tensor):
            self.d_cols = cols.to(self.device, torch.int32)
            self._cols = None
        else:
            self._cols = np.ascontiguousarray(cols, dtype=np.int32)
            self.d_cols = torch.from_numpy(self._cols).to(self.device)
        self.shape = (self.d_rowptr.numel() - 1, ncols)
        self._vals = None  # scalar CSR values, allocated on first use (a block-scalar matrix may never need them)
        self._plans = {}
        # block-scalar storage (component-diagonal forms on blocked spaces, include/mpcx.h mpcx_matrix_args_t::block_vals):
        # dict(bs, svals [nnz / bs^2], mask uint8 [nnz / bs^2], ov_pos int64, ov_val, diag_pos int64, diagval) or None
        self._compact = None
        self._compact_stale = False  # the scalar values do not reflect the block-scalar ones yet
        self._twin_stale = False  # the values live in the reordered twin's matrix (locality.py, lazy hand-back) and are not copied yet
        self._exchange = None
        self._pending = None
        self._ready = None  # event recorded at the end of an assembly on a side stream
        self.near_nullspace = None  # la.NullSpace, setNearNullSpace

    def setNearNullSpace(self, nullspace):
        """the near-null space the multigrid preconditioner builds its prolongators from (petsc4py ``Mat.setNearNullSpace``,
        python/benchmarks/bench_contact_3D.py:320)"""
        if nullspace is not None and nullspace.basis().shape[0] != self.shape[0]:
            raise ValueError("setNearNullSpace: the vectors do not have the matrix' number of rows")
        self.near_nullspace = nullspace

    def _wait_ready(self):
        ev = self._ready
        if ev is not None:
            import torch

            # (nothing to wait for on the stream that recorded the event: the accessors are read several times per call
            # from inside the side stream itself)
            if torch._C._cuda_getCurrentRawStream(self.device.index) != getattr(self, "_ready_raw", None):
                torch.cuda.current_stream(self.device).wait_event(ev)

    @property
    def vals(self):
        """the CSR values (device tensor); the current stream first waits for an assembly still running on a side
        stream, and a posted ``assemble()`` exchange is completed"""
        self._wait_ready()
        if self._vals is None:
            import torch

            self._vals = torch.zeros(self.d_cols.numel(), dtype=self.dtype, device=self.device)
            if self.device.type == "cuda":
                self._vals_streams = {torch._C._cuda_getCurrentRawStream(self.device.index)}  # the allocating stream
        elif self._ready is not None:
            # the values may have been allocated on a side stream (first assembly): tell the caching allocator about every
            # other stream that reads them, so that the block is not handed out again while such a read is queued
            import torch

            raw = torch._C._cuda_getCurrentRawStream(self.device.index)
            seen = getattr(self, "_vals_streams", None)
            if seen is not None and raw not in seen:
                self._vals.record_stream(torch.cuda.current_stream(self.device))
                seen.add(raw)
        if self._compact_stale:
            self._expand_compact()
        if self._twin_stale:
            from . import locality

            locality.hand_back(self)
        if self._pending is not None:
            p, self._pending = self._pending, None
            p.finish()
        return self._vals

    def _expand_compact(self):
        """scalar CSR values from the block-scalar ones: (k, k) entries of every block unless masked, zeros elsewhere,
        then the overlay (master contributions at their target positions, slave / Dirichlet diagonals)"""
        import torch

        from . import _device as D

        c = self._compact
        L = _native.lib()
        st = D.stream_ptr()
        bs = c["bs"]
        _native.check(L.mpcx_block_expand(self.shape[0] // bs, self.d_rowptr.data_ptr(), bs, c["svals"].data_ptr(),
                                          c["mask"].data_ptr(), self._vals.data_ptr(), st), "mpcx_block_expand")
        if c["ov_pos"] is not None and c["ov_pos"].numel():
            _native.check(L.mpcx_scatter_add_f64(self._vals.data_ptr(), c["ov_pos"].data_ptr(), c["ov_pos"].numel(),
                                                 c["ov_val"].data_ptr(), st), "mpcx_scatter_add_f64")
        if c["diag_pos"] is not None and c["diag_pos"].numel():
            dv = torch.full((c["diag_pos"].numel(),), float(c["diagval"]), dtype=torch.float64, device=self.device)
            _native.check(L.mpcx_scatter_add_f64(self._vals.data_ptr(), c["diag_pos"].data_ptr(), c["diag_pos"].numel(),
                                                 dv.data_ptr(), st), "mpcx_scatter_add_f64")
        self._compact_stale = False

    @property
    def is_block_scalar(self) -> bool:
        """the last assembly left the values in block-scalar storage (one value per bs x bs block + overlay); ``vals`` /
        ``to_scipy`` expand them on demand, ``problem.spmv`` multiplies straight from this layout"""
        return self._compact is not None and self._compact_stale

    @property
    def rowptr(self) -> np.ndarray:
        """host copy of the row offsets (int64), downloaded on first use"""
        if self._rowptr is None:
            self._rowptr = self.d_rowptr.cpu().numpy()
        return self._rowptr

    @property
    def cols(self) -> np.ndarray:
        """host copy of the column indices (int32), downloaded on first use"""
        if self._cols is None:
            self._cols = self.d_cols.cpu().numpy()
        return self._cols

    @property
    def nnz(self) -> int:
        return self.d_cols.numel()

    def zeroEntries(self):
        self._compact_stale = False
        self._twin_stale = False
        self.vals.zero_()

    def attach_exchange(self, exchange):
        self._exchange = exchange

    def assemblyBegin(self):
        """post the exchange of the interface rows' partial sums (PETSc MatAssemblyBegin); nothing waits here"""
        if self._exchange is not None:
            vals = self.vals  # completes anything posted earlier
            self._pending = _Pending(self._exchange, self._exchange.reduce_matrix_begin(vals))

    def assemblyEnd(self):
        """wait for the posted exchange and add the received sums into the owned rows (PETSc MatAssemblyEnd)"""
        _ = self.vals

    def assemble(self):
        """``A.assemble()`` (python/src/dolfinx_mpc/assemble_matrix.py:64): rows another rank owns are shipped to
        it and added there.  The transfer is posted here and completed when the values are next read
        (``A.vals``, ``to_scipy``, a solver, the next assembly), so that it overlaps whatever is enqueued in
        between; single process / unpartitioned mesh: nothing to ship."""
        self.assemblyBegin()
        return None

    def to_scipy(self):
        import scipy.sparse

        return scipy.sparse.csr_matrix((self.vals.detach().cpu().numpy(), self.cols, self.rowptr), shape=self.shape)

    def norm(self) -> float:
        import torch

        return float(torch.linalg.vector_norm(self.vals))
