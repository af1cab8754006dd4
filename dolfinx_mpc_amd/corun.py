"""Co-running the matrix and the vector assembly of a step on one GPU.

The reference assembles A and b one after the other on the host (python/benchmarks/bench_periodic.py:97-108; the loops
are cpp/assemble_matrix.cpp:488-547 and cpp/assemble_vector.cpp:65-90).  Here the two calls run on two library streams
(la.side_stream), and their kernels have complementary bounds: the matrix row-block kernels are HBM-bound (config 5:
HBM busy 0.86 of the copy rate, VALU issue 0.39), the vector kernels VALU-bound (VALU 0.70, HBM 0.15).  Left alone they
still run one after the other -- whichever kernel is dispatched first fills every CU (LDS, wave slots, registers) and the
other one gets the slots of its tail: the step costs the SUM (config 5: 9.9 + 5.5 = 15.9 ms).

What this module does: a row-block launch of the matrix side is cut in two sub-ranges of its row blocks (a sub-range of a
plan is the same plan with ``block_row0`` / ``block_ent_off`` advanced -- no kernel knows).  The FIRST part is launched
with an LDS floor (``mpcx_matrix_args_t::lds_floor``): at most ``MATRIX_WGS`` of its workgroups share a CU, so the vector
kernel that arrives on the other stream finds LDS, wave slots and registers on EVERY CU and the two kinds of workgroups
are co-resident; the SECOND part is launched without a floor and takes whatever the vector kernel has left when it ends.
The vector launch gets a floor of its own so that it cannot crowd the matrix part out either.

Policy (``MPCX_CORUN``): ``0`` never; ``1`` always; ``auto`` (default): a matrix object is launched this way when the
PREVIOUS assembly into it was still running while a vector assembly was enqueued (``note_vector_call``) -- the
time-loop / benchmark pattern -- so a caller who only assembles matrices keeps the uncapped launch.
"""

from __future__ import annotations

import ctypes as C
import os

from . import _native

LDS_CU = 160 * 1024


def _env_float(name, default):
    try:
        return float(os.environ.get(name, default))
    except ValueError:
        return float(default)


def mode() -> str:
    return os.environ.get("MPCX_CORUN", "0").lower()


def matrix_floor(wgs: int) -> int:
    """smallest LDS request per workgroup that keeps a kernel at <= ``wgs`` workgroups per CU"""
    return LDS_CU // (wgs + 1) + 512


def params(kernel_name: str | None = None) -> dict:
    """split fraction and floors of a co-run launch (environment overrides for the sweeps of tools/probes/corun_probe.py)"""
    wgs = int(_env_float("MPCX_CORUN_MATRIX_WGS", 2))
    mfloor = int(_env_float("MPCX_CORUN_MATRIX_FLOOR", matrix_floor(wgs)))
    return {"frac": _env_float("MPCX_CORUN_FRAC", 0.6), "matrix_floor": mfloor,
            "vector_floor": int(_env_float("MPCX_CORUN_VECTOR_FLOOR", 0)),
            "vector_lds_max": LDS_CU - (LDS_CU // mfloor) * mfloor if mfloor > 0 else LDS_CU}


# ---------------------------------------------------------------------------------------------------------
# "auto": who runs beside whom.  A matrix assembly records its completion event (la.side_stream: ``A._ready``); a vector
# assembly that is enqueued while the latest matrix event of the device is still pending marks that matrix object
# ("a vector call arrived while I was running") and the next assembly into it is cut.  A cut assembly that no vector call
# met clears the mark again.
# ---------------------------------------------------------------------------------------------------------
_last_matrix = {}  # device index -> MPCMatrix of the latest assembly


def note_matrix_call(A):
    """called by assemble_matrix before it enqueues: returns True if this assembly should be launched in two parts"""
    m = mode()
    if m in ("0", "off", "no"):
        _last_matrix.clear()
        return False
    dev = getattr(A.device, "index", 0)
    met = getattr(A, "_corun_met", None)
    _last_matrix[dev] = A
    A._corun_met = False  # set by note_vector_call while this assembly is in flight
    if m in ("1", "on", "always"):
        return True
    return bool(met)


def note_vector_call(device):
    """called by assemble_vector before it enqueues: is a matrix assembly of this device still in flight?"""
    m = mode()
    if m in ("0", "off", "no"):
        return False  # (nothing is looked at: an event query is not allowed while a stream is being captured into a graph)
    if m in ("1", "on", "always"):
        return True
    import torch

    A = _last_matrix.get(getattr(device, "index", 0))
    if A is None or torch.cuda.is_current_stream_capturing():
        return False
    ev = getattr(A, "_ready", None)
    try:
        running = ev is not None and not ev.query()
    except Exception:  # noqa: BLE001  (an event of a destroyed stream)
        running = False
    if running:
        A._corun_met = True
    return running


def splittable(a) -> bool:
    """row-block family launches (entity lists, pair records, node blocks) with enough blocks to cut"""
    return (a.algorithm == 2 and a.plan.num_blocks >= int(_env_float("MPCX_CORUN_MIN_BLOCKS", 2048))
            and bool(a.plan.block_row0) and bool(a.plan.block_ent_off) and a.n_entities > 0)


def split(a, frac: float, floor: int):
    """the two argument blocks of a cut launch: blocks [0, k) with the LDS floor and without the master contributions,
    blocks [k, n) without a floor and with everything that follows the bulk kernel (master contributions: the row blocks
    are written in store mode, so they must all be in place first)"""
    nb = int(a.plan.num_blocks)
    k = int(nb * frac)
    if nb >= 64:
        k = (k // 8) * 8
    if k <= 0 or k >= nb:
        return [a]
    first = _native.MatrixArgs.from_buffer_copy(a)
    second = _native.MatrixArgs.from_buffer_copy(a)
    for name in ("leftover", "kernel_name", "block_scalar", "second"):
        if hasattr(a, name):
            setattr(first, name, getattr(a, name))
            setattr(second, name, getattr(a, name))
    first.plan.num_blocks = k
    first.lds_floor = int(floor)
    first.n_slave_entities = 0
    first.mpc_plan_targets = 0
    second.plan.num_blocks = nb - k
    second.plan.block_row0 = a.plan.block_row0 + 4 * k
    second.plan.block_ent_off = a.plan.block_ent_off + 8 * k
    second.lds_floor = 0
    return [first, second]


def split_calls(calls, p: dict):
    """expand the (memset, args, keep) list of assemble_matrix"""
    out = []
    for memset, a, keep in calls:
        if not splittable(a):
            out.append((memset, a, keep))
            continue
        parts = split(a, p["frac"], p["matrix_floor"])
        for n, part in enumerate(parts):
            out.append((memset and n == 0, part, keep))
    return out
