"""Named timer scopes and profiler ranges round the library's entry points -- the scopes the reference registers with
``dolfinx.common.Timer`` ("~MPC: Assemble matrix (C++)" cpp/assemble_matrix.cpp:677, "~MPC: Assemble vector (C++)" /
"~MPC: Apply lifting (C++)" python/src/dolfinx_mpc/assemble_vector.py:47,99, "~MPC: Create Matrix" /
"~MPC: Create sparsity pattern" cpp/utils.h:149,388, "~MPC: Create slip constraint" / "~MPC: Inelastic condition"
cpp/ContactConstraint.h:367,914, "~MPC: Facet normal projection" python/src/dolfinx_mpc/utils/mpc_utils.py:75), under the
SAME names, so that a user's ``list_timings`` habit and grep patterns carry over.

* ``Timer(name)``: context manager / start-stop object like dolfinx's; every stop adds (1, wall seconds) to the registry.
  The assembly calls are asynchronous (kernels are enqueued on the library's streams): a scope measures the HOST side of a
  call unless ``MPCX_TIMER_SYNC=1`` makes every stop wait for the device first (then it is the reference's synchronous
  figure; off by default -- a synchronisation per call would serialise the two library streams).
* every scope is also a roctx range (``roctxRangePushA`` / ``roctxRangePop`` of libroctx64, when present): rocprofv3
  ``--marker-trace`` shows the same names on the timeline.
* ``timing(name)`` -> (count, wall seconds), ``timings()`` -> dict, ``list_timings()`` prints the table, ``reset_timings()``."""

from __future__ import annotations

import ctypes
import functools
import os
import time
from typing import Dict, Optional, Tuple

_registry: Dict[str, list] = {}
_roctx = None
_roctx_tried = False
_SYNC = os.environ.get("MPCX_TIMER_SYNC", "0") == "1"


def _roctx_lib():
    global _roctx, _roctx_tried
    if not _roctx_tried:
        _roctx_tried = True
        if os.environ.get("MPCX_ROCTX", "1") != "0":
            for name in ("libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so"):
                try:
                    lib = ctypes.CDLL(name)
                    lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                    lib.roctxRangePushA.restype = ctypes.c_int
                    lib.roctxRangePop.restype = ctypes.c_int
                    _roctx = lib
                    break
                except (OSError, AttributeError):
                    continue
    return _roctx


class Timer:
    """``with Timer("~MPC: ..."):`` or ``t = Timer(name); t.start(); ...; t.stop()`` (dolfinx.common.Timer's shape)"""

    __slots__ = ("name", "_t0", "_bname")

    def __init__(self, name: Optional[str] = None):
        self.name = name or ""
        self._bname = self.name.encode()
        self._t0 = None

    def start(self):
        lib = _roctx_lib()
        if lib is not None:
            lib.roctxRangePushA(self._bname)
        self._t0 = time.perf_counter()

    def stop(self) -> float:
        if self._t0 is None:
            return 0.0
        if _SYNC:
            try:
                import torch

                if torch.cuda.is_available():
                    torch.cuda.synchronize()
            except ImportError:
                pass
        dt = time.perf_counter() - self._t0
        self._t0 = None
        if _roctx is not None:
            _roctx.roctxRangePop()
        rec = _registry.get(self.name)
        if rec is None:
            _registry[self.name] = [1, dt]
        else:
            rec[0] += 1
            rec[1] += dt
        return dt

    def elapsed(self) -> Tuple[float]:
        return (0.0 if self._t0 is None else time.perf_counter() - self._t0,)

    def __enter__(self):
        self.start()
        return self

    def __exit__(self, *exc):
        self.stop()
        return False


def timing(name: str) -> Tuple[int, float]:
    """(number of calls, total wall seconds) of a scope; raises KeyError like dolfinx.common.timing for an unknown name"""
    n, t = _registry[name]
    return int(n), float(t)


def timings() -> Dict[str, Tuple[int, float]]:
    return {k: (int(v[0]), float(v[1])) for k, v in _registry.items()}


def reset_timings() -> None:
    _registry.clear()


def list_timings(file=None) -> str:
    """the table dolfinx.common.list_timings prints (one rank): name, reps, average and total wall seconds"""
    rows = sorted(_registry.items())
    w = max([len(k) for k, _ in rows] + [10])
    lines = [f"{'[MPI_AVG] Summary of timings':{w}s} |  reps    wall avg    wall tot" + ("   (stops synchronise the device)" if _SYNC else "   (host side of asynchronous calls; MPCX_TIMER_SYNC=1: device included)"),
             "-" * (w + 36)]
    for k, (n, t) in rows:
        lines.append(f"{k:{w}s} | {n:5d}  {t / max(n, 1):10.6f}  {t:10.6f}")
    text = "\n".join(lines)
    print(text, file=file)
    return text


def timed(name: str):
    """decorator: the call runs inside ``Timer(name)`` (and the roctx range of that name)"""

    def deco(fn):
        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            t = Timer(name)
            t.start()
            try:
                return fn(*args, **kwargs)
            finally:
                t.stop()

        return wrapper

    return deco
