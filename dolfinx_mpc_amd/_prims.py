"""Device primitives of the set-up path: thin wrappers over libmpcx's rocPRIM entry points (include/mpcx.h:
mpcx_scan_exclusive_*, mpcx_sort_pairs_*, mpcx_run_heads / mpcx_run_fill).  torch only allocates the buffers; every
scan, sort and run-length step of the plan builders goes through the C ABI, so a caller without torch can do the same
with hipMalloc'd memory."""

from __future__ import annotations

import ctypes as C

from . import _device as D
from . import _native


def _workspace(query, dev):
    """run ``query(temp_ptr, byref(size))`` with temp = NULL to learn the size, allocate, return (tensor, size)"""
    import torch

    nbytes = C.c_size_t(0)
    _native.check(query(None, C.byref(nbytes)), "workspace query")
    temp = torch.empty(max(int(nbytes.value), 1), dtype=torch.uint8, device=dev)
    return temp, nbytes


def scan_i32_i64(counts):
    """exclusive scan of an int32 tensor into int64 [n + 1] (last entry: the total)"""
    import torch

    L = _native.lib()
    n = counts.numel()
    out = torch.empty(n + 1, dtype=torch.int64, device=counts.device)
    st = D.stream_ptr()
    call = lambda t, nb: L.mpcx_scan_exclusive_i32_i64(counts.data_ptr(), n, out.data_ptr(), t, nb, st)  # noqa: E731
    temp, nb = _workspace(call, counts.device)
    _native.check(call(temp.data_ptr(), C.byref(nb)), "mpcx_scan_exclusive_i32_i64")
    return out


def scan_i32(counts):
    """exclusive scan of an int32 tensor into int32 [n + 1]"""
    import torch

    L = _native.lib()
    n = counts.numel()
    out = torch.empty(n + 1, dtype=torch.int32, device=counts.device)
    st = D.stream_ptr()
    call = lambda t, nb: L.mpcx_scan_exclusive_i32(counts.data_ptr(), n, out.data_ptr(), t, nb, st)  # noqa: E731
    temp, nb = _workspace(call, counts.device)
    _native.check(call(temp.data_ptr(), C.byref(nb)), "mpcx_scan_exclusive_i32")
    return out


def scan_i64(counts):
    """exclusive scan of an int64 tensor into int64 [n + 1]"""
    import torch

    L = _native.lib()
    n = counts.numel()
    out = torch.empty(n + 1, dtype=torch.int64, device=counts.device)
    st = D.stream_ptr()
    call = lambda t, nb: L.mpcx_scan_exclusive_i64(counts.data_ptr(), n, out.data_ptr(), t, nb, st)  # noqa: E731
    temp, nb = _workspace(call, counts.device)
    _native.check(call(temp.data_ptr(), C.byref(nb)), "mpcx_scan_exclusive_i64")
    return out


def segment_offsets(sorted_keys, shift: int, num_segments: int):
    """int64 [num_segments + 1]: first position of every segment id (key >> shift) in a sorted key tensor"""
    import torch

    out = torch.empty(num_segments + 1, dtype=torch.int64, device=sorted_keys.device)
    _native.check(_native.lib().mpcx_segment_offsets(sorted_keys.data_ptr(), sorted_keys.numel(), shift, num_segments,
                                                     out.data_ptr(), D.stream_ptr()), "mpcx_segment_offsets")
    return out


def sort_pairs(keys, vals, end_bit: int = 64):
    """stable sort of (int64 key, int32 | int64 value) pairs by the key bits [0, end_bit); returns new tensors"""
    import torch

    L = _native.lib()
    n = keys.numel()
    assert keys.dtype == torch.int64 and vals.numel() == n
    fn = L.mpcx_sort_pairs_i64_i32 if vals.dtype == torch.int32 else L.mpcx_sort_pairs_i64_i64
    assert vals.dtype in (torch.int32, torch.int64)
    ko, vo = torch.empty_like(keys), torch.empty_like(vals)
    st = D.stream_ptr()
    end_bit = max(1, min(int(end_bit), 64))
    call = lambda t, nb: fn(keys.data_ptr(), ko.data_ptr(), vals.data_ptr(), vo.data_ptr(), n, 0, end_bit, t, nb, st)  # noqa: E731
    temp, nb = _workspace(call, keys.device)
    _native.check(call(temp.data_ptr(), C.byref(nb)), "mpcx_sort_pairs")
    return ko, vo


def runs(sorted_keys, want_keys: bool = True, want_marks: bool = False):
    """run-length structure of a sorted int64 tensor: (run_keys [nr] or None, run_start int64 [nr + 1]); with
    ``want_marks`` also (heads int32 [n], their exclusive scan int64 [n + 1]): element t lies in run
    ``scan[t] + heads[t] - 1``"""
    import torch

    L = _native.lib()
    n = sorted_keys.numel()
    dev = sorted_keys.device
    if n == 0:
        out = (torch.empty(0, dtype=torch.int64, device=dev) if want_keys else None), torch.zeros(1, dtype=torch.int64, device=dev)
        return out + (torch.empty(0, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)) if want_marks else out
    st = D.stream_ptr()
    heads = torch.empty(n, dtype=torch.int32, device=dev)
    _native.check(L.mpcx_run_heads(sorted_keys.data_ptr(), n, heads.data_ptr(), st), "mpcx_run_heads")
    excl = scan_i32_i64(heads)
    nr = int(excl[-1].item())
    rk = torch.empty(nr, dtype=torch.int64, device=dev) if want_keys else None
    rs = torch.empty(nr + 1, dtype=torch.int64, device=dev)
    _native.check(L.mpcx_run_fill(sorted_keys.data_ptr(), heads.data_ptr(), excl.data_ptr(), n, D.ptr(rk), rs.data_ptr(), st),
                  "mpcx_run_fill")
    return (rk, rs, heads, excl) if want_marks else (rk, rs)
