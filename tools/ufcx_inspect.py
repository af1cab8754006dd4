#!/usr/bin/env python
"""Compile an imported UFCx kernel with hipRTC (no GPU needed) and print the resource usage of every generated
kernel (VGPRs, SGPRs, scratch bytes, LDS) from the code object's metadata.

    python tools/ufcx_inspect.py tests/ufcx/laplace_p1_tet.c tabulate_tensor_laplace_p1_tet 2 4 1 4 1 [nv]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def inspect(source: str, name: str, rank: int, nd0: int, bs0: int, nd1: int = 0, bs1: int = 0, nv: int = 4, dump=None):
    import ctypes as C

    from dolfinx_mpc_amd import _native

    L = _native.lib()
    d = _native.UfcxDescT(source.encode(), name.encode(), rank, nd0, bs0, nd1, bs1, nv)
    h = L.mpcx_ufcx_compile(d)
    if not h:
        raise RuntimeError(L.mpcx_last_error().decode())
    n = L.mpcx_ufcx_code_size(h)
    buf = C.create_string_buffer(n)
    L.mpcx_ufcx_code(h, buf)
    L.mpcx_ufcx_free(h)
    path = dump or os.path.join(tempfile.gettempdir(), f"ufcx_{name}.co")
    with open(path, "wb") as fh:
        fh.write(buf.raw)
    notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", path], stdout=subprocess.PIPE, text=True).stdout
    out = {}
    for blk in notes.split("- .agpr_count")[1:]:
        g = lambda key: (re.search(rf"\.{key}:\s+(\S+)", blk) or [None, "?"])[1]  # noqa: E731
        out[g("name")] = dict(vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), scratch=g("private_segment_fixed_size"),
                              lds=g("group_segment_fixed_size"), spill=g("vgpr_spill_count"))
    return out, path


if __name__ == "__main__":
    src = open(sys.argv[1]).read()
    res, path = inspect(src, sys.argv[2], *(int(v) for v in sys.argv[3:]))
    for k, v in res.items():
        print(f"{k:36s} {v}")
    print("code object:", path)
