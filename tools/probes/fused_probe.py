#!/usr/bin/env python
"""mpcx_assemble_fused (matrix + vector cluster kernels of config 2 in one launch) against the two separate launches: same
values, and the time of a step.  python tools/probes/fused_probe.py [N]"""
import argparse
import ctypes as C
import os
import sys
import time

os.environ.setdefault("MPCX_VCUBE_ROWS", "512")  # the vector plan on the matrix plan's row blocks (512-row tiles)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(N=256, timing=True):
    import torch

    import bench
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import _native
    from dolfinx_mpc_amd.la import create_vector

    am = sys.modules["dolfinx_mpc_amd.assemble_matrix"]
    av = sys.modules["dolfinx_mpc_amd.assemble_vector"]
    args = argparse.Namespace(n=N, no_tile=False, tile=[8, 8, 8], scaling="strong", numbering="tiled", cell="tet", ufcx=None)
    w = bench.poisson_workload(args, 0, 1, 1)
    label, f, (m0, m1) = w.blocks[0]
    lv, Lf, mv = w.vectors[0]
    A = dm.create_matrix(f, m0, m1)
    b = create_vector(w.V)
    dm.assemble_matrix(f, (m0, m1), bcs=w.bcs, A=A)
    dm.assemble_vector(Lf, mv, b=b)
    torch.cuda.synchronize()
    Lib = _native.lib()
    ma, km = am.matrix_args(f, 0, A, m0, m1, w.bcs, 2, store_mode=1)
    va, kv = av.vector_args(Lf, 0, b, mv, 0)
    assert ma.kernel_name == "cube" and va.kernel_name == "cube_own", (ma.kernel_name, va.kernel_name)
    chain = [ma]
    while getattr(chain[-1], "second", None) is not None:
        chain.append(chain[-1].second)
    print("matrix parts:", [(int(c.plan.num_blocks), int(c.cube_rec_bytes), int(c.cube_flags), int(c.n_slave_entities)) for c in chain],
          "vector blocks:", int(va.plan.num_blocks), "max rows (own+halo):", int(va.plan.max_rows))
    # the plans' row blocks
    cp = [v for k, od in A._plans.items() if k == ("objcache", "cubes") for v in od.values()][0]
    parts, keep, info = cp[1] if isinstance(cp, tuple) and len(cp) == 2 else cp
    d_row0 = keep[0]
    own = [v for k, od in w.mesh._device.items() if k == ("objcache", "vcube_own") for v in od.values()][0]
    own = own[1] if isinstance(own, tuple) and len(own) == 2 else own
    v_row0 = own[1][0]
    same = d_row0.shape == v_row0.shape and bool(torch.equal(d_row0.cpu(), v_row0.cpu()))
    print("same row blocks:", same, tuple(d_row0.shape), tuple(v_row0.shape))
    if not same:
        return None
    nb = int(va.plan.num_blocks)
    part_index = torch.full((nb,), -1, dtype=torch.int32, device="cuda")
    p0 = parts[0]
    assert p0[2] == 64 and p0[5] == 1, "part 0 is not the narrow / parallelepiped part"
    ids = p0[3]
    if ids is None:
        part_index[:] = torch.arange(nb, dtype=torch.int32, device="cuda")
    else:
        part_index[ids.long()] = torch.arange(ids.numel(), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for c in chain:
        c.stream = st
    va.stream = st

    def separate():
        b.array.zero_()
        for c in chain:
            _native.check(Lib.mpcx_assemble_matrix(C.byref(c)), "m")
        _native.check(Lib.mpcx_assemble_vector(C.byref(va)), "v")

    def fused():
        b.array.zero_()
        _native.check(Lib.mpcx_assemble_fused(C.byref(chain[0]), C.byref(va), part_index.data_ptr()), "fused")
        for c in chain[1:]:
            _native.check(Lib.mpcx_assemble_matrix(C.byref(c)), "m")

    separate()
    torch.cuda.synchronize()
    vref, bref = A._vals.clone(), b.array.clone()
    A._vals.zero_()
    fused()
    torch.cuda.synchronize()
    ea = float((A._vals - vref).abs().max() / vref.abs().max())
    eb = float((b.array - bref).abs().max() / bref.abs().max())
    print(f"fused vs separate: |dA| {ea:.2e}  |db| {eb:.2e}")
    if not timing:
        return {"dA": ea, "db": eb}

    def timed(fn, n=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def api_step():
        dm.assemble_matrix(f, (m0, m1), bcs=w.bcs, A=A)
        dm.assemble_vector(Lf, mv, b=b)

    t = {"separate_one_stream_ms": timed(separate), "fused_ms": timed(fused), "api_two_streams_ms": timed(api_step)}
    print(f"one stream, separate launches: {t['separate_one_stream_ms']:.3f} ms   fused: {t['fused_ms']:.3f} ms   "
          f"API step (two streams): {t['api_two_streams_ms']:.3f} ms")
    return {"dA": ea, "db": eb, **t}


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 256)
