#!/usr/bin/env python
"""Config 2's two cluster kernels on DISJOINT sets of compute units (hipExtStreamCreateWithCUMask): the matrix kernel is
HBM / LDS bound (VALU issue 0.29) and the vector kernel VALU bound (0.85), but on two ordinary streams they hardly overlap --
the matrix workgroups take 148 of the 160 KB of LDS of every CU they are resident on.  Does giving the
matrix kernel a few CUs of its own and the vector kernel the rest shorten the step?  Prints alone / together times per split.

    python tools/probes/cu_mask_probe.py [N]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main(N):
    import numpy as np
    import torch

    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import _native, fem
    from dolfinx_mpc_amd.la import create_vector
    from dolfinx_mpc_amd.mesh import create_box

    am = sys.modules["dolfinx_mpc_amd.assemble_matrix"]
    av = sys.modules["dolfinx_mpc_amd.assemble_vector"]
    mesh = create_box((0, 0, 0), (1, 1, 1), (N, N, N), "tetrahedron", (8, 8, 8))
    V = fem.functionspace(mesh, ("Lagrange", 1))
    bc = fem.dirichletbc(0.0, fem.locate_dofs_geometrical(
        V, lambda x: np.isclose(x[1], 0) | np.isclose(x[1], 1) | np.isclose(x[2], 0) | np.isclose(x[2], 1)), V)
    mpc = dm.MultiPointConstraint(V)

    def rel(x):
        o = x.copy()
        o[0] = 1 - x[0]
        return o

    mpc.create_periodic_constraint_geometrical(V, lambda x: np.isclose(x[0], 1), rel, [bc])
    mpc.finalize()
    a, L = fem.form_stiffness(V), fem.form_source(V, fem.FN_BENCH_PERIODIC)
    A = dm.create_matrix(a, mpc)
    b = create_vector(V)
    dm.assemble_matrix(a, mpc, bcs=[bc], A=A)
    dm.assemble_vector(L, mpc, b=b)
    torch.cuda.synchronize()
    Lib = _native.lib()
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    hip.hipExtStreamCreateWithCUMask.restype = C.c_int

    def masked_stream(bits):
        words = (C.c_uint32 * 8)(*[sum(1 << (i - 32 * w) for i in bits if 32 * w <= i < 32 * w + 32) for w in range(8)])
        s = C.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
        if rc != 0:
            raise RuntimeError(f"hipExtStreamCreateWithCUMask rc {rc}")
        return torch.cuda.ExternalStream(s.value)

    def run(s1, s2, label):
        with torch.cuda.stream(s1):
            ma, k1 = am.matrix_args(a, 0, A, mpc, mpc, [bc], 2, store_mode=1, with_mpc_kernel=False)
        with torch.cuda.stream(s2):
            va, k2 = av.vector_args(L, 0, b, mpc, 0)
        fm = lambda: _native.check(Lib.mpcx_assemble_matrix(C.byref(ma)), "m")  # noqa: E731
        fv = lambda: _native.check(Lib.mpcx_assemble_vector(C.byref(va)), "v")  # noqa: E731

        def timed(fns, reps=10):
            torch.cuda.synchronize()
            for f in fns:
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            s1.wait_event(e0), s2.wait_event(e0)
            for _ in range(reps):
                for f in fns:
                    f()
            d1, d2 = torch.cuda.Event(), torch.cuda.Event()
            d1.record(s1), d2.record(s2)
            torch.cuda.current_stream().wait_event(d1), torch.cuda.current_stream().wait_event(d2)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        out = {"matrix_alone_ms": round(timed([fm]), 3), "vector_alone_ms": round(timed([fv]), 3),
               "both_ms": round(timed([fm, fv]), 3), "both_vector_first_ms": round(timed([fv, fm]), 3)}
        print(label, json.dumps(out), flush=True)

    run(torch.cuda.Stream(), torch.cuda.Stream(), "two ordinary streams")
    allc = list(range(256))
    for n in (16, 32, 48, 64, 96):
        for layout in ("strided", "low"):
            m = allc[:: 256 // n][:n] if layout == "strided" else allc[:n]
            rest = [c for c in allc if c not in set(m)]
            try:
                run(masked_stream(m), masked_stream(rest), f"matrix on {n} CUs ({layout}), vector on {256 - n}")
            except RuntimeError as e:
                print("failed:", e)
                return


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 256)
