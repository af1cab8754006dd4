#!/usr/bin/env python
"""Do the HBM-bound matrix cluster kernel and the VALU-bound vector cluster kernel overlap when both are resident
on every CU?  Times (HIP events) the two kernels of config 2 alone at full and at halved occupancy (LDS floors:
MPCX_CUBE_LDS_FLOOR / MPCX_VCUBE_LDS_FLOOR are read once per process, so every arm is a child run) and together on
two streams.  Usage: python tools/overlap_probe.py [N]   (parent: spawns the arms)"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def arm(N):
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
    Lib = _native.lib()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
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

    out = {"matrix_alone_ms": timed([fm]), "vector_alone_ms": timed([fv]), "both_two_streams_ms": timed([fm, fv]),
           "both_two_streams_vm_ms": timed([fv, fm])}
    print("ARM " + json.dumps(out))


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "arm" else 256
    if len(sys.argv) > 1 and sys.argv[1] == "arm":
        arm(int(sys.argv[2]))
        sys.exit(0)
    arms = {"full occupancy": {}, "matrix 1 WG/CU": {"MPCX_CUBE_LDS_FLOOR": "90000"},
            "vector 1 WG/CU": {"MPCX_VCUBE_LDS_FLOOR": "90000"},
            "matrix 1 WG/CU + vector free": {"MPCX_CUBE_LDS_FLOOR": "100000"},
            "both 1 WG/CU": {"MPCX_CUBE_LDS_FLOOR": "80000", "MPCX_VCUBE_LDS_FLOOR": "80000"}}
    for name, env in arms.items():
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "arm", str(N)], env=dict(os.environ, **env),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("ARM ")]
        print(name, env, line[0][4:] if line else r.stderr[-800:], flush=True)
