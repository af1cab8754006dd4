#!/usr/bin/env python
"""Config 2's problem (periodic Poisson, P1 tets, N^3 cubes) in the reference's other three scalar types (float32, complex64,
complex128: python/src/dolfinx_mpc/multipointconstraint.py:55-64, cpp/assemble_matrix.cpp:729-812) -- VERDICT r4 U-4: "no
timing exists".  These run the general kernels of csrc/mpcx_scalar.hip (LDS row blocks in the storage type, or thread per
entity with device atomics); float64 on the same general kernels (MPCX_NO_CUBE=1) and on its tuned ones are printed beside
them.  One JSON line.

    python tools/bench_scalar_types.py [N]   (default 128: the complex128 matrix of 256^3 alone is 4 GB of values)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    import torch

    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import fem
    from dolfinx_mpc_amd.mesh import create_box

    N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    mesh = create_box((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), (N, N, N), "tetrahedron", (8, 8, 8))
    V = fem.functionspace(mesh, ("Lagrange", 1))
    walls = fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[1], 0) | np.isclose(x[1], 1) | np.isclose(x[2], 0) | np.isclose(x[2], 1))

    def rel(x):
        o = x.copy()
        o[0] = 1 - x[0]
        return o

    out = {"N": N, "dofs": int(V.num_dofs), "cells": int(mesh.num_cells), "types": {}}

    def timed(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    for name, T, env in (("float64 (tuned: cluster kernels)", np.float64, {}), ("float64 (general row-block kernels)", np.float64, {"MPCX_NO_CUBE": "1"}),
                         ("float32", np.float32, {}), ("complex64", np.complex64, {}), ("complex128", np.complex128, {})):
        os.environ.update(env)
        try:
            cplx = np.issubdtype(np.dtype(T), np.complexfloating)
            bc = fem.dirichletbc(np.array(0.0, dtype=T), walls, V)
            mpc = dm.MultiPointConstraint(V, dtype=T)
            mpc.create_periodic_constraint_geometrical(V, lambda x: np.isclose(x[0], 1), rel, [bc], scale=1.0)  # (a real scale: timing only)
            mpc.finalize()
            a = fem.form(fem.form_stiffness(V, constant=(2.0 - 1.0j) if cplx else 1.0), dtype=T)
            L = fem.form(fem.form_source(V, fem.FN_BENCH_PERIODIC), dtype=T)
            A = dm.assemble_matrix(a, mpc, bcs=[bc])
            b = dm.assemble_vector(L, mpc)
            rec = {}
            for alg in ("rowblock", "atomic") if T is not np.float64 else (None,):
                tm = timed(lambda: dm.assemble_matrix(a, mpc, bcs=[bc], A=A, algorithm=alg))
                tv = timed(lambda: dm.assemble_vector(L, mpc, b=b, algorithm=alg))
                rec[alg or "auto"] = {"assemble_matrix_ms": round(tm, 3), "assemble_vector_ms": round(tv, 3),
                                      "DoFs_per_s": V.num_dofs / ((tm + tv) * 1e-3)}
            out["types"][name] = rec
            del A, b, a, L, mpc
        except Exception as e:  # noqa: BLE001
            out["types"][name] = {"error": repr(e)[:300]}
        finally:
            for k in env:
                os.environ.pop(k, None)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
