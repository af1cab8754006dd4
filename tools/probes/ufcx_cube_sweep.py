#!/usr/bin/env python
"""Config 2 (periodic Poisson, P1 tets, N^3 cubes) with the benchmark's forms as FFCx-shaped C text, timed under several
compile settings of the imported-kernel cluster path (csrc/mpcx_ufcx.cpp: MPCX_UFCX_FP, MPCX_UFCX_CUBE_THREADS,
MPCX_UFCX_CUBE_PIPE, ...).  One process, one problem; every setting recompiles the two kernels and rebuilds nothing else.

    python tools/probes/ufcx_cube_sweep.py [N] "KEY=VAL KEY=VAL" "KEY=VAL" ...

Prints per setting: matrix call ms, vector call ms, step ms (matrix + vector back to back, wall clock), and the error
against the built-in operators' result."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    import torch

    import dolfinx_mpc_amd as dm
    from bench import hip_time
    from dolfinx_mpc_amd import _device as D
    from dolfinx_mpc_amd import fem
    from dolfinx_mpc_amd.codegen import BENCH_PERIODIC_F, generate
    from dolfinx_mpc_amd.quadrature import make_quadrature

    args = sys.argv[1:]
    N = int(args[0]) if args and args[0].isdigit() else 256
    settings = [a for a in args if not a.isdigit()] or [""]

    class A:
        pass

    ba = A()
    ba.n, ba.config, ba.cell, ba.numbering, ba.ufcx, ba.scaling = N, 2, "tet", "tiled", None, "strong"
    ba.no_tile, ba.tile, ba.degree = False, [8, 8, 8], 1
    import bench

    w = bench.poisson_workload(ba, 0, 1, 1)
    label, f_ref, (m0, m1) = w.blocks[0]
    lv, L_ref, mv = w.vectors[0]
    bcs = w.bcs
    A_ref = dm.assemble_matrix(f_ref, (m0, m1), bcs=bcs)
    b_ref = dm.assemble_vector(L_ref, mv)
    torch.cuda.synchronize()
    vref = A_ref.vals.clone()
    bref = b_ref.array.clone()
    sa, na = generate("stiffness", "tetrahedron", 1, 1, make_quadrature("tetrahedron", 0))
    sl, nl = generate("source", "tetrahedron", 1, 1, make_quadrature("tetrahedron", 5), fexpr=BENCH_PERIODIC_F)
    for setting in settings:
        kv = dict(p.split("=", 1) for p in setting.split() if "=" in p)
        old = {k: os.environ.get(k) for k in kv}
        os.environ.update(kv)
        D._ufcx_handles.clear()
        t0 = time.time()
        fa, fl = fem.form_ufcx([w.V, w.V], sa, na), fem.form_ufcx([w.V], sl, nl)
        Au = dm.assemble_matrix(fa, (m0, m1), bcs=bcs, A=A_ref)
        bu = dm.assemble_vector(fl, mv, b=b_ref)
        torch.cuda.synchronize()
        t_first = time.time() - t0
        ea = float((Au.vals - vref).abs().max() / vref.abs().max())
        eb = float((bu.array - bref).abs().max() / bref.abs().max())
        tm = hip_time(lambda: dm.assemble_matrix(fa, (m0, m1), bcs=bcs, A=A_ref), 10)
        tv = hip_time(lambda: dm.assemble_vector(fl, mv, b=b_ref), 10)

        def step():
            dm.assemble_matrix(fa, (m0, m1), bcs=bcs, A=A_ref)
            dm.assemble_vector(fl, mv, b=b_ref)

        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        ts = (time.perf_counter() - t0) / 20 * 1e3
        print(f"[{setting or 'default'}] matrix {tm:.3f} ms  vector {tv:.3f} ms  step {ts:.3f} ms  first {t_first:.2f} s  "
              f"err A {ea:.1e} b {eb:.1e}", flush=True)
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        del fa, fl


if __name__ == "__main__":
    main()
