"""TEST INFRASTRUCTURE (oracle): the CPU restatement run on all host cores, the way the
reference is deployed (one serial assembly loop per MPI rank over its share of the cells,
python/benchmarks/Makefile `mpirun -n 23`).

Cells are split into P contiguous slabs; P forked workers each run the oracle's loops
(oracle/mpc_oracle.c through oracle/pyoracle.py) over their slab into private value arrays of
the global pattern and then add up one segment each of all arrays -- the stand-in for PETSc's off-process stash and
`b.ghostUpdate(ADD, REVERSE)`.  Timed: wall time from the common start to the last worker's exit.
Used only by bench.py's cpu_baseline leg, in a fresh interpreter (no torch, no HIP runtime in
the forked processes).

    python -m oracle.cpu_parallel N P   ->  one JSON line
"""

from __future__ import annotations

import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(N: int, P: int):
    from dolfinx_mpc_amd import fem
    from oracle import pyoracle as po
    from problems import case_cube_periodic, oracle_mpc

    case = case_cube_periodic(N, 1, 0.0)
    V = case.V
    mpc = oracle_mpc(po, case)
    pattern = po.create_pattern(case.a, mpc, mpc)
    nnz, ndofs, ncells = pattern[1].size, V.num_dofs, case.mesh.num_cells
    bounds = np.linspace(0, ncells, P + 1).astype(np.int64)
    ctx = mp.get_context("fork")
    vals = [ctx.RawArray("d", int(nnz)) for _ in range(P)]
    vecs = [ctx.RawArray("d", int(ndofs)) for _ in range(P)]
    go = ctx.Barrier(P + 1)
    done = ctx.Barrier(P)
    times = ctx.RawArray("d", 3 * P)
    A_sum = ctx.RawArray("d", int(nnz))
    b_sum = ctx.RawArray("d", int(ndofs))
    seg_a = np.linspace(0, nnz, P + 1).astype(np.int64)
    seg_b = np.linspace(0, ndofs, P + 1).astype(np.int64)

    def work(r):
        cells = np.arange(bounds[r], bounds[r + 1], dtype=np.int32)
        a = fem.form_stiffness(V, cells=cells)
        L = fem.form_source(V, fem.FN_BENCH_PERIODIC, cells=cells)
        out_a = np.frombuffer(vals[r], dtype=np.float64)
        out_b = np.frombuffer(vecs[r], dtype=np.float64)
        go.wait()
        t0 = time.perf_counter()
        # slave / Dirichlet diagonals are added once, by the parent's rank-0 equivalent (bcs=[] here
        # would change the zeroing of Dirichlet rows, so every worker passes the bcs and the
        # diagonal entries are simply overwritten by the parent afterwards)
        po.assemble_matrix(a, mpc, bcs=case.bcs, pattern=pattern, fast=True, out_vals=out_a)
        t1 = time.perf_counter()
        po.assemble_vector(L, mpc, b=out_b, fast=True)
        t2 = time.perf_counter()
        # reduce-scatter: worker r adds segment r of every private array (the stand-in for the
        # off-process stash / ghost update; the reference only ships interface rows)
        done.wait()
        ra = np.frombuffer(A_sum, dtype=np.float64)[seg_a[r]:seg_a[r + 1]]
        rb = np.frombuffer(b_sum, dtype=np.float64)[seg_b[r]:seg_b[r + 1]]
        ra[:] = 0.0
        rb[:] = 0.0
        for q in range(P):
            ra += np.frombuffer(vals[q], dtype=np.float64)[seg_a[r]:seg_a[r + 1]]
            rb += np.frombuffer(vecs[q], dtype=np.float64)[seg_b[r]:seg_b[r + 1]]
        t3 = time.perf_counter()
        times[3 * r], times[3 * r + 1], times[3 * r + 2] = t1 - t0, t2 - t1, t3 - t2

    procs = [ctx.Process(target=work, args=(r,)) for r in range(P)]
    for p in procs:
        p.start()
    go.wait()
    t0 = time.perf_counter()
    for p in procs:
        p.join()
    total = time.perf_counter() - t0
    if any(p.exitcode != 0 for p in procs):
        raise RuntimeError("a worker failed")
    tm = np.frombuffer(times, dtype=np.float64).reshape(P, 3)
    b = np.frombuffer(b_sum, dtype=np.float64)
    # sanity: the sum over slabs is the global right-hand side
    if P <= 8 and N <= 32:
        ref = po.assemble_vector(case.L, mpc, fast=True)
        assert np.allclose(b, ref, rtol=1e-12, atol=1e-14 * abs(ref).max())
    print(json.dumps({
        "value": ndofs / total, "unit": "DoFs/s", "cores": P, "kind": "port",
        "sample": f"same workload at N={N} ({ncells} cells, {ndofs} dofs) on {P} forked workers (cell slabs, private "
                  f"value arrays, reduce-scatter by the workers): slowest matrix {tm[:, 0].max():.2f}s, vector "
                  f"{tm[:, 1].max():.2f}s, reduction incl. wait {tm[:, 2].max():.2f}s, wall {total:.2f}s",
        "t_wall_s": total,
    }))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 96, int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1))
