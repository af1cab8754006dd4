"""TEST INFRASTRUCTURE (oracle): the CPU restatement run on many host cores, the way the
reference is deployed (one serial assembly loop per MPI rank over its share of the cells,
python/benchmarks/Makefile `mpirun -n 23`).

The cells are split into P contiguous slabs.  P threads (the oracle's C loops are called through
ctypes, which releases the GIL) each assemble their slab into a PRIVATE array that covers only the
contiguous range of rows their cells touch -- the analogue of an MPI rank's local matrix -- and then
every thread adds the private rows of its ownership range (its own and its neighbours' overlapping
interface rows) into the global arrays: the stand-in for PETSc's off-process stash and
`b.ghostUpdate(ADD, REVERSE)`.  Timed: wall time from the common start to the last thread's end.
Only bench.py's cpu_baseline leg uses this.

    python -m oracle.cpu_parallel N P [degree]  ->  one JSON line
"""

from __future__ import annotations

import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def assemble_allcores(V, a_of_cells, L_of_cells, mpc, bcs, pattern, P: int, check=None):
    """a_of_cells(cells) / L_of_cells(cells): the forms restricted to a cell range.  Returns
    (wall seconds, per-thread (matrix, vector, reduce) seconds, A values, b)."""
    from oracle import pyoracle as po

    rowptr = np.ascontiguousarray(pattern[0], dtype=np.int64)
    ncells = V.mesh.num_owned_cells
    bs = V.dofmap.bs
    bounds = np.linspace(0, ncells, P + 1).astype(np.int64)
    dm = V.dofmap.list
    # rows touched by slab r: its cells' dofs and the masters of the slaves among them
    lo, hi = np.empty(P, dtype=np.int64), np.empty(P, dtype=np.int64)
    moff, mast = mpc.masters_offsets, mpc.masters
    lo_cells = np.empty(P, dtype=np.int64)
    for r in range(P):
        d = dm[bounds[r]:bounds[r + 1]]
        lo[r], hi[r] = int(d.min()) * bs, (int(d.max()) + 1) * bs
        lo_cells[r] = lo[r]
        rows = np.unique(d) * bs
        sl = (rows[:, None] + np.arange(bs)[None, :]).reshape(-1)
        sl = sl[mpc.is_slave[sl] != 0]
        if sl.size:
            m = np.concatenate([mast[moff[s]:moff[s + 1]] for s in sl])
            if m.size:
                lo[r], hi[r] = min(lo[r], int(m.min())), max(hi[r], int(m.max()) + 1)
    # ownership boundaries follow the cells' own rows (master rows elsewhere only widen the private range)
    own = np.concatenate([[0], lo_cells[1:], [rowptr.size - 1]]).astype(np.int64)
    if np.any(np.diff(own) < 0):
        raise RuntimeError("cell slabs do not touch increasing row ranges: numbering not slab-ordered")
    nnz, n = int(rowptr[-1]), rowptr.size - 1
    A_glob, b_glob = np.zeros(nnz), np.zeros(n)
    A_glob.fill(0.0)
    priv = [None] * P
    forms = [(a_of_cells(np.arange(bounds[r], bounds[r + 1], dtype=np.int32)),
              L_of_cells(np.arange(bounds[r], bounds[r + 1], dtype=np.int32))) for r in range(P)]
    go, mid = threading.Barrier(P + 1), threading.Barrier(P)
    times = np.zeros((P, 3))
    errors = []

    def work(r):
        try:
            a, L = forms[r]
            # the local matrix is allocated and touched before the clock starts, like a preallocated PETSc Mat
            v = np.empty(int(rowptr[hi[r]] - rowptr[lo[r]]))
            w = np.empty(int(hi[r] - lo[r]))
            v.fill(0.0)
            w.fill(0.0)
            priv[r] = (v, w)
            go.wait()
            t0 = time.perf_counter()
            po.assemble_matrix(a, mpc, bcs=bcs, pattern=pattern, fast=True,
                               raw_vals=v.ctypes.data - 8 * int(rowptr[lo[r]]))
            t1 = time.perf_counter()
            po.assemble_vector(L, mpc, fast=True, raw_b=w.ctypes.data - 8 * int(lo[r]))
            t2 = time.perf_counter()
            mid.wait()
            # rows [own[r], own[r+1]): mine, plus whatever the other slabs hold for them
            r0, r1 = int(own[r]), int(own[r + 1])
            A_glob[rowptr[r0]:rowptr[r1]] = 0.0
            b_glob[r0:r1] = 0.0
            for q in range(P):
                s0, s1 = max(r0, int(lo[q])), min(r1, int(hi[q]))
                if s0 < s1:
                    vq, wq = priv[q]
                    A_glob[rowptr[s0]:rowptr[s1]] += vq[rowptr[s0] - rowptr[lo[q]]:rowptr[s1] - rowptr[lo[q]]]
                    b_glob[s0:s1] += wq[s0 - lo[q]:s1 - lo[q]]
            t3 = time.perf_counter()
            times[r] = (t1 - t0, t2 - t1, t3 - t2)
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            try:
                mid.abort()
            except Exception:  # noqa: BLE001
                pass

    threads = [threading.Thread(target=work, args=(r,)) for r in range(P)]
    for t in threads:
        t.start()
    go.wait()
    t0 = time.perf_counter()
    for t in threads:
        t.join()
    wall = time.perf_counter() - t0
    if errors:
        raise errors[0]
    return wall, times, A_glob, b_glob


def host_pattern(form, case, cases=None):
    """MPC sparsity pattern from the product's threaded C++ host builder (identical to the oracle's numpy
    restatement, tests/test_host_setup.py, which needs minutes at these sizes); set-up, not timed."""
    import dolfinx_mpc_amd as dm
    from problems import product_mpc

    mpcs = [product_mpc(c) for c in (cases or [case])]
    rp, cols = dm.create_sparsity_pattern(form, mpcs[0] if len(mpcs) == 1 else (mpcs[0], mpcs[1]), where="host")
    return rp.astype(np.int32), cols


def main(N: int, P: int, degree: int = 1, kind: str = "poisson"):
    from dolfinx_mpc_amd import fem
    from oracle import pyoracle as po
    from problems import case_contact_two_body, case_cube_periodic, oracle_mpc

    if kind == "contact":
        # two-body contact elasticity (config 4): N^3 cubes over (2N)^3; the slab that holds the slave cells also
        # holds their masters' rows in the other body (a wider private matrix, like the ghost rows of an MPI rank)
        case = case_contact_two_body(N)
        degree = 1
    else:
        case = case_cube_periodic(N, degree, 0.0)
    V = case.V
    mpc = oracle_mpc(po, case)
    pattern = host_pattern(case.a, case)
    fn = case.L.integrals[0].kernel.fn_id
    if kind == "contact":
        mu, lam = (float(v) for v in case.a.integrals[0].constants[:2])
        cL = np.array(case.L.integrals[0].constants, dtype=np.float64)
        a_of = lambda c: fem.form_elasticity(V, mu, lam, cells=c)  # noqa: E731
        L_of = lambda c: fem.form_source(V, fn, constant=cL, cells=c)  # noqa: E731
    else:
        a_of = lambda c: fem.form_stiffness(V, cells=c)  # noqa: E731
        L_of = lambda c: fem.form_source(V, fn, cells=c)  # noqa: E731
    wall, tm, A, b = assemble_allcores(V, a_of, L_of, mpc, case.bcs, pattern, P)
    if N <= 32:  # sanity: the slabs add up to the single-thread result (diagonals aside)
        ref_b = po.assemble_vector(case.L, mpc, fast=True)
        assert np.allclose(b, ref_b, rtol=1e-12, atol=1e-14 * abs(ref_b).max())
        ref = po.assemble_matrix(case.a, mpc, bcs=case.bcs, pattern=pattern, fast=True, diagval=0.0)
        assert np.allclose(A, ref.data, rtol=1e-12, atol=1e-14 * abs(ref.data).max())
    ndofs, ncells = V.num_dofs, case.mesh.num_cells
    return {
        "value": ndofs / wall, "unit": "DoFs/s", "cores": P, "kind": "port",
        "sample": f"same workload at N={N} ({kind}, P{degree}, {ncells} cells, {ndofs} dofs) on {P} threads of the oracle's C loops "
                  f"(cell slabs, private local matrices, interface rows added by the owners): slowest matrix "
                  f"{tm[:, 0].max():.2f}s, vector {tm[:, 1].max():.2f}s, reduction {tm[:, 2].max():.2f}s, wall {wall:.2f}s",
        "t_wall_s": wall,
    }


if __name__ == "__main__":
    print(json.dumps(main(int(sys.argv[1]) if len(sys.argv) > 1 else 96,
                          int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1),
                          int(sys.argv[3]) if len(sys.argv) > 3 else 1, sys.argv[4] if len(sys.argv) > 4 else "poisson")))
