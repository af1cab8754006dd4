"""Experiment: matrix and vector assembly on two HIP streams (the matrix kernel waits
on memory, the vector kernel on the VALU -- do they overlap on the chip?).

    python tools/exp_overlap.py [N]
"""

import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
import dolfinx_mpc_amd as dm  # noqa: E402
from dolfinx_mpc_amd.la import MPCMatrix, create_vector  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    mesh, V, bc, mpc, a, L = bench.build_problem(N, (8, 8, 8), 0, 1)
    rowptr, cols = dm.create_sparsity_pattern(a, mpc)
    A = MPCMatrix(rowptr, cols, V.num_dofs)
    b = create_vector(V)
    bcs = [bc]
    s_mat, s_vec = torch.cuda.Stream(), torch.cuda.Stream()

    def serial():
        dm.assemble_matrix(a, mpc, bcs=bcs, A=A, algorithm="rowblock")
        dm.assemble_vector(L, mpc, b=b)

    def overlapped(vec_first=False):
        cur = torch.cuda.current_stream()
        s_mat.wait_stream(cur)
        s_vec.wait_stream(cur)
        if vec_first:
            with torch.cuda.stream(s_vec):
                dm.assemble_vector(L, mpc, b=b)
            with torch.cuda.stream(s_mat):
                dm.assemble_matrix(a, mpc, bcs=bcs, A=A, algorithm="rowblock")
        else:
            with torch.cuda.stream(s_mat):
                dm.assemble_matrix(a, mpc, bcs=bcs, A=A, algorithm="rowblock")
            with torch.cuda.stream(s_vec):
                dm.assemble_vector(L, mpc, b=b)
        cur.wait_stream(s_mat)
        cur.wait_stream(s_vec)

    serial()
    torch.cuda.synchronize()
    ref_A = A.vals.clone()
    ref_b = b.array.clone()
    for name, fn in (("serial", serial), ("overlap mat-first", overlapped),
                     ("overlap vec-first", lambda: overlapped(True)), ("serial", serial)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 20
        for _ in range(K):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        ok = bool(torch.equal(A.vals, ref_A)) if name == "serial" else float((A.vals - ref_A).abs().max())
        print(f"{name:20s} {dt * 1e3:8.3f} ms/step  {V.num_dofs / dt / 1e9:.3f} G DoFs/s  check {ok}", flush=True)


if __name__ == "__main__":
    main()
