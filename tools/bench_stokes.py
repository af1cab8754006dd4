"""Assembly times of the Taylor-Hood Stokes blocks with the slip constraint of tests/test_stokes.py
(config 3's element types) on one GPU.   python tools/bench_stokes.py [n]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import dolfinx_mpc_amd as dm  # noqa: E402
from problems import stokes_slip_problem  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
# tiled numbering (what the benchmark meshes use) unless MPCX_STOKES_UNTILED
V, Q, bcs, raw_v, forms, L0 = stokes_slip_problem(3, n, None if os.environ.get("MPCX_STOKES_UNTILED") else (8, 8, 8))
mv = dm.MultiPointConstraint(V)
mv.add_constraint(V, *raw_v)
mv.finalize()
mq = dm.MultiPointConstraint(Q)
mq.finalize()
mpcs = [mv, mq]
out = {"cells": V.mesh.num_cells, "dofs_V": V.num_dofs, "dofs_Q": Q.num_dofs}
for (i, j), a in forms.items():
    A = dm.create_matrix(a, mpcs[i], mpcs[j])
    for alg in ("rowblock", "atomic"):
        dm.assemble_matrix(a, (mpcs[i], mpcs[j]), bcs=bcs, A=A, algorithm=alg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            dm.assemble_matrix(a, (mpcs[i], mpcs[j]), bcs=bcs, A=A, algorithm=alg)
        torch.cuda.synchronize()
        out[f"a{i}{j} {alg} ms"] = (time.perf_counter() - t0) / 5 * 1e3
    out[f"a{i}{j} nnz"] = int(A.nnz)
    del A
b = dm.assemble_vector(L0, mv)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    dm.assemble_vector(L0, mv, b=b)
torch.cuda.synchronize()
out["b0 ms"] = (time.perf_counter() - t0) / 5 * 1e3
print(json.dumps(out))
