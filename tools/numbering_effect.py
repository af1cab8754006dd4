"""What the numbering costs the row-block path: periodic P1 / P2 Poisson on a cube, generator's tile-wise numbering
vs a randomly shuffled mesh vs the shuffled mesh after reorder_spatial.   python tools/numbering_effect.py [N] [degree]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dolfinx_mpc_amd as dm  # noqa: E402
from problems import case_cube_periodic, product_mpc  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
degree = int(sys.argv[2]) if len(sys.argv) > 2 else 1


def timed(f, reps=5):
    f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


for label, kw in (("tiled (generator)", dict(reorder=(8, 8, 8))), ("shuffled", dict(numbering="shuffled")),
                  ("shuffled + reorder_spatial", dict(numbering="spatial"))):
    case = case_cube_periodic(N, degree, 0.0, **kw)
    mpc = product_mpc(case)
    try:
        A = dm.assemble_matrix(case.a, mpc, bcs=case.bcs)
        b = dm.assemble_vector(case.L, mpc)
        tm = timed(lambda: dm.assemble_matrix(case.a, mpc, bcs=case.bcs, A=A))
        tv = timed(lambda: dm.assemble_vector(case.L, mpc, b=b))
        info = [p[1][2] for k, od in A._plans.items() if k in (("objcache", "rowblock"), ("objcache", "cubes")) for p in od.values()]
        ents = info[0].get("num_ents", info[0].get("num_slots")) if info else None
        print(f"{label:28s} N={N} P{degree}: matrix {tm:7.3f} ms  vector {tv:7.3f} ms  plan entries {ents}  cells {case.V.mesh.num_cells}")
    except RuntimeError as e:
        print(f"{label:28s} failed: {e}")
