"""config 3's momentum right-hand side alone: which kernel, which plan, how long (env switches from the command line)"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import dolfinx_mpc_amd as dm  # noqa: E402
import importlib  # noqa: E402

av = importlib.import_module("dolfinx_mpc_amd.assemble_vector")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
args = argparse.Namespace(n=n, no_tile=False, tile=[8, 8, 8], scaling="strong")
w = bench.stokes_workload(args, 0, 1)
name, L0, mv = w.vectors[0]
b = dm.assemble_vector(L0, mv)
a, keep = av.vector_args(L0, 0, b, mv, 0)
k = L0.integrals[0].kernel
print("kernel", a.kernel_name, "nq", int(k.qwts.size), "fn", k.fn_id, "blocks", a.plan.num_blocks, "max_rows", a.plan.max_rows,
      "lds KB", a.plan.max_rows * 8 / 1024, "cells", L0.integrals[0].num_entities, "cells/block", L0.integrals[0].num_entities / max(a.plan.num_blocks, 1))
torch.cuda.synchronize()
for _ in range(3):
    dm.assemble_vector(L0, mv, b=b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    dm.assemble_vector(L0, mv, b=b)
torch.cuda.synchronize()
print("assemble_vector call ms", (time.perf_counter() - t0) / 20 * 1e3)
