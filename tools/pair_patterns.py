"""How many distinct scatter-offset patterns do the pair records of a P2 box mesh hold?  (decides whether a dictionary
of patterns can replace the ten offset bytes of every record)  usage: python tools/pair_patterns.py N [N ...]"""
import importlib
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dolfinx_mpc_amd as dm  # noqa: E402
from problems import case_cube_periodic, product_mpc  # noqa: E402

am = importlib.import_module("dolfinx_mpc_amd.assemble_matrix")
for n in [int(a) for a in sys.argv[1:]]:
    case = case_cube_periodic(n, 2, 0.0, reorder=(8, 8, 8))
    mpc = product_mpc(case)
    A = dm.create_matrix(case.a, mpc)
    os.environ["MPCX_FORCE_KERNEL"] = "matrix=pairs"
    args, keep = am.matrix_args(case.a, 0, A, mpc, mpc, case.bcs, 2)
    assert args.kernel_name == "pairs"
    recs = [k for k in keep if isinstance(k, tuple) and len(k) == 3 and hasattr(k[2], "view")][0][2].view(-1, 4)
    w1 = (recs[:, 1].to(torch.int64) >> 16) & 0xffff
    key = torch.stack([w1, recs[:, 2].to(torch.int64) & 0xffffffff, recs[:, 3].to(torch.int64) & 0xffffffff], 1)
    uniq = torch.unique(key, dim=0)
    i = (recs[:, 0].to(torch.int64) >> 27) & 15
    per_i = [int(torch.unique(key[i == k], dim=0).shape[0]) for k in range(10)]
    print(f"N={n}: pairs {recs.shape[0]}, distinct offset patterns {uniq.shape[0]}, per local row {per_i}", flush=True)
