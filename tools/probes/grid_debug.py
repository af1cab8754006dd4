import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import dolfinx_mpc_amd as dm
from problems import case_cube_periodic
from conftest import *  # noqa
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
case = case_cube_periodic(N, 1, 0.0, reorder=(4, 4, 4))
from test_gpu_parity import product_mpc
mpc = product_mpc(case)
b = dm.assemble_vector(case.L, mpc).numpy().copy()
os.environ["MPCX_TENSOR_GRID"] = "0"
b2 = dm.assemble_vector(case.L, mpc).numpy().copy()
print("N", N, "max diff", abs(b - b2).max(), abs(b2).max())
