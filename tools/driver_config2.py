"""Config 2 (periodic Poisson, P1, N^3 cubes) assembled by the torch-free C++ driver (examples/mpcx_driver.cpp): the step
time a consumer of the bare C ABI gets.   python tools/driver_config2.py [N] [steps]"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_driver import DRIVER, problem_file, read_bundle  # noqa: E402

from dolfinx_mpc_amd.workloads import case_cube_periodic  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
t0 = time.perf_counter()
case = case_cube_periodic(N, 1, 0.0, reorder=(8, 8, 8))
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    pin, pout = os.path.join(d, "problem.bin"), os.path.join(d, "result.bin")
    problem_file(case, pin)
    t_file = time.perf_counter() - t0
    run = subprocess.run([DRIVER, pin, pout, str(steps)], capture_output=True, text=True)
    print(run.stdout.strip(), file=sys.stderr)
    if run.returncode != 0:
        raise SystemExit(run.stderr)
    res = read_bundle(pout)
t = res["timings"]
n = case.V.num_dofs
print(json.dumps({"N": N, "dofs": n, "nnz": int(res["vals"].size), "steps": steps, "host_setup_s": t[0], "upload_s": t[1], "plans_s": t[2],
                  "ms_per_step": 1e3 * t[3], "DoFs_per_s": n / t[3], "clusters": int(t[4]), "leftover_cells": int(t[5]),
                  "matrix_launches": int(t[6]), "problem_file_s": t_file,
                  "checks": {"sum_vals": float(res["vals"].sum()), "sum_b": float(res["b"].sum())}}))
