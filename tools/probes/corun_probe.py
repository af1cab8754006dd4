#!/usr/bin/env python
"""Do the HBM-bound matrix kernels and the VALU-bound vector kernels of a step run side by side when the matrix launch
is capped (dolfinx_mpc_amd/corun.py)?  One process, one problem set-up, a list of arms; every arm = environment settings
read per call + module constants of the vector plan, timed like bench.py (K steps between two synchronisations) and, for
the split, with each call alone.

    python tools/probes/corun_probe.py --config 5 [--size 246] [--arms name,name ...] [--steps 10]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


ARMS = {
    # name: (env, vector owner rows or None)
    "off": ({"MPCX_CORUN": "0"}, None),
    "w2_f60": ({"MPCX_CORUN": "1", "MPCX_CORUN_MATRIX_WGS": "2", "MPCX_CORUN_FRAC": "0.6"}, 4096),
    "w2_f100": ({"MPCX_CORUN": "1", "MPCX_CORUN_MATRIX_WGS": "2", "MPCX_CORUN_FRAC": "0.999"}, 4096),
    "w2_f40": ({"MPCX_CORUN": "1", "MPCX_CORUN_MATRIX_WGS": "2", "MPCX_CORUN_FRAC": "0.4"}, 4096),
    "w2_f80": ({"MPCX_CORUN": "1", "MPCX_CORUN_MATRIX_WGS": "2", "MPCX_CORUN_FRAC": "0.8"}, 4096),
    "w3_f60": ({"MPCX_CORUN": "1", "MPCX_CORUN_MATRIX_WGS": "3", "MPCX_CORUN_FRAC": "0.6"}, 3072),
    "w3_f100": ({"MPCX_CORUN": "1", "MPCX_CORUN_MATRIX_WGS": "3", "MPCX_CORUN_FRAC": "0.999"}, 3072),
    "w1_f60": ({"MPCX_CORUN": "1", "MPCX_CORUN_MATRIX_WGS": "1", "MPCX_CORUN_FRAC": "0.6"}, 8192),
    "w1_f100": ({"MPCX_CORUN": "1", "MPCX_CORUN_MATRIX_WGS": "1", "MPCX_CORUN_FRAC": "0.999"}, 8192),
    "w2_f60_v": ({"MPCX_CORUN": "1", "MPCX_CORUN_MATRIX_WGS": "2", "MPCX_CORUN_FRAC": "0.6", "MPCX_CORUN_VECTOR_FLOOR": "52000"}, 4096),
    "w2_f100_v": ({"MPCX_CORUN": "1", "MPCX_CORUN_MATRIX_WGS": "2", "MPCX_CORUN_FRAC": "0.999", "MPCX_CORUN_VECTOR_FLOOR": "52000"}, 4096),
    "off_v4096": ({"MPCX_CORUN": "0"}, 4096),
    "w2_f60_t256": ({"MPCX_CORUN": "1", "MPCX_CORUN_MATRIX_WGS": "2", "MPCX_CORUN_FRAC": "0.6", "MPCX_VECTOR_THREADS": "256"}, 4096),
    "w2_f100_t256": ({"MPCX_CORUN": "1", "MPCX_CORUN_MATRIX_WGS": "2", "MPCX_CORUN_FRAC": "0.999", "MPCX_VECTOR_THREADS": "256"}, 4096),
}
ALL_KEYS = sorted({k for env, _ in ARMS.values() for k in env})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=5)
    ap.add_argument("--size", dest="n", type=int, default=0)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--arms", default="")
    ap.add_argument("--generic", action="store_true", help="config 2 with the cluster kernels off")
    pa = ap.parse_args()
    if pa.generic:
        os.environ["MPCX_NO_CUBE"] = "1"
    import numpy as np  # noqa: F401
    import torch

    import bench
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import corun
    from dolfinx_mpc_amd.la import create_vector, wait_assembly

    av = sys.modules["dolfinx_mpc_amd.assemble_vector"]
    args = argparse.Namespace(n=pa.n or {2: 256, 3: 128, 4: 56, 5: 246}[pa.config], no_tile=False, tile=[8, 8, 8], scaling="strong",
                              cell="tet", numbering="tiled", ufcx=None, config=pa.config)
    t = time.time()
    if pa.config in (2, 5):
        w = bench.poisson_workload(args, 0, 1, 1 if pa.config == 2 else 2)
    elif pa.config == 3:
        w = bench.stokes_workload(args, 0, 1)
    else:
        w = bench.contact_workload(args, 0, 1)
    mats = {label: dm.create_matrix(f, m0, m1) for label, f, (m0, m1) in w.blocks}
    torch.cuda.synchronize()
    print(f"# set-up {time.time() - t:.1f} s; {w.config['workload']}", flush=True)
    names = [a for a in pa.arms.split(",") if a] or list(ARMS)
    base_rows = av.VECTOR_OWNER_ROWS
    for name in names:
        env, vrows = ARMS[name]
        for k in ALL_KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        av.VECTOR_OWNER_ROWS = vrows or base_rows
        vforms = list(w.vectors)  # (the owner plan is cached per (form, rows): no fresh form needed)
        vecs = {label: create_vector(m.function_space) for label, _f, m in vforms}

        def mat():
            for label, f, (m0, m1) in w.blocks:
                dm.assemble_matrix(f, (m0, m1), bcs=w.bcs, A=mats[label], algorithm="rowblock")

        def vec():
            for label, f, m in vforms:
                dm.assemble_vector(f, m, b=vecs[label])

        def step():
            mat()
            vec()

        def timed(fn, reps):
            wait_assembly()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            wait_assembly()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3

        def synced(fn, reps):
            tt = 0.0
            for _ in range(reps):
                tt += timed(fn, 1)
            return tt / reps

        try:
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            out = {"arm": name, "step_pipelined_ms": timed(step, pa.steps), "step_synced_ms": synced(step, 5),
                   "matrix_alone_ms": timed(mat, 5), "vector_alone_ms": timed(vec, 5)}
            # LDS of the vector plan (own + halo rows)
            try:
                a, _k = av.vector_args(vforms[0][1], 0, vecs[vforms[0][0]], vforms[0][2], 0)
                out["vector_lds"] = int(a.plan.max_rows) * 8
                out["vector_kernel"] = a.kernel_name
            except Exception as e:  # noqa: BLE001
                out["vector_lds"] = str(e)[:80]
            out["params"] = corun.params() if env.get("MPCX_CORUN") != "0" else None
        except Exception as e:  # noqa: BLE001
            out = {"arm": name, "error": str(e)[:300]}
        print("ARM " + json.dumps(out), flush=True)
        del vecs, vforms


if __name__ == "__main__":
    main()
