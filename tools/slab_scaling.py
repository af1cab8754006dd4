"""Strong-scaling prediction from ONE GPU: every slab a rank would hold when config 2 (P1 Poisson
256^3) or config 5 is cut into 2 / 4 / 8 slabs is built and assembled alone on this GPU -- kernel time per step, host
cost per step (plain calls and HIP-graph replay), pack / add kernels of the interface rows -- and the per-step time of
the N-GPU job is predicted as  max over ranks (kernels)  +  exchange,  exchange = pack + bytes / 153 GB/s (one xGMI
link per neighbour pair, /opt/skills/guides/MI355X_MICROARCH.md) + add, overlapped with the vector kernel for the matrix
rows (dolfinx_mpc_amd/distributed.py posts them before the vector assembly).

    python tools/slab_scaling.py [config] [N] > profiles/r04_slab_scaling_config2.json"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import dolfinx_mpc_amd as dm  # noqa: E402
from dolfinx_mpc_amd import _native  # noqa: E402
from dolfinx_mpc_amd.graph import CapturedStep  # noqa: E402
from dolfinx_mpc_amd.la import create_vector  # noqa: E402

config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else (256 if config == 2 else 246)
LINK_GBS, LINK_LAT_US = 153.0, 10.0
L = _native.lib()
out = {"config": config, "N": N, "link_GBs": LINK_GBS, "link_latency_us": LINK_LAT_US, "worlds": {}}
ONE = None
if "--one" in sys.argv:  # child mode: one rank of one cut in a fresh process (as the N-GPU job runs it), one JSON line
    k = sys.argv.index("--one")
    ONE = (int(sys.argv[k + 1]), int(sys.argv[k + 2]))
WORLDS = tuple(int(v) for v in os.environ.get("MPCX_SLAB_WORLDS", "1,2,4,8").split(","))  # (config 5 at 384^3: "8" -- one GPU cannot hold it)
for world in ((ONE[0],) if ONE else WORLDS):
    ranks = []
    for rank in ((ONE[1],) if ONE else range(world)):
        if ONE is None:
            import subprocess

            run = subprocess.run([sys.executable, os.path.abspath(__file__), str(config), str(N), "--one", str(world), str(rank)],
                                 capture_output=True, text=True)
            line = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
            if run.returncode != 0 or not line:
                raise SystemExit(run.stderr[-2000:])
            ranks.append(json.loads(line[-1]))
            print(f"world {world} rank {rank}: {ranks[-1]}", file=sys.stderr, flush=True)
            continue
        args = argparse.Namespace(n=N, no_tile=False, tile=[8, 8, 8], scaling="strong", numbering="tiled", cell="tet", ufcx=None)
        w = bench.poisson_workload(args, rank, world, 1 if config == 2 else 2)
        label, f, (m0, m1) = w.blocks[0]
        lv, fv, mv = w.vectors[0]
        A = dm.create_matrix(f, m0, m1)
        b = create_vector(mv.function_space)

        def step():
            dm.assemble_matrix(f, (m0, m1), bcs=w.bcs, A=A)
            dm.assemble_vector(fv, mv, b=b)

        step()
        torch.cuda.synchronize()
        t_m = bench.hip_time(lambda: dm.assemble_matrix(f, (m0, m1), bcs=w.bcs, A=A), 5)
        t_v = bench.hip_time(lambda: dm.assemble_vector(fv, mv, b=b), 5)

        def timed(fn, n=50):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            th = time.perf_counter() - t0
            torch.cuda.synchronize()
            return th / n * 1e6, (time.perf_counter() - t0) / n * 1e6

        h_plain, w_plain = min((timed(step) for _ in range(3)), key=lambda hw: hw[1])  # (bench.py issues plain calls: the best of three)
        g = CapturedStep(step)
        h_graph, w_graph = timed(g.replay)
        # interface rows this rank sends up (one node plane; P2: + the edge dofs of the plane) and their entries
        V = w.V
        n_plane = (N + 1) ** 2 if config == 2 else (2 * N + 1) ** 2
        rows_if = n_plane if rank < world - 1 else 0
        nnz_row = A.nnz / max(V.num_dofs, 1)
        n_ent = int(rows_if * nnz_row)
        t_pack = t_add = 0.0
        if n_ent > 0:
            idx = torch.randint(0, A.nnz, (n_ent,), dtype=torch.int64, device="cuda").sort().values
            buf = torch.empty(n_ent, dtype=torch.float64, device="cuda")
            t_pack = bench.hip_time(lambda: _native.check(L.mpcx_gather_f64(A.vals.data_ptr(), idx.data_ptr(), n_ent, buf.data_ptr(), None), "g"), 5)
            t_add = bench.hip_time(lambda: _native.check(L.mpcx_scatter_add_f64(A.vals.data_ptr(), idx.data_ptr(), n_ent, buf.data_ptr(), None), "s"), 5)
        ranks.append({"rank": rank, "cells": int(w.mesh.num_owned_cells), "dofs": int(V.num_dofs), "matrix_call_ms": t_m, "vector_call_ms": t_v,
                      "step_wall_us_plain": w_plain, "step_wall_us_graph": w_graph, "host_us_plain": h_plain, "host_us_graph": h_graph,
                      "interface_matrix_bytes": n_ent * 8, "interface_vector_bytes": rows_if * 8, "pack_ms": t_pack, "add_ms": t_add})
        print(json.dumps(ranks[-1]), flush=True)
        sys.exit(0)
    # plain calls keep the two library streams concurrent; the HIP-graph replay is reported beside it (it costs the
    # host 35-45 us instead of 200 but serialises more of the step: slower wall time on these small slabs)
    slow = max(min(r["step_wall_us_plain"], r["step_wall_us_graph"]) for r in ranks)
    ex_m = max((r["pack_ms"] + r["add_ms"]) * 1e3 + r["interface_matrix_bytes"] / (LINK_GBS * 1e3) + LINK_LAT_US for r in ranks) if world > 1 else 0.0
    ex_v = max(r["interface_vector_bytes"] / (LINK_GBS * 1e3) + LINK_LAT_US for r in ranks) if world > 1 else 0.0
    # the matrix rows travel while the vector kernel runs: only what exceeds the vector call is exposed
    exposed = max(0.0, ex_m - min(r["vector_call_ms"] for r in ranks) * 1e3) + ex_v
    ndofs = (N + 1) ** 3 if config == 2 else (2 * N + 1) ** 3
    out["worlds"][world] = {"ranks": ranks, "slowest_rank_step_us": slow, "exchange_matrix_us": ex_m, "exchange_vector_us": ex_v,
                            "predicted_step_us": slow + exposed, "predicted_DoFs_per_s": ndofs / ((slow + exposed) * 1e-6)}
if 1 in out["worlds"]:
    base = out["worlds"][1]["predicted_step_us"]
    for world, d in out["worlds"].items():
        d["predicted_speedup"] = base / d["predicted_step_us"]
        d["predicted_efficiency"] = d["predicted_speedup"] / world
print(json.dumps(out, indent=1))
