#!/usr/bin/env python
"""bench.py -- headline benchmark of the constrained-assembly hot path.

Workload (BASELINE.json configs[1]; python/benchmarks/bench_periodic.py:35-110):
periodic-BC Poisson, P1 tets on the N^3 unit cube (N=256 by default:
100 663 296 cells, 16 974 593 dofs, 65 025 slaves), fp64.

One "step" = assemble_matrix (into the cached MPC-pattern CSR) + assemble_vector,
inputs resident in HBM.  metric = Ndof / (t_matrix + t_vector).  apply_lifting is
timed separately (as the reference's timers do).

    python bench.py --gpus N --steps K --warmup W

N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling, every
rank assembles its own N^3 box of a (N, N, N*world) mesh and exchanges the
partial sums of the interface-plane rows with its z-neighbours over RCCL.
Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def log(*a):
    if int(os.environ.get("RANK", 0)) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def build_problem(N: int, reorder, rank=0, world=1):
    """mesh, space, bc, constraint, forms of the periodic Poisson benchmark.
    world > 1: rank's z-slab of the (N, N, N*world) mesh on [0,1]^2 x [0,world]."""
    from dolfinx_mpc_amd import MultiPointConstraint, fem
    from dolfinx_mpc_amd.distributed import create_slab_mesh
    from dolfinx_mpc_amd.mesh import create_box

    t = time.time()
    if world == 1:
        mesh = create_box((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), (N, N, N), "tetrahedron", reorder)
    else:
        mesh = create_slab_mesh(N, rank, world, reorder)
    zmax = float(world)
    V = fem.functionspace(mesh, ("Lagrange", 1))
    log(f"mesh: {mesh.num_cells} cells, {V.num_dofs} dofs ({time.time() - t:.1f}s)")

    t = time.time()

    def dirichletboundary(x):  # bench_periodic.py:49-55 (global walls y,z in {0,1})
        return np.isclose(x[1], 0) | np.isclose(x[1], 1) | np.isclose(x[2], 0) | np.isclose(x[2], zmax)

    bdofs = fem.locate_dofs_geometrical(V, dirichletboundary)
    bc = fem.dirichletbc(0.0, bdofs, V)
    mpc = MultiPointConstraint(V)

    def periodic_relation(x):  # bench_periodic.py:63-68
        out = np.zeros(x.shape)
        out[0] = 1 - x[0]
        out[1] = x[1]
        out[2] = x[2]
        return out

    mpc.create_periodic_constraint_geometrical(V, lambda x: np.isclose(x[0], 1), periodic_relation, [bc])
    mpc.finalize()
    log(f"constraint: {mpc.slaves.size} slaves ({time.time() - t:.1f}s)")
    a = fem.form_stiffness(V)
    L = fem.form_source(V, fem.FN_BENCH_PERIODIC)
    return mesh, V, bc, mpc, a, L


def measured_traffic(path: str, kernel_substr: str, N: int):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC
    passes (FETCH_SIZE and WRITE_SIZE in separate runs, gfx950 correction
    2*FETCH_SIZE, see tools/collect_pmc.py); None if not collected for this N."""
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None
    if d.get("_workload_n") != N:
        return None
    for name, c in d.items():
        if isinstance(c, dict) and kernel_substr in name and "hbm_bytes_per_launch" in c:
            return int(c["hbm_bytes_per_launch"])
    return None


def cpu_baseline(sample_n: int):
    """The oracle (C restatement of the reference's serial loops) timed on one
    host core on a bounded sample: the same workload at N = sample_n."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import pyoracle as po
    from problems import case_cube_periodic, oracle_mpc

    case = case_cube_periodic(sample_n, 1, 0.0)
    mpc = oracle_mpc(po, case)
    pattern = po.create_pattern(case.a, mpc, mpc)
    po.assemble_matrix(case.a, mpc, bcs=case.bcs, pattern=pattern, fast=True)  # warm
    t0 = time.perf_counter()
    po.assemble_matrix(case.a, mpc, bcs=case.bcs, pattern=pattern, fast=True)
    t1 = time.perf_counter()
    po.assemble_vector(case.L, mpc, fast=True)
    t2 = time.perf_counter()
    ndofs = case.V.num_dofs
    return {
        "value": ndofs / (t2 - t0),
        "unit": "DoFs/s",
        "cores": 1,
        "kind": "port",
        "sample": f"same workload at N={sample_n} ({case.mesh.num_cells} cells, {ndofs} dofs): "
                  f"matrix {t1 - t0:.2f}s + vector {t2 - t1:.2f}s, oracle/mpc_oracle.c -O3, 1 thread",
        "t_matrix_s": t1 - t0,
        "t_vector_s": t2 - t1,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=int(os.environ.get("MPCX_BENCH_N", 256)))
    ap.add_argument("--alg", default=os.environ.get("MPCX_MATRIX_ALG", "rowblock"))
    ap.add_argument("--tile", type=int, nargs=3, default=[8, 8, 8], help="node/cell tile of the numbering")
    ap.add_argument("--no-tile", action="store_true")
    ap.add_argument("--cpu-sample-n", type=int, default=96)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-allcores", type=int, default=0, metavar="P",
                    help="additionally time the oracle on P forked workers (oracle/cpu_parallel.py); off by default")
    ap.add_argument("--setup-only", action="store_true", help="host set-up only (no GPU), for timing the plan")
    ap.add_argument("--pmc-json", default=os.path.join(ROOT, "profiles", "pmc_latest.json"),
                    help="rocprofv3 PMC summary (tools/collect_pmc.py) of the same workload: source of roofline.traffic")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    N = args.n
    reorder = None if args.no_tile else tuple(args.tile)

    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import _native
    am = sys.modules["dolfinx_mpc_amd.assemble_matrix"]  # the module (the package re-exports the function)

    # one rank = one GPU: select it before anything touches the device (the sparsity pattern is
    # built there when a GPU is present)
    import torch

    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank % torch.cuda.device_count())

    t_setup = time.time()
    mesh, V, bc, mpc, a, L = build_problem(N, reorder, rank, world)
    t = time.time()
    rowptr, cols = dm.create_sparsity_pattern(a, mpc, where="host" if args.setup_only else None)
    log(f"pattern: nnz {cols.size} ({time.time() - t:.1f}s)")
    if args.setup_only:
        log(f"host set-up total {time.time() - t_setup:.1f}s")
        return

    import torch
    import torch.distributed as dist

    # MPCX_DIST_BACKEND=gloo lets several ranks share one GPU (smoke test of the N>1 path
    # on a 1-GPU box); the real runs use RCCL ("nccl"), one rank per GPU
    backend = os.environ.get("MPCX_DIST_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    from dolfinx_mpc_amd.la import MPCMatrix, create_vector

    A = MPCMatrix(rowptr, cols, V.num_dofs)
    b = create_vector(V)
    bcs = [bc]

    exchange = None
    if world > 1:
        from dolfinx_mpc_amd.distributed import SlabExchange

        exchange = SlabExchange(mesh, rowptr, cols, rank, world, device=torch.device("cuda", dev_index))

    # the interface rows of the matrix travel while the vector kernel runs: post the transfer after the
    # matrix assembly, finish it (wait + add) after the vector assembly
    pending = []

    def step_matrix():
        dm.assemble_matrix(a, mpc, bcs=bcs, A=A, algorithm=args.alg)
        if exchange is not None:
            pending.append(exchange.reduce_matrix_begin(A))

    def step_vector():
        dm.assemble_vector(L, mpc, b=b)
        if exchange is not None:
            pending.append(exchange.reduce_vector_begin(b))
            for h in pending:
                exchange.finish(h)
            pending.clear()

    t = time.time()
    step_matrix()
    step_vector()
    torch.cuda.synchronize()
    log(f"first step incl. plan build + uploads: {time.time() - t:.1f}s; host set-up total {time.time() - t_setup:.1f}s")

    for _ in range(args.warmup):
        step_matrix()
        step_vector()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the timed region: exactly K steps --------------------------------
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        step_matrix()
        ev[k][1].record()
        step_vector()
        ev[k][2].record()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    t_mat = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    t_vec = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))

    # ---- per-kernel timing of the dominant (bulk matrix) kernel, HIP events on
    #      the launch stream ----------------------------------------------------
    Lib = _native.lib()
    alg_id = am._ALG[args.alg]
    margs, _keep = am.matrix_args(a, 0, A, mpc, mpc, bcs, alg_id, store_mode=1 if alg_id == 2 else 0,
                                  with_mpc_kernel=False)
    reps = max(args.steps, 5)
    kev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in kev:
        s.record()
        _native.check(Lib.mpcx_assemble_matrix(C.byref(margs)), "mpcx_assemble_matrix")
        e.record()
    torch.cuda.synchronize()
    t_bulk = float(np.mean([s.elapsed_time(e) for s, e in kev]))
    # lifting, separately
    lev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
    for s, e in lev:
        s.record()
        dm.apply_lifting(b, [a], [bcs], mpc)
        e.record()
    torch.cuda.synchronize()
    t_lift = float(np.mean([s.elapsed_time(e) for s, e in lev[1:]]))
    # restore a consistent A/b
    step_matrix()
    step_vector()
    torch.cuda.synchronize()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # achievable HBM bandwidth on this device: device-to-device copy of 2 GiB, read + write
    probe = torch.empty(1 << 28, dtype=torch.float64, device="cuda")
    probe2 = torch.empty_like(probe)
    probe2.copy_(probe)
    pe = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    pe[0].record()
    for _ in range(5):
        probe2.copy_(probe)
    pe[1].record()
    torch.cuda.synchronize()
    copy_gbs = 5 * 2 * probe.numel() * 8 / (pe[0].elapsed_time(pe[1]) * 1e-3) / 1e9
    del probe, probe2

    ndofs_rank = V.num_dofs
    # global dof count of the (N, N, N*world) mesh: interface planes counted once
    ndofs_total = (N + 1) ** 2 * (N * world + 1)
    nc, nv, nd = mesh.num_owned_cells, 4, 4
    alg_bytes = 4 * nv * nc + 4 * nd * nc + 24 * mesh.num_nodes + 8 * cols.size + 2 * V.num_dofs
    peak = 8000.0  # GB/s, HBM3E spec (MI355X_MICROARCH.md)
    achieved = alg_bytes / (t_bulk * 1e-3) / 1e9
    out = {
        "metric": "assembled DoFs/sec (matrix+vector), periodic Poisson P1 256^3",
        "value": ndofs_total * args.steps / elapsed,
        "unit": "DoFs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"periodic-BC Poisson, P1 tets, {N}^3 unit cube per GPU, fp64 (BASELINE configs[1])",
            "cells_per_gpu": int(nc),
            "dofs_per_gpu": int(ndofs_rank),
            "slaves_per_gpu": int(mpc.slaves.size),
            "nnz_per_gpu": int(cols.size),
            "matrix_algorithm": args.alg,
            "numbering_tile": None if reorder is None else list(reorder),
            "parallelism": f"z-slab x{world}" if world > 1 else "single GPU",
        },
        "timings_ms": {"assemble_matrix": t_mat, "assemble_vector": t_vec, "apply_lifting": t_lift,
                       "matrix_bulk_kernel": t_bulk},
        "roofline": {
            "kernel": f"matrix_{args.alg}_kernel<P1 tet stiffness>",
            "bound": "hbm",
            "achieved": achieved,
            "peak": peak,
            "unit": "GB/s",
            "frac": achieved / peak,
            "traffic": measured_traffic(args.pmc_json, f"matrix_{args.alg}_kernel", N),
            "algorithmic_bytes": int(alg_bytes),
            "launch_ms": t_bulk,
            "copy_probe_GBs": copy_gbs,  # what a plain device copy reaches on this box
            "frac_of_copy_probe": achieved / copy_gbs,
        },
    }
    # second kernel of the step, both bounds: compulsory bytes B_b over the assemble_vector
    # time, and the fp64 arithmetic of the 14-point source loop (82 flop per point in the ISA:
    # 35 fma/fmac, 9 mul, 3 add) against the fp64 vector peak
    nq = int(L.integrals[0].kernel.qwts.size)
    vec_bytes = 4 * nv * nc + 4 * nd * nc + 24 * mesh.num_nodes + 9 * V.num_dofs
    out["roofline_vector"] = {
        "kernel": "vector_kernel<P1 tet source, f of bench_periodic.py>",
        "hbm": {"achieved": vec_bytes / (t_vec * 1e-3) / 1e9, "peak": peak, "unit": "GB/s",
                "frac": vec_bytes / (t_vec * 1e-3) / 1e9 / peak, "algorithmic_bytes": int(vec_bytes)},
        "fp64_valu": {"achieved": 82.0 * nq * nc / (t_vec * 1e-3) / 1e12, "peak": 78.6, "unit": "TFLOP/s",
                      "frac": 82.0 * nq * nc / (t_vec * 1e-3) / 1e12 / 78.6, "quadrature_points": nq},
        "launch_ms": t_vec,
        "note": "longer than the matrix kernel; neither bound is reached: see DESIGN.md section 5",
    }
    if not args.no_cpu_baseline and world == 1:  # reported on rank 0 at N=1 only
        log("timing the CPU baseline (oracle, 1 core) ...")
        out["cpu_baseline"] = cpu_baseline(args.cpu_sample_n)
        if args.cpu_allcores > 0:
            # the way the reference is deployed: one serial loop per MPI rank over its cells (SURVEY 8d ii);
            # fresh interpreter so that nothing forks with a live HIP runtime
            import subprocess

            r = subprocess.run([sys.executable, "-m", "oracle.cpu_parallel", str(args.cpu_sample_n), str(args.cpu_allcores)],
                               cwd=ROOT, capture_output=True, text=True, timeout=900)
            if r.returncode == 0:
                out["cpu_baseline_allcores"] = json.loads(r.stdout.strip().splitlines()[-1])
            else:
                log("oracle.cpu_parallel failed: " + r.stderr[-300:])
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
