#!/usr/bin/env python
"""Where the FIRST assembly call of a configuration spends its time: runs bench.py's set-up,
then the first step under cProfile with blocking launches (AMD_SERIALIZE_KERNEL=3), so that host-side cumulative
times include the device work each Python caller launched.

    python tools/first_call_probe.py [--config 2] [--size 256] > gpurun_out/first_call.txt
"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

os.environ.setdefault("AMD_SERIALIZE_KERNEL", "3")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--size", dest="n", type=int, default=256)
    ap.add_argument("--top", type=int, default=45)
    a = ap.parse_args()
    import torch

    import bench
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd.la import create_vector

    args = argparse.Namespace(n=a.n, no_tile=False, tile=[8, 8, 8], numbering="tiled", ufcx=None, cell="tet", scaling="strong",
                              config=a.config, alg="rowblock")
    torch.zeros(1, device="cuda")
    w = bench.poisson_workload(args, 0, 1, 1 if a.config == 2 else 2)
    t = time.time()
    mats = {label: dm.create_matrix(f, m0, m1) for label, f, (m0, m1) in w.blocks}
    vecs = {label: create_vector(m.function_space) for label, _f, m in w.vectors}
    torch.cuda.synchronize()
    print(f"pattern {time.time() - t:.3f}s")

    def step():
        for label, f, (m0, m1) in w.blocks:
            dm.assemble_matrix(f, (m0, m1), bcs=w.bcs, A=mats[label])
        for label, f, m in w.vectors:
            dm.assemble_vector(f, m, b=vecs[label])
        torch.cuda.synchronize()

    pr = cProfile.Profile()
    t = time.time()
    pr.enable()
    step()
    pr.disable()
    print(f"first step {time.time() - t:.3f}s (blocking launches)")
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(a.top)
    print(s.getvalue())
    t = time.time()
    step()
    print(f"second step {time.time() - t:.4f}s")


if __name__ == "__main__":
    main()
