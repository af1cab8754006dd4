#!/usr/bin/env python
"""clusters per row block of the owner-computes cluster vector plan of config 2 (is the kernel's tail a few large blocks?)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch

    import bench
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd.la import create_vector

    av = sys.modules["dolfinx_mpc_amd.assemble_vector"]
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    args = argparse.Namespace(n=N, no_tile=False, tile=[8, 8, 8], scaling="strong", numbering="tiled", cell="tet", ufcx=None)
    w = bench.poisson_workload(args, 0, 1, 1)
    lv, Lf, mv = w.vectors[0]
    b = create_vector(w.V)
    dm.assemble_vector(Lf, mv, b=b)
    torch.cuda.synchronize()
    va, kv = av.vector_args(Lf, 0, b, mv, 0)
    nb = int(va.plan.num_blocks)
    own = [v for k, od in w.mesh._device.items() if k == ("objcache", "vcube_own") for v in od.values()][0]
    own = own[1] if isinstance(own, tuple) and len(own) == 2 else own
    print("kernel", va.kernel_name, "blocks", nb, "max rows (own + halo)", int(va.plan.max_rows))
    def tensors(o):
        if isinstance(o, torch.Tensor):
            yield o
        elif isinstance(o, (list, tuple)):
            for x in o:
                yield from tensors(x)
        elif isinstance(o, dict):
            for x in o.values():
                yield from tensors(x)

    for i, t in enumerate(tensors(own)):
        if isinstance(t, torch.Tensor) and t.dim() == 1 and t.numel() == nb + 1 and t.dtype == torch.int64:
            d = (t[1:] - t[:-1]).cpu().numpy()
            print(f"  int64[{nb + 1}] array #{i}: per-block count min {d.min()} mean {d.mean():.1f} p99 {np.percentile(d, 99):.0f} max {d.max()}"
                  f"  (blocks above 1.5 x mean: {(d > 1.5 * d.mean()).sum()})")


if __name__ == "__main__":
    main()
