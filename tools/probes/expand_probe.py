"""Taylor-Hood a00 (config 3): block-scalar launch + mpcx_block_expand against the fused CSR-valued launch.
usage: python tools/probes/expand_probe.py [n]   (env MPCX_EXPAND_WIDE=0/1)"""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
import dolfinx_mpc_amd as dm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
args = types.SimpleNamespace(n=n, no_tile=False, tile=(8, 8, 8), ufcx=None)
w = bench.stokes_workload(args, 0, 1)
label, f, (m0, m1) = w.blocks[0]
A = dm.create_matrix(f, m0, m1)


def timeit(fn, reps=6):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return min(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))


def assemble():
    dm.assemble_matrix(f, (m0, m1), bcs=w.bcs, A=A)
    torch.cuda.current_stream().wait_event(A._ready) if getattr(A, "_ready", None) is not None else None


t_bs = timeit(assemble)
assert A.is_block_scalar


def expand():
    A._compact_stale = True
    A._expand_compact()


_ = A.vals
t_ex = timeit(expand)
ref = A.vals.clone()
print(f"n={n} nnz={A.nnz} block-scalar launch {t_bs:.2f} ms, expand {t_ex:.2f} ms ({A.nnz * 8 / t_ex / 1e9:.2f} TB/s written), "
      f"wide={os.environ.get('MPCX_EXPAND_WIDE', '1')}")
if os.environ.get("MPCX_EXPAND_CHECK"):
    os.environ["MPCX_BLOCK_SCALAR"] = "0"
    B = dm.create_matrix(f, m0, m1)
    dm.assemble_matrix(f, (m0, m1), bcs=w.bcs, A=B)
    print("max |expanded - fused| =", float((B.vals - ref).abs().max()))
