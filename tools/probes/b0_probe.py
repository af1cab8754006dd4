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

# Experiment (round 5): the order of the entities INSIDE a block.  Consecutive lanes take consecutive cells -- the six
# tets of a cube, which share nodes -- so an LDS add of local node i hits the same address from several lanes (35 LDS
# cycles per instruction, 40 % conflicts).  A strided order inside every block spreads neighbouring cells over different
# waves (own_lmap is indexed by entity, so the order inside a block is free).
if os.environ.get("B0_PERMUTE"):
    import ctypes as C
    import numpy as np

    stride = int(os.environ["B0_PERMUTE"])
    nb = int(a.plan.num_blocks)
    off = torch.empty(nb + 1, dtype=torch.int64, device="cuda")
    C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(off.data_ptr()), C.c_void_p(a.plan.block_ent_off), C.c_size_t(8 * (nb + 1)), 3)
    offh = off.cpu().numpy()
    total = int(offh[-1])
    ents = torch.empty(total, dtype=torch.int32, device="cuda")
    C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(ents.data_ptr()), C.c_void_p(a.plan.block_ents), C.c_size_t(4 * total), 3)
    pos = torch.arange(total, device="cuda", dtype=torch.int64)
    blk = torch.searchsorted(off, pos, right=True) - 1
    start = off[blk]
    length = (off[blk + 1] - start)
    local = pos - start
    # position p of a block of length n takes the entity at (p * stride) mod n' (n' = n rounded down to a multiple that keeps
    # the map a bijection: use the transposed-matrix order instead -- rows of `stride` entities read column by column)
    ncol = (length + stride - 1) // stride
    src_local = (local % ncol) * stride + local // ncol
    ok = src_local < length
    # (ragged last column: fall back to identity for the positions that would leave the block)
    key = torch.where(ok, src_local, local)
    # make it a bijection: sort positions of each block by key (stable)
    order = torch.argsort(blk * (int(length.max()) + stride + 1) * 2 + key, stable=True)
    new = ents[order]
    C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(a.plan.block_ents), C.c_void_p(new.data_ptr()), C.c_size_t(4 * total), 3)
    torch.cuda.synchronize()
    ref = b.array.clone()
    for _ in range(3):
        dm.assemble_vector(L0, mv, b=b)
    torch.cuda.synchronize()
    print("permuted (stride %d): max diff vs before %.3e" % (stride, float((b.array - ref).abs().max() / ref.abs().max())))
    t0 = time.perf_counter()
    for _ in range(20):
        dm.assemble_vector(L0, mv, b=b)
    torch.cuda.synchronize()
    print("assemble_vector call ms (permuted)", (time.perf_counter() - t0) / 20 * 1e3)
