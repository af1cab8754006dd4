"""owner-plan shape of the cluster vector kernel on slabs of the 8-way cut of config 2 (why ranks with a lower neighbour are slower)"""
import argparse, os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import dolfinx_mpc_amd as dm
from dolfinx_mpc_amd.la import create_vector
av = importlib.import_module("dolfinx_mpc_amd.assemble_vector")
for rank in (0, 1):
    args = argparse.Namespace(n=256, no_tile=False, tile=[8, 8, 8], scaling="strong", numbering="tiled", cell="tet", ufcx=None)
    w = bench.poisson_workload(args, rank, 8, 1)
    lv, fv, mv = w.vectors[0]
    V = mv.function_space
    b = create_vector(V)
    a, keep = av.vector_args(fv, 0, b, mv, 0)
    pk = [k for k in keep if isinstance(k, tuple) and len(k) == 9][0]
    row0, off, order, lmap, hoff, spill, src, rows, seg = pk
    per = (off[1:] - off[:-1]).cpu().numpy()
    halo = (hoff[1:] - hoff[:-1]).cpu().numpy()
    r0 = row0.cpu().numpy()
    print("rank", rank, "dofs", V.num_dofs, "owned nodes", V.mesh.num_owned_nodes, "blocks", a.plan.num_blocks, "max_rows", a.plan.max_rows,
          "clusters/block mean %.0f max %d" % (per.mean(), per.max()), "halo/block mean %.0f max %d" % (halo.mean(), halo.max()),
          "rows/block min %d max %d" % (np.diff(r0).min(), np.diff(r0).max()), "blocks with >2x mean clusters", int((per > 2 * per.mean()).sum()),
          "tile hints", None if V.dof_tile_offsets is None else len(V.dof_tile_offsets))
    top = np.argsort(per)[-5:]
    print("   heaviest blocks", [(int(t), int(per[t]), int(halo[t]), int(r0[t]), int(r0[t + 1] - r0[t])) for t in top])
    # matrix cluster plan: parts by record format
    lm, fm, (m0, m1) = w.blocks[0]
    A = dm.create_matrix(fm, m0, m1)
    dm.assemble_matrix(fm, (m0, m1), bcs=w.bcs, A=A)
    for v in A._plans[("objcache", "cubes")].values():
        parts, keep2, info = v[1]
        print("   matrix plan", {k: info[k] for k in ("num_blocks", "num_ents", "max_rows", "max_nnz", "narrow_blocks", "closed_form_blocks")},
              [(int(p[0].num_blocks), p[2], int(p[4][-1].item())) for p in parts])
    nnz_row = np.diff(A.rowptr)
    print("   rows", A.shape[0], "max nnz/row", int(nnz_row.max()), "rows with > 16 entries", int((nnz_row > 16).sum()),
          "of which ghost rows", int((nnz_row[V.mesh.num_owned_nodes:] > 16).sum()), "ghost rows", V.num_dofs - V.mesh.num_owned_nodes)
