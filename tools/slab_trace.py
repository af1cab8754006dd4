"""one slab of an 8-rank cut of config 2, a few steps: for rocprofv3 --kernel-trace --stats"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import dolfinx_mpc_amd as dm
from dolfinx_mpc_amd.la import create_vector
world, rank = int(sys.argv[1]), int(sys.argv[2])
args = argparse.Namespace(n=256, no_tile=False, tile=[8, 8, 8], scaling="strong", numbering="tiled", cell="tet", ufcx=None)
w = bench.poisson_workload(args, rank, world, 1)
label, f, (m0, m1) = w.blocks[0]
lv, fv, mv = w.vectors[0]
A = dm.create_matrix(f, m0, m1)
b = create_vector(mv.function_space)
for _ in range(12):
    dm.assemble_matrix(f, (m0, m1), bcs=w.bcs, A=A)
    dm.assemble_vector(fv, mv, b=b)
torch.cuda.synchronize()
