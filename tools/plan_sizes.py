import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import torch, numpy as np
import bench
class A: pass
a = A(); a.n = 56; a.no_tile = False; a.tile = [8,8,8]; a.scaling = "strong"
w = bench.contact_workload(a, 0, 1)
import dolfinx_mpc_amd as dm
label, f, (m0, m1) = w.blocks[0]
Am = dm.assemble_matrix(f, (m0, m1), bcs=w.bcs, algorithm="rowblock")
for k, od in Am._plans.items():
    if k[1] == "mpc_plan_dev":
        for objs, val in od.values():
            tgt, off, ent, pq, coef, has = val
            print("targets", tgt.numel(), "tuples", ent.numel(), "max per target", int((off[1:]-off[:-1]).max()), "mean", ent.numel()/tgt.numel())
            # distinct entities per target
am = sys.modules["dolfinx_mpc_amd.assemble_matrix"]
print("slave entities", am._slave_entities(f, 0, m0, m1)[0].size, "cells", w.mesh.num_cells)
