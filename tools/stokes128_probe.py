"""Step-by-step timing of the config-3 set-up and assembly at full size (debug aid)."""
import faulthandler, os, sys, time
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import dolfinx_mpc_amd as dm
from problems import stokes_slip_problem
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
t = time.time()
def lap(msg):
    global t
    torch.cuda.synchronize()
    print(f"[{time.time()-t:7.1f}s] {msg}  (gpu mem {torch.cuda.memory_allocated()/2**30:.1f} GiB)", flush=True)
    t = time.time()
V, Q, bcs, raw_v, forms, L0 = stokes_slip_problem(3, N, reorder=(8, 8, 8))
lap(f"problem: cells {V.mesh.num_cells} dofsV {V.num_dofs} dofsQ {Q.num_dofs} slaves {raw_v[0].size}")
mv = dm.MultiPointConstraint(V); mv.add_constraint(V, *raw_v); mv.finalize()
mq = dm.MultiPointConstraint(Q); mq.finalize()
lap("constraints finalized")
mpcs = [mv, mq]
for (i, j), f in forms.items():
    A = dm.create_matrix(f, mpcs[i], mpcs[j])
    lap(f"a{i}{j} pattern nnz {A.nnz}")
    dm.assemble_matrix(f, (mpcs[i], mpcs[j]), bcs=bcs, A=A, algorithm="rowblock")
    lap(f"a{i}{j} first assembly")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(3):
        dm.assemble_matrix(f, (mpcs[i], mpcs[j]), bcs=bcs, A=A, algorithm="rowblock")
    ev[1].record(); torch.cuda.synchronize()
    print(f"   a{i}{j} steady {ev[0].elapsed_time(ev[1])/3:.2f} ms", flush=True)
    del A
    torch.cuda.empty_cache()
