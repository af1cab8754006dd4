"""config 2: how long does the step take in the first batches of a fresh process?  (batches of 5 steps, one sync per batch)"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import dolfinx_mpc_amd as dm
from dolfinx_mpc_amd.la import create_vector
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
args = argparse.Namespace(n=N, no_tile=False, tile=[8, 8, 8], scaling="strong", numbering="tiled", cell="tet", ufcx=None)
w = bench.poisson_workload(args, 0, 1, 1)
label, f, (m0, m1) = w.blocks[0]
lv, fv, mv = w.vectors[0]
A = dm.create_matrix(f, m0, m1)
b = create_vector(mv.function_space)
def step():
    dm.assemble_matrix(f, (m0, m1), bcs=w.bcs, A=A)
    dm.assemble_vector(fv, mv, b=b)
step(); torch.cuda.synchronize()
if os.environ.get("BUSY"):
    # an unrelated load first: is the transient the device's (clocks) or the library's (first uses)?
    z = torch.rand(8192, 8192, device="cuda")
    for _ in range(int(os.environ["BUSY"])):
        z = (z @ z).clamp_(0, 1)
    torch.cuda.synchronize()
out, host, mallocs = [], [], []
for batch in range(10):
    m0_ = torch.cuda.memory_stats().get("num_device_alloc", 0)
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    out.append(round((time.perf_counter() - t0) / 5 * 1e3, 3))
    host.append(round(th / 5 * 1e3, 3))
    mallocs.append(torch.cuda.memory_stats().get("num_device_alloc", 0) - m0_)
print("ms per step, batches of 5:", out)
print("host ms per step:", host, "device allocations per batch:", mallocs)
