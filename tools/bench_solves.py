"""The solves that follow the assembly in the reference's drivers, on one GPU.  One JSON line.
    python tools/bench_solves.py stokes [n]    Taylor-Hood channel (python/tests/test_stokes_channelflow.py) on n^3 cubes:
                                               MINRES, additive field split (V-cycle on a00, Jacobi on the pressure mass)
    python tools/bench_solves.py contact [n]   config 4 (bench_contact_3D.py:287-330): CG + V-cycle with the translations
                                               alone and with ``A.setNearNullSpace(rigid_motions_nullspace(V))``"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import dolfinx_mpc_amd as dm  # noqa: E402
from dolfinx_mpc_amd import fem  # noqa: E402
from dolfinx_mpc_amd.problem import LinearProblem  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "stokes"
n = int(sys.argv[2]) if len(sys.argv) > 2 else (16 if what == "stokes" else 20)
rtol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-8
out = {"problem": what, "n": n, "rtol": rtol}
if what == "stokes":
    from dolfinx_mpc_amd.mesh import create_unit_cube

    mesh = create_unit_cube(n, n, n, reorder=(4, 4, 4))
    V = fem.functionspace(mesh, ("Lagrange", 2, (3,)))
    Q = fem.functionspace(mesh, ("Lagrange", 1))
    walls = fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[1], 0) | np.isclose(x[1], 1))
    bc = fem.dirichletbc(np.zeros(3), walls, V)
    ind = lambda x: np.isclose(x[0], 1) | np.isclose(x[2], 1)  # noqa: E731

    def rel(x):
        o = x.copy()
        o[0][np.isclose(x[0], 1)] -= 1
        o[2][np.isclose(x[2], 1)] -= 1
        return o

    mu = dm.MultiPointConstraint(V)
    mu.create_periodic_constraint_geometrical(V, ind, rel, [bc])
    mu.finalize()
    mp = dm.MultiPointConstraint(Q)
    mp.create_periodic_constraint_geometrical(Q, ind, rel, [])
    mp.finalize()
    a = [[fem.form_stiffness(V), fem.form_div_test(V, Q, constant=-1.0)], [fem.form_div_trial(Q, V, constant=-1.0), None]]
    L = [fem.form_source(V, fem.FN_CONSTANT_VEC, constant=np.array([1.0, 1.0, 0.0, 0.0])), None]
    out.update(dofs_V=V.num_dofs, dofs_Q=Q.num_dofs)
    for tag, P in (("mass_P", [[None, None], [None, fem.form_mass(Q)]]), ("no_P", None)):
        prob = LinearProblem(a, L, [mu, mp], bcs=[bc], P=P, solver_options={"rtol": rtol, "max_it": 5000})
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        uh, ph = prob.solve()
        torch.cuda.synchronize()
        x = V.tabulate_dof_coordinates()
        err = float(abs(uh.x.array[0::3] - 0.5 * x[:, 1] * (1.0 - x[:, 1])).max())
        out[tag] = dict(prob.info, total_s=time.perf_counter() - t0, velocity_error=err, pressure_ptp=float(np.ptp(ph.x.array)))
else:
    from dolfinx_mpc_amd.utils import rigid_motions_nullspace

    args = argparse.Namespace(n=n, no_tile=False, tile=[8, 8, 8], scaling="strong")
    w = bench.contact_workload(args, 0, 1)
    (_, a, (mpc, _)), (_, L, _) = w.blocks[0], w.vectors[0]
    out.update(dofs=w.V.num_dofs, slaves=int(mpc.slaves.size))
    sols = {}
    for tag in ("translations", "rigid_body_modes"):
        prob = LinearProblem(a, L, mpc, w.bcs, solver_options={"rtol": rtol, "pc_type": "gamg", "max_it": 1000})
        if tag == "rigid_body_modes":
            prob.A.setNearNullSpace(rigid_motions_nullspace(w.V))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        u = prob.solve()
        torch.cuda.synchronize()
        out[tag] = dict(prob.info, total_s=time.perf_counter() - t0)
        sols[tag] = u.x.array.copy()
    out["solution_difference"] = float(abs(sols["translations"] - sols["rigid_body_modes"]).max() / abs(sols["translations"]).max())
print(json.dumps(out))
