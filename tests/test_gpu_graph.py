"""dolfinx_mpc_amd.graph.CapturedStep: a steady-state step captured into a HIP graph replays the same kernels on the same
device arrays -- same A and b as the plain calls -- and refresh() carries changed values in."""

import numpy as np
import pytest

import dolfinx_mpc_amd as dm
from dolfinx_mpc_amd import fem
from dolfinx_mpc_amd.graph import CapturedStep
from problems import case_cube_periodic, oracle_outputs, product_mpc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("degree", [1, 2])
def test_replay_reproduces_the_plain_step(oracle, degree):
    case = case_cube_periodic(6 if degree == 1 else 4, degree, 0.4, reorder=(2, 2, 2))
    mpc = product_mpc(case)
    A = dm.assemble_matrix(case.a, mpc, bcs=case.bcs)
    b = dm.assemble_vector(case.L, mpc)

    def step():
        dm.assemble_matrix(case.a, mpc, bcs=case.bcs, A=A)
        dm.assemble_vector(case.L, mpc, b=b)
        dm.apply_lifting(b, [case.a], [case.bcs], mpc)

    g = CapturedStep(step)
    A.vals.zero_()
    b.array.zero_()
    g.replay()
    ref = oracle_outputs(oracle, case)
    assert abs(A.to_scipy().data - ref["A"].data).max() <= 1e-12 * max(1.0, abs(ref["A"].data).max())
    assert abs(b.numpy() - ref["b_lifted"]).max() <= 1e-12 * max(1.0, abs(ref["b_lifted"]).max())
    for _ in range(3):  # replays do not accumulate
        g.replay()
    assert abs(b.numpy() - ref["b_lifted"]).max() <= 1e-12 * max(1.0, abs(ref["b_lifted"]).max())


def test_refresh_carries_changed_values(oracle):
    from dolfinx_mpc_amd.mesh import create_unit_cube

    mesh = create_unit_cube(5, 5, 5, reorder=(2, 2, 2))
    V = fem.functionspace(mesh, ("Lagrange", 1))
    f = fem.Function(V)
    f.interpolate(lambda x: 1.0 + x[0])
    c = fem.Constant(2.0)
    a = fem.form_stiffness(V, coefficient=f, constant=c)
    mpc = dm.MultiPointConstraint(V)
    mpc.finalize()
    om = oracle.OracleMPC.empty(V)
    A = dm.assemble_matrix(a, mpc)
    g = CapturedStep(lambda: dm.assemble_matrix(a, mpc, A=A))
    g.replay()
    assert abs(A.to_scipy().data - oracle.assemble_matrix(a, om).data).max() < 1e-12
    f.x.array[:] = 3.0 - f.x.array
    c.value[:] = 0.5
    g.replay()  # a replay alone does not look at host values ...
    stale = A.to_scipy().data.copy()
    g.refresh()  # ... refresh() does
    g.replay()
    new = oracle.assemble_matrix(a, om).data
    assert abs(A.to_scipy().data - new).max() < 1e-12 and abs(stale - new).max() > 1e-3
