"""BASELINE config 4 at FULL size on one GPU: two stacked cubes (57^3 over 114^3 cubes, 10 M tets,
1.72 M nodes, 5.15 M dofs of vector P1), inelastic contact (cpp/ContactConstraint.h:908-1174), linear
elasticity with E = 1e3, nu = 0 (python/benchmarks/bench_contact_3D.py:62-270, --no-slip, theta = 0).
The oracle needs minutes at this size, so the checks are size-independent properties:

1. device atomics and LDS row blocks agree;
2. slave and Dirichlet rows AND columns hold exactly `diagval` on the diagonal;
3. symmetry of K^T A K:  x^T A y == y^T A x;
4. exactness: the benchmark's data (bottom clamped, top pushed down by 0.425, nu = 0, no body force)
   has the uniaxial solution u = (0, 0, -0.2125 z), which vector P1 represents exactly, is continuous
   across the interface (so it satisfies u_s = u_m) and -- the interface grids being nested 2:1 -- makes
   the tractions of the two bodies cancel in the master rows:  (A_mpc u)_i == (apply_lifting(0))_i on
   every free row.  This ties the elasticity kernel, the multi-master elimination and the lifting together;
5. b (constant body force f) sums to f * volume = 2 f per component, slaves carry nothing.

MPCX_CONTACT_N0 overrides the resolution of the top cube (default 57)."""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N0 = int(os.environ.get("MPCX_CONTACT_N0", 57))


@pytest.fixture(scope="module")
def problem():
    import torch

    import dolfinx_mpc_amd as dm
    from problems import contact_problem

    mesh, ft, V, bcs, a, L, (sm, mm) = contact_problem(N0, None, 0.0, reorder=(8, 8, 8), body_force=(0.5, -0.25, -1.0))
    mpc = dm.MultiPointConstraint(V)
    mpc.create_contact_inelastic_condition(ft, sm, mm)
    mpc.finalize()
    if N0 == 57:
        assert mesh.num_cells == 6 * (57**3 + 114**3) and V.num_dofs == 3 * (58**3 + 115**3) == 5_147_961
    assert mpc.num_local_slaves == 3 * (2 * N0 + 1) ** 2
    A = dm.assemble_matrix(a, mpc, bcs=bcs, algorithm="rowblock")
    torch.cuda.synchronize()
    return dict(mesh=mesh, V=V, bcs=bcs, mpc=mpc, a=a, L=L, A=A)


def _rowids(A):
    import torch

    counts = (A.d_rowptr[1:] - A.d_rowptr[:-1]).to(torch.int64)
    return torch.repeat_interleave(torch.arange(A.shape[0], device=A.device), counts)


def _spmv(A, rowid, x, absolute=False):
    import torch

    y = torch.zeros(A.shape[0], dtype=torch.float64, device=A.device)
    v = A.vals.abs() if absolute else A.vals
    y.index_add_(0, rowid, v * x[A.d_cols.to(torch.int64)])
    return y


def _marked(p, dev):
    import torch

    m = torch.zeros(p["V"].num_dofs, dtype=torch.bool, device=dev)
    m[torch.from_numpy(p["mpc"].slaves.astype(np.int64)).to(dev)] = True
    for bc in p["bcs"]:
        m[torch.from_numpy(bc.dof_indices()[0].astype(np.int64)).to(dev)] = True
    return m


def test_constraint_shape(problem):
    """<= 3 masters per slave, same-component masters, weights sum to one (nested grids: 1 or 2 masters)"""
    mpc = problem["mpc"]
    off = mpc.masters.offsets
    n = np.diff(off)[mpc.slaves]
    assert n.min() >= 1 and n.max() <= 3
    first = off[mpc.slaves]
    assert np.all(mpc.masters.array[first] % 3 == mpc.slaves % 3)
    sums = np.add.reduceat(mpc.coefficients()[0], first)
    assert np.allclose(sums, 1.0, rtol=0, atol=1e-12)


def test_atomic_and_rowblock_agree(problem):
    import torch

    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd.la import MPCMatrix

    p = problem
    B = MPCMatrix(p["A"].rowptr, p["A"].cols, p["V"].num_dofs)
    dm.assemble_matrix(p["a"], p["mpc"], bcs=p["bcs"], A=B, algorithm="atomic")
    scale = float(p["A"].vals.abs().max())
    diff = float((p["A"].vals - B.vals).abs().max())
    assert diff <= 1e-12 * scale, (diff, scale)
    del B
    torch.cuda.empty_cache()


def test_constrained_rows_and_columns_are_identity(problem):
    p = problem
    A = p["A"]
    marked = _marked(p, A.device)
    rowid = _rowids(A)
    cols = A.d_cols.long()
    diag = rowid == cols
    off = (marked[rowid] | marked[cols]) & ~diag
    assert float(A.vals[off].abs().max()) == 0.0
    d = A.vals[diag & marked[rowid]]
    assert d.numel() == int(marked.sum()) and bool((d == 1.0).all())


def test_symmetry(problem):
    import torch

    A = problem["A"]
    gen = torch.Generator(device=A.device).manual_seed(11)
    x = torch.rand(A.shape[0], dtype=torch.float64, device=A.device, generator=gen) - 0.5
    y = torch.rand(A.shape[0], dtype=torch.float64, device=A.device, generator=gen) - 0.5
    rowid = _rowids(A)
    xAy = float(torch.dot(x, _spmv(A, rowid, y)))
    yAx = float(torch.dot(y, _spmv(A, rowid, x)))
    scale = float(torch.dot(x.abs(), _spmv(A, rowid, y.abs(), absolute=True)))
    assert abs(xAy - yAx) <= 1e-12 * scale


def test_uniaxial_solution_is_reproduced_through_contact_and_lifting(problem):
    import torch

    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd.la import create_vector

    p = problem
    A, V, mpc, bcs = p["A"], p["V"], p["mpc"], p["bcs"]
    b = create_vector(V)
    dm.apply_lifting(b, [p["a"]], [bcs], mpc)  # b = -K^T A g
    x = V.tabulate_dof_coordinates()
    u_h = np.zeros(V.num_dofs)
    u_h[2::3] = -0.2125 * x[:, 2]
    u = torch.from_numpy(u_h).to(A.device)
    rowid = _rowids(A)
    r = _spmv(A, rowid, u) - b.array
    free = ~_marked(p, A.device)
    scale = float(_spmv(A, rowid, u.abs(), absolute=True).max())
    assert float(r[free].abs().max()) <= 1e-11 * scale, (float(r[free].abs().max()), scale)


def test_body_force_vector(problem):
    import torch

    import dolfinx_mpc_amd as dm

    p = problem
    b = dm.assemble_vector(p["L"], p["mpc"])
    dev = b.array.device
    sl = torch.from_numpy(p["mpc"].slaves.astype(np.int64)).to(dev)
    assert float(b.array[sl].abs().max()) == 0.0
    # basis functions sum to one and the contact weights of every slave sum to one: nothing is lost
    sums = b.array.view(-1, 3).sum(dim=0).cpu().numpy()
    assert np.allclose(sums, 2.0 * np.array([0.5, -0.25, -1.0]), rtol=1e-11, atol=0)
