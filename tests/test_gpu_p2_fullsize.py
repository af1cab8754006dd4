"""BASELINE config 5's vector path at the size bench.py --config 5 runs (P2 periodic Poisson, 246^3 cubes = 119.8 M
dofs; MPCX_FULLSIZE_P2_N to change): size-independent properties of the owner-computes vector kernel
(vector_ownblock_kernel + vector_spill_reduce_kernel) -- it must agree with the halo-recomputing row-block kernel, and
the entries of b must add up to the integral of f (the P2 basis is a partition of unity; the periodic constraint moves
slave entries to their masters with coefficient 1)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = int(os.environ.get("MPCX_FULLSIZE_P2_N", 246))  # the size bench.py --config 5 runs (119.8 M dofs)


@pytest.fixture(scope="module")
def problem():
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import fem
    from dolfinx_mpc_amd.mesh import create_unit_cube

    mesh = create_unit_cube(N, N, N, reorder=(8, 8, 8))
    V = fem.functionspace(mesh, ("Lagrange", 2))
    walls = fem.locate_dofs_geometrical(
        V, lambda x: np.isclose(x[1], 0) | np.isclose(x[1], 1) | np.isclose(x[2], 0) | np.isclose(x[2], 1))
    bc = fem.dirichletbc(0.0, walls, V)
    mpc = dm.MultiPointConstraint(V)

    def rel(x):
        out = x.copy()
        out[0] = 1 - x[0]
        return out

    mpc.create_periodic_constraint_geometrical(V, lambda x: np.isclose(x[0], 1), rel, [bc])
    mpc.finalize()
    return dict(mesh=mesh, V=V, mpc=mpc, L=fem.form_source(V, fem.FN_BENCH_PERIODIC))


def test_owner_computes_vector_matches_halo_kernel(problem, monkeypatch):
    import importlib

    import dolfinx_mpc_amd as dm

    av = importlib.import_module("dolfinx_mpc_amd.assemble_vector")  # (the package re-exports the function of that name)
    p = problem
    monkeypatch.setenv("MPCX_VECTOR_OWNER", "1")
    b1 = dm.assemble_vector(p["L"], p["mpc"])
    args, keep = av.vector_args(p["L"], 0, b1, p["mpc"], 0)
    assert args.own_lmap, "owner-computes plan expected for the 24-point rule"
    del keep
    monkeypatch.setenv("MPCX_VECTOR_OWNER", "0")
    b0 = dm.assemble_vector(p["L"], p["mpc"])
    scale = float(b0.array.abs().max())
    assert float((b1.array - b0.array).abs().max()) <= 1e-12 * scale
    # repeated assembly into the same vector: the spill array is rewritten, not accumulated
    monkeypatch.setenv("MPCX_VECTOR_OWNER", "1")
    dm.assemble_vector(p["L"], p["mpc"], b=b1)
    assert float((b1.array - b0.array).abs().max()) <= 1e-12 * scale


def test_vector_entries_add_up_to_the_integral(problem):
    import torch

    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd.quadrature import make_quadrature

    p = problem
    b = dm.assemble_vector(p["L"], p["mpc"])
    dev = b.array.device
    nq = int(p["L"].integrals[0].kernel.qwts.size)
    q, w = make_quadrature("tetrahedron", 6)
    assert q.shape[0] == nq == 24
    X = torch.from_numpy(q).to(dev)
    W = torch.from_numpy(w).to(dev)
    lam = torch.cat([1 - X.sum(dim=1, keepdim=True), X], dim=1)  # (nq, 4)
    xg = torch.from_numpy(p["mesh"].geometry.x).to(dev)
    cells = torch.from_numpy(p["mesh"].geometry.dofmap).to(dev).to(torch.int64)
    total = 0.0
    chunk = 2_000_000
    for s in range(0, cells.shape[0], chunk):
        c = xg[cells[s: s + chunk]]
        det = torch.linalg.det(c[:, 1:, :] - c[:, :1, :]).abs()
        xq = torch.einsum("qv,mvd->mqd", lam, c)
        f = xq[..., 0] * torch.sin(5.0 * np.pi * xq[..., 1]) + torch.exp(
            -((xq[..., 0] - 0.9) ** 2 + (xq[..., 1] - 0.5) ** 2 + (xq[..., 2] - 0.1) ** 2) / 0.02)
        total += float((f * W[None, :]).sum(dim=1).mul(det).sum())
    got = float(b.array.sum())
    assert abs(got - total) <= 1e-11 * max(1.0, abs(total)), (got, total)


def test_p2_matrix_properties_at_size(problem):
    """BASELINE config 5's MATRIX at a size where the plans matter: matrix_rowblock_kernel<P2> against
    the thread-per-entity atomic kernel on the same pattern, symmetry (x.Ay == y.Ax), identity rows for Dirichlet and
    slave dofs, and zero row sums of the stiffness matrix away from the constrained dofs -- properties that do not
    need an oracle run at this size"""
    import torch

    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import fem
    from dolfinx_mpc_amd.la import MPCMatrix, Vector
    from dolfinx_mpc_amd.problem import spmv

    p = problem
    V, mpc = p["V"], p["mpc"]
    walls = fem.locate_dofs_geometrical(
        V, lambda x: np.isclose(x[1], 0) | np.isclose(x[1], 1) | np.isclose(x[2], 0) | np.isclose(x[2], 1))
    bc = fem.dirichletbc(0.0, walls, V)
    a = fem.form_stiffness(V)
    A = dm.assemble_matrix(a, mpc, bcs=[bc], algorithm="rowblock")
    assert any(("objcache", k) in A._plans for k in ("pairs", "rowblock", "cubes")), "an LDS row-block kernel was expected"
    B = MPCMatrix(A.d_rowptr, A.d_cols, A.shape[1])  # same pattern, second value array
    dm.assemble_matrix(a, mpc, bcs=[bc], A=B, algorithm="atomic")
    amax = float(A.vals.abs().max())
    assert amax > 0
    assert float((A.vals - B.vals).abs().max()) <= 1e-12 * amax
    del B
    n = A.shape[0]
    g = torch.Generator(device=A.device).manual_seed(7)
    x, y = Vector(n), Vector(n)
    x.array.copy_(torch.rand(n, generator=g, device=A.device, dtype=torch.float64) - 0.5)
    y.array.copy_(torch.rand(n, generator=g, device=A.device, dtype=torch.float64) - 0.5)
    Ax, Ay = spmv(A, x), spmv(A, y)
    s1, s2 = float(torch.dot(x.array, Ay.array)), float(torch.dot(y.array, Ax.array))
    assert abs(s1 - s2) <= 1e-11 * float(torch.linalg.vector_norm(x.array) * torch.linalg.vector_norm(Ay.array))
    # Dirichlet and slave rows are identity rows (diagval = 1): (A x)[r] == x[r]
    rows = torch.from_numpy(np.concatenate([bc.dof_indices()[0], mpc.slaves]).astype(np.int64)).to(A.device)
    assert torch.equal(Ax.array[rows], x.array[rows])
    # stiffness: constants lie in the kernel -- rows whose neighbours are all free sum to zero
    one = Vector(n)
    one.array.fill_(1.0)
    r = spmv(A, one).array
    X = torch.from_numpy(V.tabulate_dof_coordinates()).to(A.device)
    h = 2.0 / N
    inner = ((X[:, 0] > h) & (X[:, 0] < 1 - h) & (X[:, 1] > h) & (X[:, 1] < 1 - h) & (X[:, 2] > h) & (X[:, 2] < 1 - h))
    assert int(inner.sum()) > 0.5 * n
    assert float(r[inner].abs().max()) <= 1e-11 * amax


@pytest.mark.parametrize("where", ["interior", "dirichlet_wall"])
def test_sampled_sub_box_against_the_oracle(problem, oracle, where):
    """not only the library's kernels against each other -- rows of the 120 M-dof matrix against the ORACLE on
    a re-meshed 5^3-cube sub-box (tests/problems.py assert_sub_box_rows_match_oracle), in the interior and at a Dirichlet wall"""
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import fem
    from problems import assert_sub_box_rows_match_oracle

    p = problem
    V, mpc = p["V"], p["mpc"]
    walls = fem.locate_dofs_geometrical(
        V, lambda x: np.isclose(x[1], 0) | np.isclose(x[1], 1) | np.isclose(x[2], 0) | np.isclose(x[2], 1))
    bc = fem.dirichletbc(0.0, walls, V)
    A = dm.assemble_matrix(fem.form_stiffness(V), mpc, bcs=[bc])
    nb = 5
    corner = (N // 3, 0 if where == "dirichlet_wall" else N // 2, N // 2 - 2)
    n = assert_sub_box_rows_match_oracle(oracle, p["mesh"], p["mesh"].num_cells, V, A, N, corner, nb, wall_y0=(where == "dirichlet_wall"))
    assert n > 300
