"""BASELINE config 5's vector path at a size where the plans matter (P2 periodic Poisson, default 128^3 cubes =
17 M dofs; MPCX_FULLSIZE_P2_N to change): size-independent properties of the owner-computes vector kernel
(vector_ownblock_kernel + vector_spill_reduce_kernel) -- it must agree with the halo-recomputing row-block kernel, and
the entries of b must add up to the integral of f (the P2 basis is a partition of unity; the periodic constraint moves
slave entries to their masters with coefficient 1)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = int(os.environ.get("MPCX_FULLSIZE_P2_N", 128))


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
