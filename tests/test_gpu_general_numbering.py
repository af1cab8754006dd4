"""The HIP path on arbitrarily numbered meshes.  The kernels make no assumption about
the numbering -- a shuffled mesh only costs speed (every row block is touched by cells from everywhere) -- and
``reorder_spatial`` gives the locality back that the generators' tile-wise numbering has."""
import numpy as np
import pytest

from problems import (case_contact_two_body, case_cube_elasticity_slip, case_cube_periodic, oracle_outputs, product_mpc,
                      product_outputs)

pytestmark = pytest.mark.gpu

MAKERS = [lambda nb: case_cube_periodic(4, 1, 0.0, numbering=nb), lambda nb: case_cube_periodic(3, 2, 0.0, numbering=nb),
          lambda nb: case_cube_elasticity_slip(3, numbering=nb), lambda nb: case_contact_two_body(2, 3, 0.4, numbering=nb)]


@pytest.mark.parametrize("alg", ["atomic", "rowblock"])
@pytest.mark.parametrize("numbering", ["shuffled", "spatial"])
@pytest.mark.parametrize("make", MAKERS, ids=["p1", "p2", "elasticity", "contact"])
def test_renumbered_meshes_match_oracle(oracle, make, numbering, alg):
    case = make(numbering)
    ref = oracle_outputs(oracle, case)
    out = product_outputs(case, algorithm=alg)
    assert np.array_equal(out["A"].indptr, ref["A"].indptr) and np.array_equal(out["A"].indices, ref["A"].indices)
    scale = max(1.0, abs(ref["A"].data).max())
    assert abs(out["A"].data - ref["A"].data).max() <= 1e-12 * scale
    for k in ("b", "b_lifted"):
        if k in ref:
            assert abs(out[k] - ref[k]).max() <= 1e-12 * max(1.0, abs(ref[k]).max())


@pytest.mark.parametrize("numbering", ["shuffled", "spatial"])
def test_cluster_kernels_run_on_renumbered_meshes(oracle, numbering):
    """the headline kernels (cell clusters) off the generator's own cell order: nodes renumbered,
    cells shuffled, local vertices of every cell permuted -- all cells end up in clusters, the cluster plan is the
    one that runs, and matrix / vector / lifting match the oracle on the same mesh"""
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import fem
    from dolfinx_mpc_amd.clusters import mesh_clusters_device
    from dolfinx_mpc_amd.mesh import Mesh
    from problems import Case, _walls_yz, periodic_raw

    base = case_cube_periodic(6, 1, 0.0, numbering=numbering).mesh
    rng = np.random.default_rng(11)
    cells = base.geometry.dofmap.copy()
    for c in range(cells.shape[0]):
        cells[c] = cells[c][rng.permutation(4)]
    mesh = Mesh(base.geometry.x, cells, "tetrahedron")
    mesh.node_tile_offsets = base.node_tile_offsets
    V = fem.functionspace(mesh, ("Lagrange", 1))
    bc = fem.dirichletbc(0.3, fem.locate_dofs_geometrical(V, _walls_yz), V)
    case = Case("clusters_" + numbering, V, fem.form_stiffness(V), fem.form_source(V, fem.FN_BENCH_PERIODIC), [bc],
                periodic_raw(V, [bc]))
    d_verts, left = mesh_clusters_device(mesh, mesh.num_cells)
    assert left.size == 0 and d_verts.shape[0] * 6 == mesh.num_cells
    ref = oracle_outputs(oracle, case)
    mpc = product_mpc(case)
    A = dm.assemble_matrix(case.a, mpc, bcs=case.bcs)
    assert ("objcache", "cubes") in A._plans
    S = A.to_scipy()
    assert np.array_equal(S.indptr, ref["A"].indptr) and np.array_equal(S.indices, ref["A"].indices)
    assert abs(S.data - ref["A"].data).max() <= 1e-12 * max(1.0, abs(ref["A"].data).max())
    b = dm.assemble_vector(case.L, mpc)
    assert abs(b.numpy() - ref["b"]).max() <= 1e-12 * max(1.0, abs(ref["b"]).max())
    dm.apply_lifting(b, [case.a], [case.bcs], mpc)
    assert abs(b.numpy() - ref["b_lifted"]).max() <= 1e-12 * max(1.0, abs(ref["b_lifted"]).max())


@pytest.mark.parametrize("degree", [1, 2])
def test_reorder_spatial_shrinks_the_row_block_plan(degree):
    """entities evaluated per row block (the halo the row-block kernels pay for): a shuffled 16^3 mesh makes nearly
    every cell touch nd different blocks; after reorder_spatial the plan is within 2x of the generator's own
    tile-wise numbering"""
    import dolfinx_mpc_amd as dm

    def plan_entities(case):
        mpc = product_mpc(case)
        A = dm.assemble_matrix(case.a, mpc, bcs=case.bcs, algorithm="rowblock")
        info = [p[1][2] for k, od in A._plans.items() if k == ("objcache", "rowblock") for p in od.values()]
        assert info, "row-block plan expected"
        return info[0]["num_ents"], case.V.mesh.num_cells

    n = 16 if degree == 1 else 10
    import os

    os.environ["MPCX_NO_CUBE"] = "1"  # this test compares the per-cell plans
    os.environ["MPCX_FORCE_KERNEL"] = "matrix=rowblock"  # (P2: the entity lists of the per-cell row blocks, not the pair records)
    try:
        shuffled, nc = plan_entities(case_cube_periodic(n, degree, 0.0, numbering="shuffled"))
        spatial, _ = plan_entities(case_cube_periodic(n, degree, 0.0, numbering="spatial"))
        tiled, _ = plan_entities(case_cube_periodic(n, degree, 0.0, reorder=(8, 8, 8)))
    finally:
        del os.environ["MPCX_NO_CUBE"]
        del os.environ["MPCX_FORCE_KERNEL"]
    assert spatial < 0.75 * shuffled, (shuffled, spatial, tiled, nc)
    assert spatial <= 2.0 * tiled, (shuffled, spatial, tiled, nc)
