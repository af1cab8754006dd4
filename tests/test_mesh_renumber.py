"""General numberings ("the row-block algorithm depends on the tiled structured
numbering; there is no reordering for a general mesh").  dolfinx_mpc_amd.mesh.renumber / reorder_spatial on the host,
and the oracle's equivariance under renumbering -- an independent check of the oracle itself: the same problem on a
shuffled mesh must give P A P^T and P b."""
import numpy as np
import pytest

from dolfinx_mpc_amd.mesh import create_stacked_cubes, create_unit_cube, facet_vertices, renumber, reorder_spatial
from problems import case_contact_two_body, case_cube_elasticity_slip, case_cube_periodic, oracle_outputs


def _volume(mesh):
    X = mesh.geometry.x[mesh.geometry.dofmap]
    return np.abs(np.linalg.det(X[:, 1:] - X[:, :1])).sum() / 6.0


def _span(mesh):
    """mean distance in the numbering between the nodes of a cell"""
    c = mesh.geometry.dofmap.astype(np.int64)
    return float((c.max(axis=1) - c.min(axis=1)).mean())


def test_renumber_keeps_the_mesh():
    mesh = create_unit_cube(5, 4, 3)
    rng = np.random.default_rng(1)
    m2 = renumber(mesh, rng.permutation(mesh.num_nodes), rng.permutation(mesh.num_cells))
    assert m2.num_nodes == mesh.num_nodes and m2.num_cells == mesh.num_cells
    assert abs(_volume(m2) - 1.0) < 1e-14
    assert m2.exterior_facets().shape == mesh.exterior_facets().shape
    # the same set of cells as point sets
    key = lambda m: np.sort(np.round(m.geometry.x[m.geometry.dofmap].reshape(m.num_cells, -1), 12), axis=0)
    assert np.array_equal(key(m2), key(mesh))


def test_renumber_moves_facet_tags_with_their_cells():
    mesh, ft, _ = create_stacked_cubes(2, None, 0.0, None)
    rng = np.random.default_rng(2)
    m2, ft2 = renumber(mesh, rng.permutation(mesh.num_nodes), rng.permutation(mesh.num_cells), ft)
    assert np.array_equal(ft2.values, ft.values)
    a = np.sort(mesh.geometry.x[facet_vertices(mesh, ft.entities)].reshape(ft.values.size, -1), axis=1)
    b = np.sort(m2.geometry.x[facet_vertices(m2, ft2.entities)].reshape(ft.values.size, -1), axis=1)
    assert np.allclose(a, b, rtol=0, atol=0)


def test_reorder_spatial_restores_locality():
    mesh = create_unit_cube(12, 12, 12)
    rng = np.random.default_rng(3)
    shuffled = renumber(mesh, rng.permutation(mesh.num_nodes), rng.permutation(mesh.num_cells))
    ordered = reorder_spatial(shuffled, tile_nodes=64)
    assert abs(_volume(ordered) - 1.0) < 1e-13
    assert _span(shuffled) > 0.4 * mesh.num_nodes  # random: the nodes of a cell are anywhere
    assert _span(ordered) < 0.12 * mesh.num_nodes  # Z-order: a few tiles apart at most
    assert np.array_equal(ordered.node_tile_offsets, np.arange(0, mesh.num_nodes, 64))
    # cells follow their lowest node
    low = ordered.geometry.dofmap.min(axis=1)
    assert np.all(np.diff(low) >= 0)


@pytest.mark.parametrize("make", [lambda nb: case_cube_periodic(3, 1, 0.0, numbering=nb),
                                  lambda nb: case_cube_periodic(2, 2, 0.0, numbering=nb),
                                  lambda nb: case_cube_elasticity_slip(2, numbering=nb),
                                  lambda nb: case_contact_two_body(2, 3, 0.4, numbering=nb)],
                         ids=["p1", "p2", "elasticity", "contact"])
def test_oracle_is_equivariant_under_renumbering(oracle, make):
    """the reference assembles the same operator whatever the numbering: with P matching dofs by coordinate,
    A' = P A P^T, b' = P b (the slaves are picked geometrically in these cases, so the constraint moves along)"""
    base, other = make(None), make("shuffled")
    ra, rb = oracle_outputs(oracle, base), oracle_outputs(oracle, other)
    def keys(case):
        # dof coordinates + the centroid of the cells round the dof: two bodies in contact have distinct nodes at the
        # same coordinates, told apart by the side their cells lie on
        x = case.V.tabulate_dof_coordinates()
        dm = case.V.dofmap.list
        cc = x[dm].mean(axis=1)
        acc, cnt = np.zeros_like(x), np.zeros(x.shape[0])
        for i in range(dm.shape[1]):
            np.add.at(acc, dm[:, i], cc)
            np.add.at(cnt, dm[:, i], 1.0)
        return np.round(np.concatenate([x, acc / cnt[:, None]], axis=1), 9)

    xa, xb = keys(base), keys(other)
    ka = np.lexsort(xa.T)
    kb = np.lexsort(xb.T)
    assert np.allclose(xa[ka], xb[kb])
    bs = base.V.dofmap.bs
    pa = (ka[:, None] * bs + np.arange(bs)).reshape(-1)  # unrolled dofs in coordinate order
    pb = (kb[:, None] * bs + np.arange(bs)).reshape(-1)
    Aa = ra["A"].toarray()[np.ix_(pa, pa)]
    Ab = rb["A"].toarray()[np.ix_(pb, pb)]
    scale = max(1.0, np.abs(Aa).max())
    assert np.abs(Aa - Ab).max() <= 1e-12 * scale
    for k in ("b", "b_lifted"):
        if k in ra:
            assert np.abs(ra[k][pa] - rb[k][pb]).max() <= 1e-12 * max(1.0, np.abs(ra[k]).max())
