"""Irregular (Delaunay) meshes -- VERDICT r3 P-2: meshes with variable valence, no cell clusters and no numbering
locality, as gmsh hands them to the reference (python/tests/test_cube_contact.py:15-160), with constraints from the
library's builders (non-matching periodic faces, a slip wall, two-body contact).

CPU part: the generator gives valid conforming meshes, the oracle satisfies the reference's own identities
(python/src/dolfinx_mpc/utils/test.py:202-265) on them.  GPU part (-m gpu): every entry of both dispatch tables and both
algorithms against the oracle, pattern bit-exact, values to 1e-12 of the largest entry."""

import numpy as np
import pytest

from dolfinx_mpc_amd.mesh import create_delaunay_box, create_stacked_delaunay
from problems import irregular_cases, oracle_mpc, oracle_outputs, product_outputs

CASES = irregular_cases()
IDS = [f"irr{i}" for i in range(len(CASES))]
RTOL = 1e-12


@pytest.mark.parametrize("dim,n,seed", [(3, 4, 0), (3, 5, 9), (2, 7, 2)])
def test_delaunay_box_is_a_valid_conforming_mesh(dim, n, seed):
    mesh = create_delaunay_box((0.0,) * dim, (1.0,) * dim, (n,) * dim, seed)
    x, cells = mesh.geometry.x, mesh.geometry.dofmap.astype(np.int64)
    xv = x[cells][:, :, :dim]
    vol = np.abs(np.linalg.det(xv[:, 1:] - xv[:, :1])) / (6.0 if dim == 3 else 2.0)
    assert vol.min() > 0 and abs(vol.sum() - 1.0) < 1e-12  # fills the box, no flat cell
    # conforming: every facet belongs to one cell (boundary) or two, and the boundary facets tile the surface of the box
    from dolfinx_mpc_amd.mesh import local_facets

    lf = local_facets(mesh.cell_name)
    fv = np.sort(cells[:, lf], axis=2).reshape(-1, lf.shape[1])
    _, cnt = np.unique(fv, axis=0, return_counts=True)
    assert set(np.unique(cnt)) <= {1, 2}
    ext = mesh.exterior_facets()
    pts = x[cells[ext[:, 0]][np.arange(ext.shape[0])[:, None], lf[ext[:, 1]]]][:, :, :dim]
    if dim == 3:
        area = 0.5 * np.linalg.norm(np.cross(pts[:, 1] - pts[:, 0], pts[:, 2] - pts[:, 0]), axis=1)
        assert abs(area.sum() - 6.0) < 1e-12
    else:
        assert abs(np.linalg.norm(pts[:, 1] - pts[:, 0], axis=1).sum() - 4.0) < 1e-12
    # irregular: the vertex valence varies, the numbering has no locality hints
    val = np.bincount(cells.ravel())
    assert val.max() >= 2 * max(val[val > 0].min(), 1) and mesh.node_tile_offsets is None
    # opposite faces do not match
    a = np.sort(x[np.isclose(x[:, 0], 0.0)][:, 1])
    b = np.sort(x[np.isclose(x[:, 0], 1.0)][:, 1])
    assert a.size == b.size and not np.allclose(a, b)


def test_stacked_delaunay_bodies_touch_without_sharing_nodes():
    from dolfinx_mpc_amd.mesh import CONTACT_BOTTOM_INTERFACE, CONTACT_TOP_INTERFACE

    mesh, ft, ct = create_stacked_delaunay(2, 3, 1)
    top = np.unique(mesh.geometry.dofmap[ct == 2])
    bot = np.unique(mesh.geometry.dofmap[ct == 0])
    assert np.intersect1d(top, bot).size == 0
    assert ft.find(CONTACT_BOTTOM_INTERFACE).shape[0] > 0 and ft.find(CONTACT_TOP_INTERFACE).shape[0] > 0


@pytest.mark.parametrize("make", CASES, ids=IDS)
def test_oracle_identities_on_irregular_meshes(oracle, make):
    """A_mpc[free, free] == K^T A K, b_mpc[free] == K^T b with the builders' constraints (several masters per slave)"""
    po = oracle
    case = make()
    mpc = oracle_mpc(po, case)
    emp = po.OracleMPC.empty(case.V)
    out = oracle_outputs(po, case)
    A_org = po.assemble_matrix(case.a, emp, bcs=case.bcs, diagval=case.diagval)
    po.compare_mpc_lhs(A_org, out["A"], mpc, atol=5e3 * np.finfo(np.float64).resolution * max(1.0, abs(A_org).max()))
    b_org = po.assemble_vector(case.L, emp)
    po.compare_mpc_rhs(b_org, out["b"], mpc)
    po.apply_lifting(b_org, [case.a], [case.bcs], emp, scale=case.scale)
    po.compare_mpc_rhs(b_org, out["b_lifted"], mpc)
    # (the 2D slip wall has one master per slave; every other case has several)
    assert case.raw[1].size >= case.raw[0].size


def _close(got, ref, what):
    scale = max(1.0, abs(ref).max())
    d = abs(got - ref).max()
    assert d <= RTOL * scale, f"{what}: max diff {d:.3e} > {RTOL * scale:.3e}"


def _check(case, ref, out, what):
    if "A" in ref:
        assert np.array_equal(out["A"].indptr, ref["A"].indptr) and np.array_equal(out["A"].indices, ref["A"].indices)
        _close(out["A"].data, ref["A"].data, f"{case.name} A [{what}]")
    for k in ("b", "b_lifted"):
        if k in ref:
            _close(out[k], ref[k], f"{case.name} {k} [{what}]")


@pytest.mark.gpu
@pytest.mark.parametrize("alg", ["atomic", "rowblock"])
@pytest.mark.parametrize("make", CASES, ids=IDS)
def test_gpu_irregular_both_algorithms(oracle, make, alg):
    case = make()
    _check(case, oracle_outputs(oracle, case), product_outputs(case, algorithm=alg), alg)


def _table_names(which):
    from dolfinx_mpc_amd import dispatch

    return [k.name for k in (dispatch.MATRIX if which == "matrix" else dispatch.VECTOR)]


@pytest.mark.gpu
@pytest.mark.parametrize("name", _table_names("matrix"))
@pytest.mark.parametrize("make", CASES, ids=IDS)
def test_gpu_irregular_every_matrix_entry(oracle, make, name, monkeypatch):
    monkeypatch.setenv("MPCX_FORCE_KERNEL", f"matrix={name}")
    case = make()
    ref = oracle_outputs(oracle, case)
    out = product_outputs(case, algorithm="rowblock")
    _check(case, {"A": ref["A"]}, out, f"matrix={name}")


@pytest.mark.gpu
@pytest.mark.parametrize("name", _table_names("vector"))
@pytest.mark.parametrize("make", CASES, ids=IDS)
def test_gpu_irregular_every_vector_entry(oracle, make, name, monkeypatch):
    monkeypatch.setenv("MPCX_FORCE_KERNEL", f"vector={name}")
    case = make()
    ref = oracle_outputs(oracle, case)
    out = product_outputs(case, algorithm=None)
    _check(case, {k: v for k, v in ref.items() if k != "A"}, out, f"vector={name}")


@pytest.mark.gpu
@pytest.mark.parametrize("env", ["MPCX_NO_MPC_PLAN=1", "MPCX_FORCE_KERNEL=matrix=pairs+MPCX_PAIRS_DICT=1",
                                 "MPCX_FORCE_KERNEL=matrix=pairs+MPCX_PAIRS_MAX_NNZ=700", "MPCX_ROWBLOCK_MAX_NNZ=700+MPCX_ROWBLOCK_MAX_ROWS=24"])
@pytest.mark.parametrize("make", CASES, ids=IDS)
def test_gpu_irregular_plan_variants(oracle, make, env, monkeypatch):
    """the fall-backs an irregular mesh can hit: no master-contribution plan, the pattern dictionary (a Delaunay mesh
    has no repeating offset patterns), tiny row blocks (fat master rows next to the capacity)"""
    for part in env.split("+"):
        monkeypatch.setenv(*part.split("=", 1))
    case = make()
    _check(case, oracle_outputs(oracle, case), product_outputs(case, algorithm="rowblock"), env)
