"""Automatic locality (dolfinx_mpc_amd/locality.py): with MPCX_AUTO_REORDER=1 every assembly runs on the spatially
reordered twin of the problem and hands its results back in the caller's numbering -- same pattern, same values as the
oracle on the caller's own numbering, for every small case, the irregular meshes and shuffled box meshes; live values
(coefficients, Dirichlet data, a moved mesh) reach the twin."""

import numpy as np
import pytest

import dolfinx_mpc_amd as dm
from dolfinx_mpc_amd import fem
from problems import (all_small_cases, case_cube_periodic, irregular_cases, oracle_mpc, oracle_outputs, product_mpc,
                      product_outputs)

pytestmark = pytest.mark.gpu

CASES = all_small_cases() + irregular_cases() + [lambda: case_cube_periodic(6, 1, 0.0, numbering="shuffled"),
                                                 lambda: case_cube_periodic(4, 2, 0.3, numbering="shuffled")]
RTOL = 1e-12


def _close(got, ref, what):
    scale = max(1.0, abs(ref).max())
    d = abs(got - ref).max()
    assert d <= RTOL * scale, f"{what}: max diff {d:.3e} > {RTOL * scale:.3e}"


@pytest.mark.parametrize("handback", ["fused", "fused-ordered", "eager"])
@pytest.mark.parametrize("make", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_twin_assembly_matches_oracle_in_caller_numbering(oracle, make, handback, monkeypatch):
    """fused (default): the twin's kernels write through mpcx_matrix_args_t::val_map / mpcx_vector_args_t::row_map into the
    caller's CSR and vector (-ordered: the row-block write-outs through out_map / out_delta); eager: passes of their own
    (mpcx_permute_values, a gather) after the twin's assembly"""
    monkeypatch.setenv("MPCX_AUTO_REORDER", "1")
    monkeypatch.setenv("MPCX_TWIN_HANDBACK", handback.split("-")[0])
    monkeypatch.setenv("MPCX_TWIN_WRITE_ORDER", "1" if handback.endswith("ordered") else "0")
    case = make()
    ref = oracle_outputs(oracle, case)
    out = product_outputs(case, algorithm="rowblock")
    if case.V.mesh.node_tile_offsets is None:
        assert getattr(case.V.mesh, "_twin", None) is not None  # the twin really ran
    if "A" in ref:
        assert np.array_equal(out["A"].indptr, ref["A"].indptr) and np.array_equal(out["A"].indices, ref["A"].indices)
        _close(out["A"].data, ref["A"].data, case.name + " A")
    for k in ("b", "b_lifted"):
        if k in ref:
            _close(out[k], ref[k], f"{case.name} {k}")


def test_twin_follows_live_values_and_moved_mesh(oracle, monkeypatch):
    """a coefficient written through a kept view, a changed Dirichlet value and a moved mesh between two assemblies"""
    from dolfinx_mpc_amd.mesh import create_delaunay_box

    monkeypatch.setenv("MPCX_AUTO_REORDER", "1")
    mesh = create_delaunay_box((0, 0, 0), (1, 1, 1), (4, 4, 4), 11)
    V = fem.functionspace(mesh, ("Lagrange", 1))
    f = fem.Function(V)
    f.interpolate(lambda x: 1.0 + x[0] + 2 * x[1])
    a = fem.form_stiffness(V, coefficient=f, constant=2.0)
    g = fem.Function(V)
    g.interpolate(lambda x: x[2])
    bc = fem.dirichletbc(g, fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[1], 0.0)), V)
    L = fem.form_source(V, fem.FN_POLY3)
    m = dm.MultiPointConstraint(V)
    m.create_periodic_constraint_geometrical(V, lambda x: np.isclose(x[0], 1.0), lambda x: np.stack([x[0] - 1, x[1], x[2]]), [bc], 1.0)
    raw = (m._slaves.copy(), m._masters.copy(), m._coeffs.copy(), m._owners.copy(), m._offsets.copy())
    m.finalize()
    om = oracle.OracleMPC.from_raw(V, *raw)

    def both():
        A = dm.assemble_matrix(a, m, bcs=[bc]).to_scipy()
        b = dm.assemble_vector(L, m)
        dm.apply_lifting(b, [a], [[bc]], m)
        Ao = oracle.assemble_matrix(a, om, bcs=[bc])
        bo = oracle.assemble_vector(L, om)
        oracle.apply_lifting(bo, [a], [[bc]], om)
        _close(A.data, Ao.data, "A")
        _close(b.numpy(), bo, "b")

    both()
    assert getattr(mesh, "_twin", None) is not None
    view = f.x.array
    view[:] = view * 0.5 + 3.0  # through a kept view
    g.x.array[:] += 1.5
    both()
    x = mesh.geometry.x.copy()
    x[:, 0] *= 1.25
    x[:, 2] += 0.1 * x[:, 1]
    mesh.geometry.x = x
    both()


def test_twin_is_skipped_for_tiled_meshes_and_when_switched_off(monkeypatch):
    from dolfinx_mpc_amd import locality
    from dolfinx_mpc_amd.mesh import create_delaunay_box, create_unit_cube

    tiled = create_unit_cube(4, 4, 4, reorder=(2, 2, 2))
    plain = create_delaunay_box((0, 0, 0), (1, 1, 1), (3, 3, 3), 0)
    monkeypatch.setenv("MPCX_AUTO_REORDER", "1")
    assert not locality.wanted(tiled) and locality.wanted(plain)
    monkeypatch.setenv("MPCX_AUTO_REORDER", "0")
    assert not locality.wanted(plain)
    monkeypatch.delenv("MPCX_AUTO_REORDER")
    assert not locality.wanted(plain)  # below MPCX_AUTO_REORDER_MIN_CELLS
    monkeypatch.setenv("MPCX_AUTO_REORDER_MIN_CELLS", "10")
    assert locality.wanted(plain)


def test_lazy_hand_back_defers_the_permutation_until_the_values_are_read(oracle, monkeypatch):
    """MPCX_TWIN_HANDBACK=lazy: assemble_matrix leaves the values in the twin's matrix; A.vals / to_scipy run the pass that
    writes them to the caller's positions, once; a later eager assembly into the same matrix is not disturbed"""
    import dolfinx_mpc_amd as dm
    from problems import case_cube_periodic, oracle_outputs, product_mpc

    monkeypatch.setenv("MPCX_AUTO_REORDER", "1")
    case = case_cube_periodic(5, 1, 0.3, numbering="shuffled")
    ref = oracle_outputs(oracle, case)["A"]
    mpc = product_mpc(case)
    monkeypatch.setenv("MPCX_TWIN_HANDBACK", "lazy")
    A = dm.assemble_matrix(case.a, mpc, bcs=case.bcs)
    assert A._twin_stale
    S = A.to_scipy()
    assert not A._twin_stale
    assert abs(S.data - ref.data).max() <= 1e-12 * abs(ref.data).max()
    dm.assemble_matrix(case.a, mpc, bcs=case.bcs, A=A)
    assert A._twin_stale
    for mode in ("eager", "fused"):
        monkeypatch.setenv("MPCX_TWIN_HANDBACK", mode)
        A.vals.fill_(-7.0)
        dm.assemble_matrix(case.a, mpc, bcs=case.bcs, A=A)
        assert not A._twin_stale
        assert abs(A.to_scipy().data - ref.data).max() <= 1e-12 * abs(ref.data).max()
