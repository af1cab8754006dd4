"""The dof-transformation seam of imported element kernels (include/mpcx.h mpcx_ufcx_desc_t::transform0_name /
transform1_name, cell_info0 / cell_info1): the reference applies the element's transformations to the element tensor right
after the kernel call (cpp/assemble_matrix.cpp:432-436, 507-508; cpp/assemble_vector.cpp:184) with the cell permutation words
of the mesh (cpp/assemble_matrix.cpp:606-616).

Known answer: P3 Lagrange triangles with the edge dofs listed in the dofmap in the edge's GLOBAL direction whatever the cell's
local orientation (the "raw" dofmap DOLFINx builds for elements whose transformations are not baked into the dofmap), a cell
permutation word with one bit per reversed local edge, and a transformation that swaps the two dofs of a reversed edge (rows
for the test space, columns for the trial space): assembling the raw kernel tensor through that hook must give exactly the
matrix / vector the library's own P3 space gives (there the permutation is part of the dofmap, dolfinx_mpc_amd/elements.py).
Checked for the oracle (the hook restated in oracle/mpc_oracle.c) on the CPU and for every imported-kernel path on the GPU."""

import numpy as np
import pytest

from dolfinx_mpc_amd import fem
from dolfinx_mpc_amd.mesh import create_unit_square, local_edges
from problems import Case, dict_constraint_raw, l2b, oracle_outputs, product_outputs

TRANSFORMS = r"""
/* P3 triangle: local dofs 0-2 vertices, 3 + 2 e + k the k-th node of local edge e, 9 the interior node; bit e of the cell's
   permutation word: local edge e runs against its global direction */
void p3_tri_T0(double* A, const uint32_t* cell_info, int32_t cell, int32_t n)
{
  const uint32_t w = cell_info[cell];
  for (int e = 0; e < 3; ++e)
    if ((w >> e) & 1u)
      for (int c = 0; c < n; ++c)
      {
        const double t = A[(3 + 2 * e) * n + c];
        A[(3 + 2 * e) * n + c] = A[(4 + 2 * e) * n + c];
        A[(4 + 2 * e) * n + c] = t;
      }
}
void p3_tri_T1(double* A, const uint32_t* cell_info, int32_t cell, int32_t n)
{
  const uint32_t w = cell_info[cell];
  for (int e = 0; e < 3; ++e)
    if ((w >> e) & 1u)
      for (int r = 0; r < n; ++r)
      {
        const double t = A[r * 10 + 3 + 2 * e];
        A[r * 10 + 3 + 2 * e] = A[r * 10 + 4 + 2 * e];
        A[r * 10 + 4 + 2 * e] = t;
      }
}
"""


def _cases(scramble=True):
    """(case on the library's own P3 space, the same problem on the raw-dofmap space with the hook)"""
    from dolfinx_mpc_amd.codegen import generate_general
    from dolfinx_mpc_amd.mesh import Mesh
    from dolfinx_mpc_amd.quadrature import make_quadrature

    base = create_unit_square(5, 3, "triangle")
    cells = base.geometry.dofmap.copy()
    if scramble:  # local vertex orders at random: every orientation of a shared edge occurs
        rng = np.random.default_rng(2)
        for c in range(cells.shape[0]):
            cells[c] = cells[c][rng.permutation(3)]
    mesh = Mesh(base.geometry.x, cells, "triangle")
    V = fem.functionspace(mesh, ("Lagrange", 3))
    s_m_c = {l2b([1, 0]): {l2b([0, 1]): 0.43, l2b([1, 1]): 0.11}, l2b([0, 0]): {l2b([1, 1]): 0.69}}
    bc = fem.dirichletbc(0.3, fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[1], 1.0) & (x[0] > 0.3) & (x[0] < 0.7)), V)
    ref = Case("p3_oriented", V, fem.form_stiffness(V) + fem.form_mass(V, constant=0.7), fem.form_source(V, fem.FN_SIN2D), [bc],
               dict_constraint_raw(V, s_m_c))
    # the raw space: same dofs, every edge's two dofs in the edge's global order in every cell; bit e = local edge e reversed
    le = local_edges("triangle")
    flip = np.stack([cells[:, a] > cells[:, b] for a, b in le], axis=1)
    assert flip.any() and not flip.all()
    mesh.cell_permutation_info = (flip * (1 << np.arange(3))[None, :]).sum(axis=1).astype(np.uint32)
    W = V.clone()
    raw = V.dofmap.list.copy()
    for e in range(3):
        f = flip[:, e]
        raw[f, 3 + 2 * e], raw[f, 4 + 2 * e] = V.dofmap.list[f, 4 + 2 * e], V.dofmap.list[f, 3 + 2 * e]
    W.dofmap.list[:] = raw
    forms = {}
    for kind, qdeg, kw in (("stiffness", 4, {}), ("mass", 6, dict(use_constant=True)), ("source", 3 + 4, dict(fexpr=fem.fn_c_expression(fem.FN_SIN2D)))):
        src, name = generate_general(kind, "triangle", 3, 1, make_quadrature("triangle", qdeg), **kw)
        forms[kind] = (src + TRANSFORMS, name)
    a = fem.form_ufcx([W, W], *forms["stiffness"], dof_transformations=("p3_tri_T0", "p3_tri_T1")) + \
        fem.form_ufcx([W, W], *forms["mass"], constant=fem.Constant(0.7), dof_transformations=("p3_tri_T0", "p3_tri_T1"))
    L = fem.form_ufcx([W], *forms["source"], dof_transformations=("p3_tri_T0",))
    bcw = fem.dirichletbc(0.3, bc.dof_indices()[0][: bc.dof_indices()[1]], W)
    hooked = Case("p3_raw_dofmap_with_transformations", W, a, L, [bcw], ref.raw)
    return ref, hooked


def _same(a, b, what):
    for k in a:
        x, y = (a[k].toarray(), b[k].toarray()) if k == "A" else (a[k], b[k])
        assert abs(x - y).max() <= 1e-12 * max(1.0, abs(x).max()), f"{what} {k}"


def test_oracle_applies_the_transformations():
    from oracle import pyoracle as po

    ref, hooked = _cases()
    want = oracle_outputs(po, ref)
    got = oracle_outputs(po, hooked)
    _same(want, got, "oracle")
    # without the hook the raw dofmap gives another matrix: the test can see the transformation
    hooked.a.integrals[0].kernel.ufcx_transforms = None
    hooked.a.integrals[1].kernel.ufcx_transforms = None
    other = oracle_outputs(po, hooked)["A"]
    assert abs(other - want["A"]).max() > 1e-3


def test_cell_info_is_required():
    from dolfinx_mpc_amd.mesh import create_unit_square as cus

    V = fem.functionspace(cus(2, 2), ("Lagrange", 3))
    with pytest.raises(ValueError):
        fem.form_ufcx([V], "void f(void) {}", "f", dof_transformations=("t0",))


@pytest.mark.gpu
@pytest.mark.parametrize("alg", ["atomic", "rowblock", None])
def test_gpu_paths_apply_the_transformations(oracle, alg):
    ref, hooked = _cases()
    want = oracle_outputs(oracle, ref)
    out = product_outputs(hooked, algorithm=alg)
    assert np.array_equal(out["A"].indptr, want["A"].indptr) and np.array_equal(out["A"].indices, want["A"].indices)
    _same({k: want[k] for k in out}, out, f"gpu[{alg}]")


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["MPCX_NO_MPC_PLAN", "MPCX_SLAVE_TENSORS=0", "MPCX_VECTOR_OWNER=0"])
def test_gpu_plan_variants_apply_the_transformations(oracle, monkeypatch, variant):
    key, _, val = variant.partition("=")
    monkeypatch.setenv(key, val or "1")
    ref, hooked = _cases()
    want = oracle_outputs(oracle, ref)
    out = product_outputs(hooked, algorithm="rowblock")
    _same({k: want[k] for k in out}, out, variant)


@pytest.mark.gpu
def test_locality_twin_leaves_forms_with_transformations_alone(oracle, monkeypatch):
    """the locality twin renumbers the vertices, the cell permutation words would be stale there --
    forms whose imported kernel names a dof transformation are assembled in the caller's numbering (the mesh of _cases()
    has no tile hints, so with MPCX_AUTO_REORDER=1 every other form would go through the twin)"""
    monkeypatch.setenv("MPCX_AUTO_REORDER", "1")
    ref, hooked = _cases()
    want = oracle_outputs(oracle, ref)
    out = product_outputs(hooked)
    _same({k: want[k] for k in out}, out, "forced twin")
