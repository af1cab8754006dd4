"""CPU tests of the product's host logic (no GPU compute):

* libmpcx.so loads and exports every symbol include/mpcx.h declares;
* native MultiPointConstraint finalize / cell_to_slaves / sparsity pattern /
  row-block plan against the oracle's numpy restatements of
  cpp/MultiPointConstraint.h:36-126, cpp/mpc_helpers.h:19-94, cpp/utils.h:381-496;
* API behaviour mirrored from python/src/dolfinx_mpc/multipointconstraint.py
  (add_constraint concatenation, finalize once, accessors) and the error
  conventions of SURVEY.md section 8b;
* the product fails loudly without a GPU (no CPU fallback).
"""

import os
import re

import numpy as np
import pytest

import dolfinx_mpc_amd as dm
from dolfinx_mpc_amd import _native, fem
from dolfinx_mpc_amd.mesh import create_unit_cube, create_unit_square
from problems import all_small_cases, case_cube_periodic, l2b, oracle_mpc, product_mpc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = _native.lib()
    header = open(os.path.join(ROOT, "include", "mpcx.h")).read()
    declared = set(re.findall(r"\b(mpcx_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(_native.EXPORTS)
    for name in declared:
        assert hasattr(L, name), f"libmpcx.so does not export {name}"
    assert L.mpcx_version() == int(re.search(r"#define MPCX_VERSION (\d+)", header).group(1))


def test_ctypes_structs_match_header_field_order():
    header = open(os.path.join(ROOT, "include", "mpcx.h")).read()

    def fields(struct_name):
        end = re.search(r"\}\s*" + struct_name + ";", header).start()
        start = header.rfind("typedef struct", 0, end)
        body = header[header.index("{", start) + 1 : end]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                decl = re.sub(r"\[[^\]]*\]\s*$", "", decl)  # (arrays: int32_t grid_n[3])
                names.append(re.findall(r"([A-Za-z_0-9]+)\s*$", decl)[0])
        return names

    assert fields("mpcx_kernel_t") == [f[0] for f in _native.KernelT._fields_]
    assert fields("mpcx_mpc_t") == [f[0] for f in _native.MpcT._fields_]
    assert fields("mpcx_rowblock_plan_t") == [f[0] for f in _native.RowBlockPlanT._fields_]
    assert fields("mpcx_matrix_args_t") == [f[0] for f in _native.MatrixArgs._fields_]
    assert fields("mpcx_vector_args_t") == [f[0] for f in _native.VectorArgs._fields_]
    assert fields("mpcx_lifting_args_t") == [f[0] for f in _native.LiftingArgs._fields_]


CASES = all_small_cases()


@pytest.mark.parametrize("make", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_finalize_c2s_pattern_match_oracle(oracle, make):
    case = make()
    mpc = product_mpc(case)
    om = oracle_mpc(oracle, case)
    assert np.array_equal(mpc.is_slave, om.is_slave)
    assert np.array_equal(mpc.slaves, om.slaves)
    assert mpc.num_local_slaves == om.num_local_slaves
    assert np.array_equal(mpc.masters.offsets, om.masters_offsets)
    assert np.array_equal(mpc.masters.array, om.masters)
    assert np.array_equal(mpc.coefficients()[0], om.coeffs)
    assert np.array_equal(mpc.coefficients()[1], om.masters_offsets)
    assert np.array_equal(mpc.cell_to_slaves.offsets, om.c2s_offsets)
    assert np.array_equal(mpc.cell_to_slaves.array, om.c2s)
    if case.a is not None:
        rp, cl = dm.create_sparsity_pattern(case.a, mpc)
        rp2, cl2 = oracle.create_pattern(case.a, om, om)
        assert np.array_equal(rp, rp2) and np.array_equal(cl, cl2)
        # sorted, unique columns per row
        for r in range(rp.size - 1):
            row = cl[rp[r] : rp[r + 1]]
            assert np.all(np.diff(row) > 0)


def test_pattern_threads_and_rectangular_pair(oracle):
    case = case_cube_periodic(4, 1, 0.0)
    mpc = product_mpc(case)
    p1 = dm.create_sparsity_pattern(case.a, mpc, num_threads=1)
    p8 = dm.create_sparsity_pattern(case.a, mpc, num_threads=8)
    assert np.array_equal(p1[0], p8[0]) and np.array_equal(p1[1], p8[1])
    # (mpc, empty) pair: python/tests/test_rectangular_assembly.py style row/col constraints
    emp = dm.MultiPointConstraint(case.V)
    emp.finalize()
    rp, cl = dm.create_sparsity_pattern(case.a, (mpc, emp))
    om, oe = oracle_mpc(oracle, case), oracle.OracleMPC.empty(case.V)
    rp2, cl2 = oracle.create_pattern(case.a, om, oe)
    assert np.array_equal(rp, rp2) and np.array_equal(cl, cl2)


def test_rowblock_plan_covers_every_entity_row_pair():
    case = case_cube_periodic(5, 1, 0.0, reorder=(4, 4, 4))
    mpc = product_mpc(case)
    rowptr, cols = dm.create_sparsity_pattern(case.a, mpc)
    L = _native.lib()
    p = _native._ptr
    dmap = case.V.dofmap.list
    ents = np.arange(dmap.shape[0], dtype=np.int32)
    hints = np.ascontiguousarray(case.mesh.node_tile_offsets, dtype=np.int32)
    assert hints[0] == 0 and np.all(np.diff(hints) > 0)
    for max_rows, max_nnz, use_hints in ((64, 600, True), (1000, 5120, False), (7, 100, True), (64, 2000, True)):
        h = L.mpcx_rowblock_plan_build(rowptr.size - 1, p(rowptr), max_rows, max_nnz, ents.size, 1, p(ents), p(dmap),
                                       dmap.shape[1], 1, p(hints) if use_hints else None,
                                       hints.size if use_hints else 0, 1)
        assert h
        nb = L.mpcx_rowblock_plan_num_blocks(h)
        row0 = np.empty(nb + 1, dtype=np.int32)
        off = np.empty(nb + 1, dtype=np.int64)
        be = np.empty(L.mpcx_rowblock_plan_num_ents(h), dtype=np.int32)
        L.mpcx_rowblock_plan_copy(h, p(row0), p(off), p(be))
        L.mpcx_rowblock_plan_free(h)
        assert row0[0] == 0 and row0[-1] == rowptr.size - 1 and np.all(np.diff(row0) > 0)
        assert np.diff(row0).max() <= max_rows and np.diff(rowptr[row0]).max() <= max_nnz
        if use_hints and max_rows == 64 and max_nnz == 2000:
            # capacity >= one 4x4x4 tile: every block is a union of whole tiles
            assert set(row0.tolist()) <= set(hints.tolist()) | {rowptr.size - 1}
        blk_of = np.repeat(np.arange(nb), np.diff(row0))
        # every (entity, block of one of its dofs) pair appears exactly once
        want = set()
        for e in ents:
            for b in set(blk_of[dmap[e]]):
                want.add((int(b), int(e)))
        got = set()
        for b in range(nb):
            lst = be[off[b] : off[b + 1]]
            assert np.all(np.diff(lst) > 0)
            got.update((b, int(e)) for e in lst)
        assert got == want


def test_add_constraint_concatenation_and_finalize_once():
    """python/src/dolfinx_mpc/multipointconstraint.py:118-153, 619-631"""
    mesh = create_unit_square(4, 4)
    V = fem.functionspace(mesh, ("Lagrange", 1))
    mpc = dm.MultiPointConstraint(V)
    with pytest.raises(RuntimeError, match="has not been finalized"):
        _ = mpc.slaves
    mpc.add_constraint(V, np.array([3], dtype=np.int32), np.array([0, 1], dtype=np.int64), np.array([0.5, 0.25]),
                       np.zeros(2, dtype=np.int32), np.array([0, 2], dtype=np.int32))
    mpc.add_constraint(V, np.array([7, 5], dtype=np.int32), np.array([2, 4, 6], dtype=np.int64), np.array([1.0, 2.0, 3.0]),
                       np.zeros(3, dtype=np.int32), np.array([0, 1, 3], dtype=np.int32))
    # empty additions are ignored
    mpc.add_constraint(V, np.array([], dtype=np.int32), np.array([], dtype=np.int64), np.array([]),
                       np.array([], dtype=np.int32), np.array([0], dtype=np.int32))
    assert np.array_equal(mpc._offsets, [0, 2, 3, 5])
    mpc.finalize()
    assert np.array_equal(mpc.slaves, [3, 5, 7])
    assert np.array_equal(mpc.masters.links(3), [0, 1])
    assert np.array_equal(mpc.masters.links(7), [2])
    assert np.array_equal(mpc.masters.links(5), [4, 6])
    assert mpc.masters.num_links(0) == 0
    c, off = mpc.coefficients()
    assert np.array_equal(c[off[5] : off[6]], [2.0, 3.0])
    assert mpc.num_local_slaves == 3 and mpc.function_space is V
    with pytest.raises(RuntimeError, match="already been finalized"):
        mpc.finalize()
    with pytest.raises(RuntimeError, match="already been finalized"):
        mpc.add_constraint(V, np.array([1], dtype=np.int32), np.array([0], dtype=np.int64), np.array([1.0]),
                           np.zeros(1, dtype=np.int32), np.array([0, 1], dtype=np.int32))
    with pytest.raises(NotImplementedError):  # (float32 / float64 / complex64 / complex128 are the reference's four types)
        dm.MultiPointConstraint(V, dtype=np.int64)
    # a complex constraint finalizes on the host with its coefficients in place
    mc = dm.MultiPointConstraint(V, dtype=np.complex128)
    mc.add_constraint(V, np.array([5, 3], dtype=np.int32), np.array([4, 6, 0], dtype=np.int64), np.array([2.0 + 1j, 3.0, -1j]),
                      np.zeros(3, dtype=np.int32), np.array([0, 2, 3], dtype=np.int32))
    mc.finalize(where="host")
    cc, oo = mc.coefficients()
    assert cc.dtype == np.complex128 and np.array_equal(cc[oo[5] : oo[6]], [2.0 + 1j, 3.0]) and np.array_equal(cc[oo[3] : oo[4]], [-1j])


def test_finalize_rejects_bad_indices():
    mesh = create_unit_square(2, 2)
    V = fem.functionspace(mesh, ("Lagrange", 1))
    mpc = dm.MultiPointConstraint(V)
    mpc.add_constraint(V, np.array([1], dtype=np.int32), np.array([10**6], dtype=np.int64), np.array([1.0]),
                       np.zeros(1, dtype=np.int32), np.array([0, 1], dtype=np.int32))
    with pytest.raises(RuntimeError, match="master index out of range"):
        mpc.finalize()


def test_convenience_builders_match_raw_arrays(oracle):
    from problems import case_cube_elasticity_slip, case_square_dict, case_vector_poisson

    # general (dict) constraint
    case = case_square_dict(2, (0, 1))
    mpc = dm.MultiPointConstraint(case.V)
    mpc.create_general_constraint({l2b([1, 0]): {l2b([0, 1]): 0.43, l2b([1, 1]): 0.11}, l2b([0, 0]): {l2b([0, 1]): 0.69}})
    mpc.finalize()
    om = oracle_mpc(oracle, case)
    assert np.array_equal(mpc.slaves, om.slaves) and np.array_equal(mpc.masters.array, om.masters)
    assert np.array_equal(mpc.coefficients()[0], om.coeffs)
    # sub-space variant
    case = case_vector_poisson(1, 0)
    mpc = dm.MultiPointConstraint(case.V)
    mpc.create_general_constraint({l2b([1, 0]): {l2b([1, 1]): 0.1, l2b([0.5, 1]): 0.3}}, 1, 0)
    mpc.finalize()
    om = oracle_mpc(oracle, case)
    assert np.array_equal(mpc.slaves, om.slaves) and np.array_equal(mpc.masters.array, om.masters)
    # periodic, geometrical
    case = case_cube_periodic(3, 2, 0.0)
    mpc = dm.MultiPointConstraint(case.V)

    def rel(x):
        out = x.copy()
        out[0] = 1 - x[0]
        return out

    mpc.create_periodic_constraint_geometrical(case.V, lambda x: np.isclose(x[0], 1), rel, case.bcs)
    mpc.finalize()
    om = oracle_mpc(oracle, case)
    assert np.array_equal(mpc.slaves, om.slaves) and np.array_equal(mpc.masters.array, om.masters)
    # (non-matching mapped points: tests/test_builders.py)
    # slip
    case = case_cube_elasticity_slip(3)
    x = case.V.tabulate_dof_coordinates()
    blocks = np.flatnonzero(np.isclose(x[:, 0], 1.0))
    nrm = np.array([1.0, 0.3, -0.2])
    nrm /= np.linalg.norm(nrm)
    mpc = dm.MultiPointConstraint(case.V)
    mpc.create_slip_constraint(case.V, blocks, np.tile(nrm, (blocks.size, 1)))
    mpc.finalize()
    om = oracle_mpc(oracle, case)
    assert np.array_equal(mpc.slaves, om.slaves) and np.array_equal(mpc.masters.array, om.masters)
    assert np.allclose(mpc.coefficients()[0], om.coeffs)


def test_no_cpu_fallback():
    """Without a HIP device the assembly path must raise, not fall back."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    case = case_cube_periodic(2, 1, 0.0)
    mpc = product_mpc(case)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dm.assemble_matrix(case.a, mpc, bcs=case.bcs)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dm.assemble_vector(case.L, mpc)
    src = ""
    pkg = os.path.join(ROOT, "dolfinx_mpc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                src += open(os.path.join(dirpath, f)).read()
    assert "pyoracle" not in src and "mpc_oracle" not in src and "import oracle" not in src


def test_form_errors_mirror_reference():
    case = case_cube_periodic(2, 1, 0.0)
    mpc = product_mpc(case)
    with pytest.raises(RuntimeError, match="not a bilinear form"):
        dm.create_sparsity_pattern(case.L, mpc)
    unfinal = dm.MultiPointConstraint(case.V)
    with pytest.raises(RuntimeError, match="has not been finalized"):
        dm.create_sparsity_pattern(case.a, unfinal)


def test_mesh_generators_counts_and_tiling():
    n = 4
    m = create_unit_cube(n, n, n)
    assert m.num_cells == 6 * n**3 and m.num_nodes == (n + 1) ** 3
    vol = np.abs(np.linalg.det(m.geometry.x[m.geometry.dofmap[:, 1:]] - m.geometry.x[m.geometry.dofmap[:, :1]])) / 6
    assert vol.sum() == pytest.approx(1.0)
    mt = create_unit_cube(n, n, n, reorder=(2, 2, 2))
    # same set of cells (as coordinate sets), different numbering
    def key(mesh):
        c = mesh.geometry.x[mesh.geometry.dofmap].reshape(mesh.num_cells, -1)
        return np.sort(np.round(c * 64).astype(np.int64) @ (7 ** np.arange(12) % 1000003))
    assert np.array_equal(key(m), key(mt))
    assert len(m.exterior_facets()) == 12 * n * n
    V2 = fem.functionspace(m, ("Lagrange", 2))
    E = 3 * n * (n + 1) ** 2 + 3 * n * n * (n + 1) + n**3
    assert V2.num_dofs == (n + 1) ** 3 + E == (2 * n + 1) ** 3


def test_compress_offsets_dictionary():
    L = _native.lib()
    p = _native._ptr
    rng = np.random.default_rng(3)
    base = rng.integers(0, 30, size=(37, 16), dtype=np.uint8)
    pick = rng.integers(0, 37, size=5000)
    rows = np.ascontiguousarray(base[pick])
    ids = np.empty(rows.shape[0], dtype=np.uint16)
    table = np.empty(64 * 16, dtype=np.uint8)
    n = L.mpcx_compress_offsets(p(rows), rows.shape[0], 16, 64, p(ids), p(table))
    assert n == np.unique(base[np.unique(pick)], axis=0).shape[0]
    assert np.array_equal(table[: n * 16].reshape(n, 16)[ids], rows)
    # too many distinct rows -> -1 (caller keeps the uncompressed table)
    many = np.ascontiguousarray(rng.integers(0, 255, size=(500, 9), dtype=np.uint8))
    ids = np.empty(500, dtype=np.uint16)
    table = np.empty(16 * 9, dtype=np.uint8)
    assert L.mpcx_compress_offsets(p(many), 500, 9, 16, p(ids), p(table)) == -1


def test_object_cache_is_keyed_by_identity_not_id():
    """_device.cached keeps its key objects alive and confirms hits with ``is``: an object created
    after another one died can never inherit its entry (bare id() keys could)."""
    from dolfinx_mpc_amd import _device as D

    class K:
        pass

    store, built = {}, []

    def make(tag):
        def build():
            built.append(tag)
            return tag
        return build

    a, b = K(), K()
    assert D.cached(store, "t", (a,), 0, make("a")) == "a"
    assert D.cached(store, "t", (a,), 0, make("a2")) == "a"  # hit
    assert D.cached(store, "t", (a,), 1, make("a1")) == "a1"  # other hashable part
    assert D.cached(store, "t", (b,), 0, make("b")) == "b"
    assert built == ["a", "a1", "b"]
    # the cache holds `a` strongly: a new object cannot get a's id while the entry lives
    ida = id(a)
    del a
    fresh = [K() for _ in range(1000)]
    assert all(id(f) != ida for f in fresh)
    # LRU bound
    for k in range(20):
        D.cached(store, "lru", (fresh[k],), 0, make(k), maxsize=4)
    assert len(store[("objcache", "lru")]) == 4


def test_coefficients_constants_and_bc_values_are_live():
    """Integral.coeffs / .constants and DirichletBC.values_at_dofs read the CURRENT values of the
    Function / Constant they were built from (the reference packs per assembly call,
    cpp/assemble_matrix.cpp:583-589; reads bc values per call, cpp/lifting.h:166-180)."""
    from dolfinx_mpc_amd import fem
    from dolfinx_mpc_amd.mesh import create_unit_square

    mesh = create_unit_square(3, 2)
    V = fem.functionspace(mesh, ("Lagrange", 1))
    w = fem.Function(V)
    c = fem.Constant(2.0)
    a = fem.form_stiffness(V, constant=c, coefficient=w)
    integ = a.integrals[0]
    assert np.all(integ.coeffs == 0.0) and integ.constants[0] == 2.0
    view = w.x.array  # a caller may keep the view and write through it between two assemblies
    view[:] = np.arange(V.num_dofs)
    c.value[0] = -1.0
    assert np.array_equal(integ.coeffs, np.arange(V.num_dofs, dtype=float)[V.dofmap.list])
    view[:] = 7.0
    assert np.all(integ.coeffs == 7.0)
    assert integ.constants[0] == -1.0
    g = fem.Function(V)
    bc = fem.dirichletbc(g, np.array([0, 3], dtype=np.int32), V)
    assert np.all(bc.values_at_dofs() == 0.0)
    g.x.array[:] = 4.0
    assert np.all(bc.values_at_dofs() == 4.0)
    kc = fem.Constant(1.0)
    bck = fem.dirichletbc(kc, np.array([1], dtype=np.int32), V)
    kc.value[0] = 9.0
    assert bck.values_at_dofs()[0] == 9.0


def test_periodic_builder_drops_whole_blocks_under_a_bc():
    """cpp/utils.h:1459-1496 + cpp/PeriodicConstraint.h:563-567: a slave block with ANY component under a
    Dirichlet condition is dropped entirely; conditions of other spaces are ignored."""
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import fem
    from dolfinx_mpc_amd.mesh import create_unit_square

    mesh = create_unit_square(3, 3)
    V = fem.functionspace(mesh, ("Lagrange", 1, (2,)))
    Q = fem.functionspace(mesh, ("Lagrange", 1))
    x = V.tabulate_dof_coordinates()
    top_right = np.flatnonzero(np.isclose(x[:, 0], 1) & np.isclose(x[:, 1], 1)).astype(np.int32)
    bc_y = fem.dirichletbc(0.0, top_right, V, component=1)  # only the y-component of one slave block
    bc_q = fem.dirichletbc(0.0, np.arange(Q.num_dofs, dtype=np.int32), Q)  # other space: ignored

    def relation(x):
        out = x.copy()
        out[0] = 1 - x[0]
        return out

    mpc = dm.MultiPointConstraint(V)
    mpc.create_periodic_constraint_geometrical(V, lambda x: np.isclose(x[0], 1), relation, [bc_y, bc_q])
    mpc.finalize()
    right = np.flatnonzero(np.isclose(x[:, 0], 1))
    expect = np.sort(np.concatenate([b * 2 + np.arange(2) for b in right if b != top_right[0]]))
    assert np.array_equal(mpc.slaves, expect)


def test_cell_cluster_detection():
    """clusters.kuhn_fans: six consecutive cells with the fan's vertex pattern; anything else is a leftover"""
    from dolfinx_mpc_amd.clusters import kuhn_fans
    from dolfinx_mpc_amd.distributed import create_box_slab
    from dolfinx_mpc_amd.mesh import create_unit_cube

    for mesh in (create_unit_cube(3, 4, 2), create_unit_cube(5, 5, 5, reorder=(2, 2, 2)),
                 create_box_slab((0, 0, 0), (1, 1, 1), (4, 4, 5), 1, 2, 2, (2, 2, 2))):
        verts, left = kuhn_fans(mesh.geometry.dofmap, mesh.num_owned_cells)
        assert left.size == 0 and verts.shape == (mesh.num_owned_cells // 6, 8)
        # the eight vertices of a fan are the corners of one cube: their coordinates span a box of volume h^3
        x = mesh.geometry.x[verts]
        ext = x.max(axis=1) - x.min(axis=1)
        assert np.allclose(ext.prod(axis=1), ext[0].prod())
        # cell t of fan g uses exactly the vertices of the pattern
        pattern = np.array([[0, 1, 3, 7], [0, 1, 7, 5], [0, 5, 7, 4], [0, 3, 2, 7], [0, 6, 4, 7], [0, 2, 6, 7]])
        cells = mesh.geometry.dofmap[: mesh.num_owned_cells].reshape(-1, 6, 4)
        assert np.array_equal(np.take_along_axis(verts[:, None, :].repeat(6, 1), pattern[None].repeat(verts.shape[0], 0), 2), cells)
    mesh = create_unit_cube(3, 3, 3)
    cells = mesh.geometry.dofmap.copy()
    cells[[7, 8]] = cells[[8, 7]]  # wrong order inside group 1
    cells[13] = cells[13][[1, 0, 3, 2]]  # permuted vertices inside group 2
    verts, left = kuhn_fans(cells, 100)  # 16 full groups + 4 trailing cells
    assert verts.shape[0] == 14 and sorted(left.tolist()) == list(range(6, 18)) + [96, 97, 98, 99]


def test_cluster_detection_from_topology_ignores_cell_and_vertex_order():
    """clusters.fans_from_topology (host restatement of the device detection): the same fans as the generator-order
    detector on the generator's meshes, and still all of them after the nodes were renumbered, the cells shuffled
    and the local vertices of every cell permuted (DOLFINx does all three); extra cells round an edge (duplicates)
    turn that fan into leftovers"""
    from dolfinx_mpc_amd.clusters import fans_from_topology, kuhn_fans
    from dolfinx_mpc_amd.mesh import Mesh, create_box, create_unit_cube
    from problems import renumbered

    pattern = np.array([[0, 1, 3, 7], [0, 1, 7, 5], [0, 5, 7, 4], [0, 3, 2, 7], [0, 6, 4, 7], [0, 2, 6, 7]])
    rng = np.random.default_rng(3)
    for mesh in (create_unit_cube(3, 4, 2), create_unit_cube(5, 5, 5, reorder=(2, 2, 2)),
                 create_box((0.0, 0.0, 0.0), (2.0, 1.0, 0.3), (4, 3, 5))):
        ref, _ = kuhn_fans(mesh.geometry.dofmap, mesh.num_cells)
        want = {tuple(sorted(r)) for r in ref.tolist()}
        sh = renumbered(mesh, "shuffled")
        cells = sh.geometry.dofmap.copy()
        for c in range(cells.shape[0]):
            cells[c] = cells[c][rng.permutation(4)]
        for m in (mesh, Mesh(sh.geometry.x, cells, "tetrahedron")):
            verts, left = fans_from_topology(m.geometry.x, m.geometry.dofmap, m.num_cells)
            assert left.size == 0 and verts.shape == (m.num_cells // 6, 8)
            # every fan's six tets (pattern over its eight vertices) are cells of the mesh, each cell exactly once
            tets = np.sort(np.take_along_axis(verts[:, None, :].repeat(6, 1), pattern[None].repeat(verts.shape[0], 0), 2), axis=2)
            have = np.sort(np.sort(m.geometry.dofmap, axis=1).view([("", np.int32)] * 4).ravel())
            assert np.array_equal(np.sort(tets.reshape(-1, 4).astype(np.int32).view([("", np.int32)] * 4).ravel()), have)
            # eight corners of one cube of the grid
            ext = np.ptp(m.geometry.x[verts], axis=1)
            assert np.allclose(ext.prod(axis=1), ext[0].prod())
        if mesh.node_tile_offsets is None and mesh.num_cells == 6 * 24:
            assert want == {tuple(sorted(int(v) for v in r)) for r in fans_from_topology(mesh.geometry.x, mesh.geometry.dofmap, mesh.num_cells)[0]}
    mesh = create_unit_cube(3, 3, 3)
    extra = mesh.geometry.dofmap[[7, 40, 41]]  # duplicates of three cells out of two different cubes
    cells = np.concatenate([mesh.geometry.dofmap, extra], axis=0)
    verts, left = fans_from_topology(mesh.geometry.x, cells, cells.shape[0])
    assert verts.shape[0] == 27 - 2 and left.size == 6 * 2 + 3
    assert set(left.tolist()) == set(range(6, 12)) | set(range(36, 42)) | {162, 163, 164}


def test_lagrange_basis_tables():
    from dolfinx_mpc_amd.quadrature import lagrange_basis, make_quadrature

    for cell, nd in (("tetrahedron", {1: 4, 2: 10}), ("triangle", {1: 3, 2: 6})):
        q, w = make_quadrature(cell, 4)
        for deg, n in nd.items():
            phi = lagrange_basis(cell, deg, q)
            assert phi.shape == (q.shape[0], n)
            assert np.allclose(phi.sum(axis=1), 1.0)  # partition of unity
        # P2 vertex functions integrate to -|T|/20 (tet) / 0 (triangle), edge functions to |T|/5 / |T|/3
        phi = lagrange_basis(cell, 2, q)
        vol = w.sum()
        nv = 4 if cell == "tetrahedron" else 3
        assert np.allclose(w @ phi[:, :nv], -vol / 20 if cell == "tetrahedron" else 0.0, atol=1e-15)
        assert np.allclose(w @ phi[:, nv:], vol / 5 if cell == "tetrahedron" else vol / 3)


def test_allcore_cpu_baseline_adds_up():
    """oracle/cpu_parallel.py (bench.py's all-core leg): the slabs' private matrices, reduced by the owners, equal the
    single-thread assembly (asserted inside for small N)"""
    from oracle import cpu_parallel

    r = cpu_parallel.main(10, 3, 1)
    assert r["cores"] == 3 and r["value"] > 0
    r = cpu_parallel.main(6, 2, 2)
    assert r["cores"] == 2 and "P2" in r["sample"]


def test_blocked_and_multiple_coefficients_are_packed_like_dolfinx():
    """dolfinx pack_coefficients: per entity the cell dofs of every coefficient in turn, unrolled dof * bs + k for
    blocked spaces (cpp/assemble_matrix.cpp:587-589 hands that array to the kernel)."""
    from dolfinx_mpc_amd import fem
    from dolfinx_mpc_amd.mesh import create_unit_square

    mesh = create_unit_square(2, 2)
    V = fem.functionspace(mesh, ("Lagrange", 1))
    W = fem.functionspace(mesh, ("Lagrange", 2, (2,)))
    f, g = fem.Function(V), fem.Function(W)
    f.x.array[:] = 100.0 + np.arange(V.num_dofs)
    g.x.array[:] = np.arange(W.num_dofs)
    cells = np.array([3, 0, 5], dtype=np.int32)
    form = fem.form_ufcx([V], "void k(void){}", "k", entities=cells, coefficient=[g, f])
    integ = form.integrals[0]
    assert integ.cstride == 6 * 2 + 3
    w = integ.coeffs
    assert w.shape == (3, 15)
    for e, c in enumerate(cells):
        exp_g = (W.dofmap.list[c][:, None] * 2 + np.arange(2)[None, :]).reshape(-1)
        assert np.array_equal(w[e, :12], exp_g.astype(float))
        assert np.array_equal(w[e, 12:], 100.0 + V.dofmap.list[c])


def test_geometry_is_read_only_and_moves_through_the_setter():
    """cpp/assemble_matrix.cpp:495-501 re-reads x on every call; here a moved mesh is stated with
    ``mesh.geometry.x = new`` (version counter -> device refresh) and an in-place write fails loudly"""
    from dolfinx_mpc_amd import fem
    from dolfinx_mpc_amd.mesh import create_unit_cube

    mesh = create_unit_cube(2, 2, 2)
    with pytest.raises(ValueError):
        mesh.geometry.x[:, 0] += 1.0
    V2 = fem.functionspace(mesh, ("Lagrange", 2))
    before = V2.tabulate_dof_coordinates().copy()
    v0 = mesh.geometry.version
    mesh.geometry.x = mesh.geometry.x * np.array([2.0, 1.0, 0.5])
    assert mesh.geometry.version == v0 + 1
    assert np.allclose(V2.tabulate_dof_coordinates(), before * np.array([2.0, 1.0, 0.5]))
    with pytest.raises(ValueError):
        mesh.geometry.x = np.zeros((3, 3))


def test_block_ranges_fat_row_is_plan_not_representable():
    """a CSR row longer than a row block's capacity (a master with thousands of slaves) must surface as
    PlanNotRepresentable so that algorithm='auto' falls back to the thread-per-entity kernels, not as a bare RuntimeError"""
    import importlib

    am = importlib.import_module("dolfinx_mpc_amd.assemble_matrix")
    lens = np.full(64, 15, dtype=np.int64)
    lens[17] = am.ROWBLOCK_MAX_NNZ + 5
    rowptr = np.zeros(65, dtype=np.int64)
    np.cumsum(lens, out=rowptr[1:])
    with pytest.raises(_native.PlanNotRepresentable):
        am._block_ranges(64, rowptr, am.ROWBLOCK_MAX_ROWS, am.ROWBLOCK_MAX_NNZ, 1, None)
    lens[17] = 15
    np.cumsum(lens, out=rowptr[1:])
    row0 = am._block_ranges(64, rowptr, am.ROWBLOCK_MAX_ROWS, am.ROWBLOCK_MAX_NNZ, 1, None)
    assert row0[0] == 0 and row0[-1] == 64


def test_ghost_update_defaults_follow_petsc():
    """b.ghostUpdate() without arguments is (INSERT, FORWARD) as in petsc4py; the mixed combinations raise"""
    from dolfinx_mpc_amd import la

    class Ex:
        calls = []

        def forward_vector(self, arr):
            self.calls.append("forward")

        def reduce_vector_begin(self, arr):
            self.calls.append("reverse")
            return None

    b = la.Vector.__new__(la.Vector)
    b._exchange, b._pending, b._ready = Ex(), None, None
    b._array = np.zeros(3)
    with pytest.raises(NotImplementedError):
        b.ghostUpdate(addv=la.InsertMode.ADD, mode=la.ScatterMode.FORWARD)
    with pytest.raises(NotImplementedError):
        b.ghostUpdate(addv=la.InsertMode.INSERT, mode=la.ScatterMode.REVERSE)
    b.ghostUpdate()
    b.ghostUpdate(addv=la.InsertMode.ADD, mode=la.ScatterMode.REVERSE)
    assert Ex.calls == ["forward", "reverse"]
