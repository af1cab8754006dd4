"""GPU parity: HIP kernels (through the public API -> ctypes -> C ABI) against
the CPU oracle on the same inputs.  Run with ``-m gpu`` on an MI355X.

Tolerances (fp64, stated by BASELINE.json's north_star as "a stated fp64
tolerance"): the GPU sums the same addends in a different order (atomics / LDS
accumulation), so entries agree to
    |A_gpu - A_oracle|_max <= 1e-12 * max(1, |A|_max)        (<= 30 addends of size |A|_max)
    |b_gpu - b_oracle|_max <= 1e-12 * max(1, |b|_max)
which is tighter than the reference's own acceptance threshold of 5e-12
(python/src/dolfinx_mpc/utils/test.py:207).
"""

import numpy as np
import pytest

from problems import all_small_cases, case_cube_periodic, oracle_mpc, oracle_outputs, product_mpc, product_outputs

pytestmark = pytest.mark.gpu

CASES = all_small_cases()
RTOL_A = 1e-12
RTOL_B = 1e-12


def _close(got, ref, rtol, what):
    scale = max(1.0, abs(ref).max())
    d = abs(got - ref).max()
    assert d <= rtol * scale, f"{what}: max diff {d:.3e} > {rtol * scale:.3e}"


@pytest.mark.parametrize("alg", ["atomic", "rowblock"])
@pytest.mark.parametrize("make", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_small_cases_match_oracle(oracle, make, alg):
    case = make()
    ref = oracle_outputs(oracle, case)
    out = product_outputs(case, algorithm=alg)
    if "A" in ref:
        # same sparsity pattern and same values
        assert np.array_equal(out["A"].indptr, ref["A"].indptr)
        assert np.array_equal(out["A"].indices, ref["A"].indices)
        _close(out["A"].data, ref["A"].data, RTOL_A, case.name + " A")
    for k in ("b", "b_lifted"):
        if k in ref:
            _close(out[k], ref[k], RTOL_B, f"{case.name} {k}")


@pytest.mark.parametrize("alg", ["atomic", "rowblock"])
def test_config1_cube32(oracle, alg):
    """BASELINE config 1: periodic Poisson P1 on the 32^3 cube (196 608 cells, 35 937 dofs, 961 slaves)."""
    case = case_cube_periodic(32, 1, 0.0)
    assert case.V.num_dofs == 35937 and case.mesh.num_cells == 196608 and case.raw[0].size == 961
    ref = oracle_outputs(oracle, case, fast=True)
    out = product_outputs(case, algorithm=alg)
    _close(out["A"].data, ref["A"].data, RTOL_A, "A")
    _close(out["b"], ref["b"], RTOL_B, "b")
    _close(out["b_lifted"], ref["b_lifted"], RTOL_B, "b_lifted")
    # reference identity on the GPU result itself (utils/test.py:202-242)
    mpc = oracle_mpc(oracle, case)
    A_org = oracle.assemble_matrix(case.a, oracle.OracleMPC.empty(case.V), bcs=case.bcs, fast=True)
    oracle.compare_mpc_lhs(A_org, out["A"], mpc)


def test_dictionary_compressed_plan(oracle, monkeypatch):
    """Opt-in plan variant (MPCX_OFFSET_DICT=1): 2-byte pattern ids + table of distinct
    scatter-offset rows (mpcx_compress_offsets) must give the same matrix."""
    monkeypatch.setenv("MPCX_OFFSET_DICT", "1")
    case = case_cube_periodic(12, 1, 0.0)
    ref = oracle_outputs(oracle, case, fast=True)
    out = product_outputs(case, algorithm="rowblock")
    _close(out["A"].data, ref["A"].data, RTOL_A, "A (dictionary plan)")


@pytest.mark.parametrize("make", CASES + [lambda: case_cube_periodic(9, 2, 0.0), lambda: case_cube_periodic(20, 1, 0.0)],
                         ids=[f"case{i}" for i in range(len(CASES) + 2)])
def test_device_pattern_equals_host_pattern(make):
    """create_sparsity_pattern(where="device") (mpcx_pattern_device_*) == the threaded host builder,
    bit for bit."""
    import dolfinx_mpc_amd as dm

    case = make()
    if case.a is None:
        pytest.skip("no bilinear form")
    mpc = product_mpc(case)
    rp_h, cols_h = dm.create_sparsity_pattern(case.a, mpc, where="host")
    rp_d, cols_d = dm.create_sparsity_pattern(case.a, mpc, where="device")
    assert rp_d.dtype == np.int64 and cols_d.dtype == np.int32
    assert np.array_equal(rp_h, rp_d)
    assert np.array_equal(cols_h, cols_d)


def test_single_master_long_row_fallbacks(oracle):
    """A master row with hundreds of columns (353 at N = 12): the device pattern builder hands over
    to the host builder (same result) and every matrix algorithm matches the oracle (the long row
    only receives master contributions, which go through matrix_mpc_kernel)."""
    import dolfinx_mpc_amd as dm
    from problems import case_cube_single_master

    case = case_cube_single_master(12)
    ref = oracle_outputs(oracle, case, fast=True)
    mpc = product_mpc(case)
    rp_h, cols_h = dm.create_sparsity_pattern(case.a, mpc, where="host")
    assert np.diff(rp_h).max() > 255
    rp_d, cols_d = dm.create_sparsity_pattern(case.a, mpc, where="device")
    assert np.array_equal(rp_h, rp_d) and np.array_equal(cols_h, cols_d)
    assert np.array_equal(rp_h, ref["A"].indptr) and np.array_equal(cols_h, ref["A"].indices)
    A = dm.assemble_matrix(case.a, mpc, bcs=case.bcs)  # auto
    _close(A.to_scipy().data, ref["A"].data, RTOL_A, "A (auto)")
    for alg in ("rowblock", "atomic"):
        A = dm.assemble_matrix(case.a, mpc, bcs=case.bcs, A=A, algorithm=alg)
        _close(A.to_scipy().data, ref["A"].data, RTOL_A, "A " + alg)
    b = dm.assemble_vector(case.L, mpc)
    _close(b.numpy(), ref["b"], RTOL_B, "b")


def test_rowblock_offset_overflow_falls_back_to_atomic(oracle):
    """The master sits at the end of its own 342-entry row: the 8-bit scatter offsets of its cells
    overflow, algorithm="rowblock" says so and "auto" assembles with the atomic kernel."""
    import dolfinx_mpc_amd as dm
    from problems import case_cube_single_master

    case = case_cube_single_master(12, (11 / 12, 1.0, 1.0))
    ref = oracle_outputs(oracle, case, fast=True)
    mpc = product_mpc(case)
    with pytest.raises(RuntimeError, match="row-block algorithm"):
        dm.assemble_matrix(case.a, mpc, bcs=case.bcs, algorithm="rowblock")
    A = dm.assemble_matrix(case.a, mpc, bcs=case.bcs)  # auto
    assert np.array_equal(A.rowptr, ref["A"].indptr) and np.array_equal(A.cols, ref["A"].indices)
    _close(A.to_scipy().data, ref["A"].data, RTOL_A, "A (auto -> atomic)")


@pytest.mark.parametrize("alg", ["atomic", "rowblock", None])
def test_empty_integration_domain(oracle, alg):
    """No entities: the matrix holds only the slave / Dirichlet diagonal, the vector is zero,
    lifting does nothing (the reference loops simply do not execute)."""
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import fem

    case = case_cube_periodic(4, 1, 0.7)
    V = case.V
    none = np.zeros(0, dtype=np.int32)
    a0 = fem.form_stiffness(V, cells=none)
    L0 = fem.form_source(V, fem.FN_POLY3, cells=none)
    mpc = product_mpc(case)
    A = dm.assemble_matrix(a0, mpc, bcs=case.bcs, diagval=2.0, algorithm=alg)
    As = A.to_scipy()
    expect = np.zeros(V.num_dofs)
    expect[mpc.slaves[: mpc.num_local_slaves]] = 2.0
    for bc in case.bcs:
        expect[bc.dof_indices()[0]] += 2.0
    assert np.array_equal(As.diagonal(), expect)
    assert abs(As - scipy_diag(expect)).max() == 0.0
    b = dm.assemble_vector(L0, mpc, algorithm=alg)
    dm.apply_lifting(b, [a0], [case.bcs], mpc)
    assert np.all(b.numpy() == 0.0)


def scipy_diag(d):
    import scipy.sparse

    return scipy.sparse.diags(d).tocsr()


def test_repeated_assembly_into_same_matrix(oracle):
    """A given -> zeroed and re-assembled (python/src/dolfinx_mpc/assemble_matrix.py:49-51)."""
    import dolfinx_mpc_amd as dm

    case = case_cube_periodic(5, 1, 0.0)
    mpc = product_mpc(case)
    A = dm.assemble_matrix(case.a, mpc, bcs=case.bcs)
    first = A.to_scipy().data.copy()
    for alg in ("atomic", "rowblock", "atomic"):
        dm.assemble_matrix(case.a, mpc, bcs=case.bcs, A=A, algorithm=alg)
        _close(A.to_scipy().data, first, 1e-13, "re-assembly " + alg)


def test_backsubstitution_homogenize(oracle):
    import torch

    from dolfinx_mpc_amd.la import Vector
    from problems import case_cube_contact_like

    case = case_cube_contact_like(3)
    mpc = product_mpc(case)
    rng = np.random.default_rng(1)
    u = rng.standard_normal(case.V.num_dofs)
    v = Vector(case.V.num_dofs)
    v.array.copy_(torch.from_numpy(u))
    mpc.backsubstitution(v)
    ref = u.copy()
    oracle.backsubstitution(oracle_mpc(oracle, case), ref)
    assert np.allclose(v.numpy(), ref, rtol=0, atol=1e-14)
    mpc.homogenize(v)
    assert np.all(v.numpy()[mpc.slaves] == 0.0)


# ---------------------------------------------------------------------------------------------
# kernel variants that the default configuration never reaches
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("env", ["MPCX_NO_MPC_PLAN=1", "MPCX_NO_LEAN=1", "MPCX_MPC_PLAN=host", "MPCX_NO_CUBE=1", "MPCX_PLAN_LISTS=host",
                                 "MPCX_ROWPAIR=all", "MPCX_ROWPAIR=none", "MPCX_NO_GROUP_ROWS=1", "MPCX_NO_NODEBLOCK=1",
                                 "MPCX_FORCE_KERNEL=matrix=pairs+MPCX_PAIRS_DICT=1", "MPCX_FORCE_KERNEL=matrix=pairs+MPCX_PAIRS_CONTEXT=recompute",
                                 "MPCX_FORCE_KERNEL=matrix=pairs+MPCX_PAIRS_MAX_NNZ=600"])
@pytest.mark.parametrize("alg", ["atomic", "rowblock"])
@pytest.mark.parametrize("make", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_small_cases_kernel_variants(oracle, make, alg, env, monkeypatch):
    """MPCX_NO_MPC_PLAN=1: the master contributions of the slave entities come from ``matrix_mpc_kernel``
    (one thread per slave entity, CSR searches, device atomics) -- the path a caller of the bare C ABI
    gets when it passes ``mpc_plan_off == NULL`` -- instead of the host-built plan.
    MPCX_NO_LEAN=1: the general row-block kernel instead of the lean P1 one.
    MPCX_MPC_PLAN=host: the plan from the host builder mpcx_mpc_plan_build instead of the device kernel.
    MPCX_NO_CUBE=1: the per-cell lean kernels where the default takes the cell-cluster kernels (MPCX_ALG_CUBE).
    MPCX_ROWPAIR=all / none: the (entity, local row) row-pair kernel wherever the operator has a compact context
    (default: vector-valued P1 only) / nowhere.  MPCX_NO_GROUP_ROWS=1: block entity lists in plain entity order.
    MPCX_NO_NODEBLOCK=1: component-diagonal forms on blocked spaces through the per-row compact layout of
    matrix_rowblock_kernel instead of matrix_nodeblock_kernel.  matrix=pairs + MPCX_PAIRS_DICT=1 / MPCX_PAIRS_CONTEXT=recompute /
    MPCX_PAIRS_MAX_NNZ=600: the pair-record kernel with dictionary-compressed records, with contexts computed per pair
    instead of cached, and with tiny row blocks (many blocks, segments shorter than a wave)."""
    for part in env.split("+"):
        monkeypatch.setenv(*part.split("=", 1))
    case = make()
    if case.a is None:
        pytest.skip("no bilinear form")
    ref = oracle_outputs(oracle, case)
    out = product_outputs(case, algorithm=alg)
    assert np.array_equal(out["A"].indptr, ref["A"].indptr) and np.array_equal(out["A"].indices, ref["A"].indices)
    _close(out["A"].data, ref["A"].data, RTOL_A, f"{case.name} A [{env}]")


@pytest.mark.parametrize("rows", [64, 1000])
@pytest.mark.parametrize("idx", [0, 8, 12, 15, 19, 22, 25])
def test_owner_plan_through_the_c_abi_equals_the_torch_builder(idx, rows):
    """the owner-computes vector plan from libmpcx (mpcx_owner_plan_*: fused passes + rocPRIM) is, array by array, the
    plan the torch gathers / searches / unique build"""
    import importlib

    import torch

    from dolfinx_mpc_amd import _device as D

    av = importlib.import_module("dolfinx_mpc_amd.assemble_vector")
    case = CASES[idx]()
    if case.L is None:
        pytest.skip("no linear form")
    V = case.V
    mpc = product_mpc(case)
    sd = D.space_device(V)
    nd = V.element_ndofs
    flag = torch.zeros(sd["dofmap"].numel(), dtype=torch.int32, device=sd["dofmap"].device)
    flag[::3] = 1 << 28  # any flag pattern: the builder carries the bits through
    mrow = (sd["dofmap"].view(-1) | flag).view(-1, nd)[: case.L.integrals[0].num_entities].contiguous()
    a = av._owner_plan_from_rows(mrow, V, rows * V.dofmap.bs)
    b = av._owner_plan_from_rows_torch(mrow, V, rows * V.dofmap.bs)
    assert (a is None) == (b is None)
    if a is None:
        return
    assert a[2] == b[2] and a[0].num_blocks == b[0].num_blocks and a[0].max_rows == b[0].max_rows
    names = ("row0", "off", "order", "lmap", "hoff", "spill", "sorder", "urows", "seg")
    for name, x, y in zip(names, a[1], b[1]):
        assert x.shape == y.shape and torch.equal(x.to(torch.int64), y.to(torch.int64)), f"{case.name}: {name}"
    del mpc


@pytest.mark.parametrize("env", ["MPCX_VECTOR_OWNER=1", "MPCX_VECTOR_OWNER=0"])
@pytest.mark.parametrize("make", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_small_cases_vector_kernel_variants(oracle, make, env, monkeypatch):
    """the row-block vector kernels for every case (MPCX_VECTOR_ALG=rowblock), with the owner-computes lists
    (vector_ownblock_kernel + vector_spill_reduce_kernel: every entity evaluated once, halo rows through LDS and a
    spill array) forced on / off -- the default takes them for rules of more than four points only"""
    monkeypatch.setenv("MPCX_VECTOR_ALG", "rowblock")
    monkeypatch.setenv(*env.split("="))
    case = make()
    if case.L is None:
        pytest.skip("no linear form")
    ref = oracle_outputs(oracle, case)
    out = product_outputs(case, algorithm="rowblock")
    for k in ("b", "b_lifted"):
        if k in ref:
            _close(out[k], ref[k], RTOL_B, f"{case.name} {k} [{env}]")


@pytest.mark.parametrize("env", ["MPCX_VERTEX_SOURCE=0", "MPCX_AFFINE_OWNBLOCK=0", "MPCX_AFFINE_THREADS=256"])
@pytest.mark.parametrize("make", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_affine_source_paths(oracle, make, env, monkeypatch):
    """Source forms whose integrand function is affine in x (constant / linear: the body forces of the Stokes, contact and
    elasticity cases): round 6 evaluates them from the rule's vertex moments (mpcx_kernel_t::vphi) -- inside the general
    kernels and, with an owner-computes plan, in vector_ownblock_affine_kernel.  With the owner-computes plan forced: the rule
    walked (MPCX_VERTEX_SOURCE=0), the moments inside the general instance (MPCX_AFFINE_OWNBLOCK=0), the dedicated kernel with
    another thread count -- all against the oracle, which always walks the rule (cpp/assemble_vector.cpp:65-90)."""
    case = make()
    if case.L is None:
        pytest.skip("no linear form")
    k, v = env.split("=")
    monkeypatch.setenv(k, v)
    monkeypatch.setenv("MPCX_VECTOR_OWNER", "1")
    ref = oracle_outputs(oracle, case)
    out = product_outputs(case, algorithm="rowblock")
    for key in ("b", "b_lifted"):
        if key in ref:
            _close(out[key], ref[key], RTOL_B, f"{case.name} {key}")


@pytest.mark.parametrize("env", ["MPCX_VCUBE_OWNER=0", "MPCX_VCUBE_OWNER=1", "MPCX_VCUBE_ROWS=256", "MPCX_CLUSTER_DETECT=consecutive",
                                 "MPCX_CUBE_NARROW=0", "MPCX_CUBE_MAX_ROWS=64", "MPCX_OWNER_PLAN=torch"])
@pytest.mark.parametrize("n,reorder,bc", [(4, None, 0.0), (6, (2, 2, 2), 2.3), (9, (4, 4, 4), 0.0)])
def test_cluster_vector_kernel_variants(oracle, n, reorder, bc, env, monkeypatch):
    """the P1 source through the cell-cluster kernels (algorithm "auto"): owner-computes row blocks over the clusters
    (default: no hash table, no device atomics) with large and small blocks, the LDS-hash kernel
    (MPCX_VCUBE_OWNER=0), and the generator-order cluster detector; periodic slaves and lifting included"""
    monkeypatch.setenv(*env.split("="))
    case = case_cube_periodic(n, 1, bc, reorder=reorder)
    ref = oracle_outputs(oracle, case)
    out = product_outputs(case, algorithm=None)
    for k in ("b", "b_lifted"):
        _close(out[k], ref[k], RTOL_B, f"{case.name} {k} [{env}]")
    _close(out["A"].data, ref["A"].data, RTOL_A, f"{case.name} A [{env}]")


@pytest.mark.parametrize("mode", ["tensor", "box", "points"])
@pytest.mark.parametrize("shape", ["cube", "stretched", "mirrored", "shuffled", "half_warped", "warped"])
def test_cluster_vector_on_the_tensor_grid_of_a_box(oracle, shape, mode, monkeypatch):
    """the benchmark's right-hand side on clusters that are axis-aligned boxes: (tensor) its univariate factors from a table
    filled once per launch and interval of the mesh's tensor grid (mpcx_vector_args_t::grid_*), (box, MPCX_TENSOR_GRID=0)
    evaluated per cluster on the 19 coordinates per axis the 14-point rule puts into a box (csrc/mpcx_box14.hpp), (points,
    MPCX_BOX_GRID=0) point by point.  Cubes, boxes with three different edges, a mirrored numbering (negative edge), a mesh
    where only some clusters are boxes and one where none is -- all against the oracle, and the evaluations against each other"""
    import importlib

    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import fem
    from dolfinx_mpc_amd.la import create_vector
    from dolfinx_mpc_amd.mesh import create_unit_cube
    from problems import Case, _walls_yz, periodic_raw

    env = {"tensor": ("1", "1"), "box": ("1", "0"), "points": ("0", "0")}

    def select(m):
        monkeypatch.setenv("MPCX_BOX_GRID", env[m][0])
        monkeypatch.setenv("MPCX_TENSOR_GRID", env[m][1])

    select(mode)
    if shape in ("half_warped", "warped"):
        case = case_cube_periodic(6, 1, 0.0, reorder=(2, 2, 2), warp="half" if shape == "half_warped" else True)
    elif shape == "shuffled":
        # nodes and cells in random order: the clusters are found from topology with their ring started anywhere, and
        # numbered along the coordinate axes by mpcx_cluster_canonical -- boxes all the same
        case = case_cube_periodic(7, 1, 0.0, numbering="shuffled")
    else:
        mesh = create_unit_cube(6, 5, 7, reorder=(2, 2, 2))
        x = mesh.geometry.x.copy()
        if shape == "stretched":
            x[:, 1] *= 0.7
            x[:, 2] = 0.05 + 1.3 * x[:, 2]
        elif shape == "mirrored":
            x[:, 1] = 1.0 - x[:, 1]
        mesh.geometry.x = x
        V = fem.functionspace(mesh, ("Lagrange", 1))
        bc = fem.dirichletbc(0.0, fem.locate_dofs_geometrical(V, _walls_yz), V)
        case = Case("box_" + shape, V, fem.form_stiffness(V, constant=1.3), fem.form_source(V, fem.FN_BENCH_PERIODIC, constant=0.7),
                    [bc], periodic_raw(V, [bc]))
    ref = oracle_outputs(oracle, case)
    mpc = product_mpc(case)
    av = importlib.import_module("dolfinx_mpc_amd.assemble_vector")
    args = av.vector_args(case.L, 0, create_vector(case.V), mpc, 0)[0]
    assert args.kernel_name == "cube_own", "the cluster kernel was expected to run"
    boxes = shape in ("cube", "stretched", "mirrored", "shuffled")
    assert int(args.cube_boxes) == (0 if (mode == "points" or shape == "warped") else 1)
    assert bool(args.grid_idx) == (mode == "tensor" and boxes), "tensor-grid tables: exactly on meshes of boxes"
    got = dm.assemble_vector(case.L, mpc).numpy().copy()
    _close(got, ref["b"], RTOL_B, f"{case.name} b [{mode}]")
    if mode != "points":
        select("points")
        other = dm.assemble_vector(case.L, mpc).numpy()
        assert abs(other - got).max() <= 1e-14 * abs(ref["b"]).max()
        if boxes:
            assert not np.array_equal(other, got), "the evaluations round differently: the switch had no effect"
    if mode == "tensor" and boxes:
        # moving the mesh rebuilds the intervals (geometry version in the cache key); a mesh that is no longer made of
        # boxes drops back to the per-cluster evaluation
        select("tensor")
        x = case.V.mesh.geometry.x.copy()
        x[:, 0] = x[:, 0] * (1.0 + 0.25 * x[:, 0])  # still a tensor grid, no longer uniform in x
        case.V.mesh.geometry.x = x
        ref2 = oracle_outputs(oracle, case)
        args = av.vector_args(case.L, 0, create_vector(case.V), mpc, 0)[0]
        assert bool(args.grid_idx)
        _close(dm.assemble_vector(case.L, mpc).numpy(), ref2["b"], RTOL_B, f"{case.name} b after the mesh moved")
        x[:, 1] += 0.02 * np.sin(3.0 * x[:, 0])  # sheared: no boxes
        case.V.mesh.geometry.x = x
        ref3 = oracle_outputs(oracle, case)
        args = av.vector_args(case.L, 0, create_vector(case.V), mpc, 0)[0]
        assert not bool(args.grid_idx)
        _close(dm.assemble_vector(case.L, mpc).numpy(), ref3["b"], RTOL_B, f"{case.name} b after the mesh was sheared")


@pytest.mark.parametrize("degree,warp,env", [(2, False, None), (2, False, "MPCX_VECTOR_OWNER_ROWS=512"), (1, False, "MPCX_FORCE_KERNEL=vector=ownblock"),
                                             (2, "half", None), (2, True, None)])
def test_cell_vector_from_the_tensor_grid_tables(oracle, degree, warp, env, monkeypatch):
    """the benchmark's right-hand side per cell from per-interval tables (mpcx_vector_args_t::grid_eta / grid_J, any rule): scalar
    P2 on a box mesh (24-point rule), small blocks (several row lists), P1 through the per-cell owner blocks (14-point rule), and
    meshes with warped cells (no grid: point by point) -- against the oracle, and against the point-by-point evaluation"""
    import importlib

    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd.la import create_vector

    if env:
        monkeypatch.setenv(*env.split("=", 1))
    case = case_cube_periodic(5, degree, 0.0, reorder=(2, 2, 2), warp=warp)
    ref = oracle_outputs(oracle, case)
    mpc = product_mpc(case)
    av = importlib.import_module("dolfinx_mpc_amd.assemble_vector")
    args = av.vector_args(case.L, 0, create_vector(case.V), mpc, 0)[0]
    assert args.kernel_name == "ownblock"
    assert bool(args.grid_J) == (not warp), "per-cell tables: exactly on meshes whose cells are cells of boxes"
    got = dm.assemble_vector(case.L, mpc).numpy().copy()
    _close(got, ref["b"], RTOL_B, f"{case.name} b")
    monkeypatch.setenv("MPCX_CELL_GRID", "0")
    assert not bool(av.vector_args(case.L, 0, create_vector(case.V), mpc, 0)[0].grid_J)
    other = dm.assemble_vector(case.L, mpc).numpy()
    assert abs(other - got).max() <= 1e-14 * abs(ref["b"]).max()
    if not warp:
        assert not np.array_equal(other, got), "the evaluations round differently: the switch had no effect"


def test_cluster_plan_uses_narrow_and_wide_records(oracle):
    """the cluster plan keeps 64-byte records (4-bit offsets) for row blocks whose rows have at most 16 entries before
    any cluster column and 96-byte records for the blocks with fat rows (the periodic master rows): both formats are
    launched, and together they give the oracle's matrix"""
    import dolfinx_mpc_amd as dm

    case = case_cube_periodic(32, 1, 0.0, reorder=(8, 8, 8))  # 512-row blocks = tiles: only those at x = 0 hold master rows
    mpc = product_mpc(case)
    A = dm.assemble_matrix(case.a, mpc, bcs=case.bcs)
    (parts, _keep, info), = [v[1] for v in A._plans[("objcache", "cubes")].values()]
    assert [p[2] for p in parts] == [64, 96], "narrow and wide row blocks expected"
    assert 0 < info["narrow_blocks"] < info["num_blocks"]
    ref = oracle.assemble_matrix(case.a, oracle_mpc(oracle, case), bcs=case.bcs, fast=True)
    _close(A.to_scipy().data, ref.data, RTOL_A, "A (narrow + wide cluster records)")


def test_cluster_blocks_split_by_cluster_shape(oracle, monkeypatch):
    """clusters that are parallelepipeds (x > 0.375) and clusters of moved nodes: row blocks all of whose clusters are
    parallelepipeds are launched with the closed-form kernel (cube_flags bit 0), the others with the kernel that takes
    every tet's own geometry; both record formats occur as well; moving the mesh afterwards rebuilds the split"""
    import importlib

    import dolfinx_mpc_amd as dm
    from problems import warped

    am = importlib.import_module("dolfinx_mpc_amd.assemble_matrix")
    monkeypatch.setattr(am, "CUBE_MAX_ROWS", 64)
    monkeypatch.setattr(am, "CUBE_MAX_NNZ", 64 * 16)
    # 13 nodes per direction in tiles of 4: the tile of nodes 4..7 holds distorted clusters and no fat row (narrow records),
    # the tiles next to the periodic faces hold fat rows (wide records), with and without distorted clusters
    case = case_cube_periodic(12, 1, 0.3, reorder=(4, 4, 4), warp="half")
    ref = oracle_outputs(oracle, case)
    mpc = product_mpc(case)
    A = dm.assemble_matrix(case.a, mpc, bcs=case.bcs, diagval=case.diagval)
    (parts, _keep, info), = [v[1] for v in A._plans[("objcache", "cubes")].values()]
    assert {p[5] for p in parts} == {0, 1} and {p[2] for p in parts} == {64, 96}
    assert 0 < info["closed_form_blocks"] < info["num_blocks"]
    _close(A.to_scipy().data, ref["A"].data, RTOL_A, "A (closed-form + general cluster kernels)")
    warped(case.V.mesh)  # now every cluster is distorted: same matrix object, same form
    ref2 = oracle_outputs(oracle, case)
    assert abs(ref2["A"] - ref["A"]).max() > 1e-3
    dm.assemble_matrix(case.a, mpc, bcs=case.bcs, diagval=case.diagval, A=A)
    _close(A.to_scipy().data, ref2["A"].data, RTOL_A, "A after the mesh moved")
    b = dm.assemble_vector(case.L, mpc).numpy()
    _close(b, ref2["b"], RTOL_B, "b after the mesh moved")


def test_elasticity_clusters_closed_form_and_leftover_cells(oracle):
    """vector P1 elasticity on a mesh of parallelepiped clusters and distorted ones: the former go through the closed-form
    cluster kernel, the cells of the latter through the per-cell kernel; together the oracle's matrix"""
    import dolfinx_mpc_amd as dm
    from problems import case_cube_elasticity_slip, warped

    case = case_cube_elasticity_slip(6)
    warped(case.V.mesh, half=True)
    ref = oracle_outputs(oracle, case)
    mpc = product_mpc(case)
    A = dm.assemble_matrix(case.a, mpc, bcs=case.bcs, diagval=case.diagval)
    (parts, _keep, info), = [v[1] for v in A._plans[("objcache", "cubes")].values()]
    assert all(p[5] == 1 for p in parts) and 0 < info["clusters"] < 6 ** 3
    _close(A.to_scipy().data, ref["A"].data, RTOL_A, "A (closed-form elasticity clusters + leftover cells)")
    warped(case.V.mesh)  # every cluster distorted now: the per-cell kernel takes all cells
    ref2 = oracle_outputs(oracle, case)
    dm.assemble_matrix(case.a, mpc, bcs=case.bcs, diagval=case.diagval, A=A)
    _close(A.to_scipy().data, ref2["A"].data, RTOL_A, "A after the mesh moved")


def test_p2_clusters_closed_form_and_leftover_cells(oracle, monkeypatch):
    """scalar P2 stiffness (table entry p2_cube, not a default: slower than the per-cell row blocks): parallelepiped
    clusters through the closed-form cluster kernel (27 dofs, 393 coupled pairs per cluster), the cells of distorted
    clusters through the per-cell kernel; periodic slaves, Dirichlet values, lifting; then the mesh moves and every
    cluster is distorted"""
    import dolfinx_mpc_amd as dm
    from problems import warped

    monkeypatch.setenv("MPCX_FORCE_KERNEL", "matrix=p2_cube")

    for kwargs in (dict(reorder=(2, 2, 2)), dict(numbering="shuffled"), dict(warp="half", reorder=(3, 3, 3))):
        case = case_cube_periodic(6, 2, 0.3, **kwargs)
        ref = oracle_outputs(oracle, case)
        mpc = product_mpc(case)
        A = dm.assemble_matrix(case.a, mpc, bcs=case.bcs, diagval=case.diagval)
        (parts, _keep, info), = [v[1] for v in A._plans[("objcache", "cubes")].values()]
        assert parts[0][2] == 640 and parts[0][5] == 1
        assert info["clusters"] == 6 ** 3 if "warp" not in kwargs else 0 < info["clusters"] < 6 ** 3
        _close(A.to_scipy().data, ref["A"].data, RTOL_A, f"A (P2 cluster kernel, {kwargs})")
        out = product_outputs(case)
        _close(out["b_lifted"], ref["b_lifted"], RTOL_B, f"b_lifted ({kwargs})")
    warped(case.V.mesh)
    ref2 = oracle_outputs(oracle, case)
    dm.assemble_matrix(case.a, mpc, bcs=case.bcs, diagval=case.diagval, A=A)
    _close(A.to_scipy().data, ref2["A"].data, RTOL_A, "A after the mesh moved")


def test_cluster_vector_is_reproducible_without_device_atomics(oracle):
    """owner-computes cluster vector: every row of b gets its value from ONE workgroup (LDS adds) plus the halo sums
    gathered in a fixed order -- repeated assemblies agree to the last bits up to the order of the adds inside a
    block (<= 4 ulp of the largest entry; the hash kernel's device atomics are only bounded by the addend count)"""
    import dolfinx_mpc_amd as dm

    case = case_cube_periodic(12, 1, 0.0, reorder=(4, 4, 4))
    mpc = product_mpc(case)
    runs = [dm.assemble_vector(case.L, mpc).numpy().copy() for _ in range(5)]
    spread = np.max(np.abs(np.array(runs) - runs[0]))
    assert spread <= 4 * np.finfo(float).eps * np.abs(runs[0]).max()


@pytest.mark.parametrize("make", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_finalize_on_device_equals_host_finalize(make):
    """MultiPointConstraint.finalize on the device (mpcx_mpc_finalize_device: mark -> scan -> fill;
    mpcx_cell_to_slaves_device) against the host routines (cpp/MultiPointConstraint.h:36-126,
    cpp/mpc_helpers.h:19-94 restated in mpcx_host.cpp): every array bit for bit"""
    import dolfinx_mpc_amd as dm

    case = make()
    out = {}
    for where in ("host", "device"):
        m = dm.MultiPointConstraint(case.V)
        m.add_constraint(case.V, *case.raw)
        m.finalize(where=where)
        out[where] = m
    h, d = out["host"], out["device"]
    assert d._devt is not None and not d._host, "the device path must not have touched the host arrays yet"
    assert h.num_local_slaves == d.num_local_slaves and h.num_slaves == d.num_slaves
    for name in ("is_slave", "slaves"):
        assert np.array_equal(getattr(h, name), getattr(d, name)), name
    for name in ("masters", "owners", "cell_to_slaves"):
        assert np.array_equal(getattr(h, name).offsets, getattr(d, name).offsets), name
        assert np.array_equal(getattr(h, name).array, getattr(d, name).array), name
    assert np.array_equal(h.coefficients()[0], d.coefficients()[0])


def test_finalize_on_device_errors_and_duplicates():
    """out-of-range indices raise like the host routine; a dof listed twice as a slave takes the host routine
    (its sequential semantics cannot be reproduced in parallel)"""
    import dolfinx_mpc_amd as dm

    case = case_cube_periodic(3, 1, 0.0)
    V = case.V
    z = np.zeros(1, dtype=np.int32)
    for slaves, masters, msg in (([V.num_dofs + 3], [0], "slave index out of range"), ([2], [V.num_dofs + 5], "master index out of range")):
        m = dm.MultiPointConstraint(V)
        m.add_constraint(V, np.array(slaves, dtype=np.int32), np.array(masters, dtype=np.int64), np.ones(1), z, np.array([0, 1], dtype=np.int32))
        with pytest.raises(RuntimeError, match=msg):
            m.finalize(where="device")
    m = dm.MultiPointConstraint(V)
    m.add_constraint(V, np.array([5, 5], dtype=np.int32), np.array([1, 2], dtype=np.int64), np.array([0.5, 0.25]),
                     np.zeros(2, dtype=np.int32), np.array([0, 1, 2], dtype=np.int32))
    m.finalize(where="device")
    assert m._devt is None and m.num_slaves == 1  # fell back to the host routine


@pytest.mark.parametrize("async_streams", ["1", "0"])
def test_side_streams_keep_results_ordered(oracle, monkeypatch, async_streams):
    """assemble_matrix / assemble_vector run on the library's two side streams (la.side_stream) so that back-to-back
    calls overlap; whoever reads A / b next waits for them.  A driver loop that alternates two geometries and two
    coefficient values, lifts into b on the caller's stream and reads everything back each time must see exactly the
    values of a serial run (big enough that a missing wait would show)."""
    import torch

    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import fem

    monkeypatch.setenv("MPCX_ASYNC_STREAMS", async_streams)
    case = case_cube_periodic(40, 1, 0.7, reorder=(8, 8, 8))
    V = case.V
    mpc = product_mpc(case)
    w = fem.Function(V)
    L = fem.form_source(V, fem.FN_BENCH_PERIODIC, coefficient=w)  # per-cell path with a coefficient pack
    x0 = case.mesh.geometry.x.copy()
    x1 = x0 * np.array([1.0, 1.3, 0.8])
    A = b = b2 = None
    seen = {}
    for it in range(8):
        k = it % 2
        case.mesh.geometry.x = x1 if k else x0
        w.x.array[:] = 1.0 + k
        A = dm.assemble_matrix(case.a, mpc, bcs=case.bcs, A=A)
        b = dm.assemble_vector(case.L, mpc, b=b)
        b2 = dm.assemble_vector(L, mpc, b=b2)
        dm.apply_lifting(b, [case.a], [case.bcs], mpc)
        got = (A.vals.clone(), b.array.clone(), b2.array.clone())
        torch.cuda.synchronize()
        if k not in seen:
            seen[k] = got
        else:
            for u, v, name in zip(got, seen[k], ("A", "b lifted", "b with coefficient")):
                scale = float(v.abs().max())
                assert float((u - v).abs().max()) <= 1e-13 * scale, (it, name)
    # and the values themselves against the oracle for the last geometry / coefficient
    o_mpc = oracle_mpc(oracle, case)
    ref = oracle.assemble_matrix(case.a, o_mpc, bcs=case.bcs, fast=True)
    _close(A.to_scipy().data, ref.data, RTOL_A, "A after the loop")
    _close(b2.numpy(), oracle.assemble_vector(L, o_mpc), RTOL_B, "b with coefficient after the loop")
    assert float((seen[0][0] - seen[1][0]).abs().max()) > 0  # the two geometries really differ


def _table_names(which):
    from dolfinx_mpc_amd import dispatch

    return [k.name for k in (dispatch.MATRIX if which == "matrix" else dispatch.VECTOR)]


@pytest.mark.parametrize("name", _table_names("matrix"))
@pytest.mark.parametrize("make", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_dispatch_table_matrix_entries(oracle, make, name, monkeypatch):
    """every entry of the matrix dispatch table (dolfinx_mpc_amd/dispatch.py) forced wherever it applies
    (MPCX_FORCE_KERNEL=matrix=<name>; ignored where it does not), on every case"""
    monkeypatch.setenv("MPCX_FORCE_KERNEL", f"matrix={name}")
    case = make()
    if case.a is None:
        pytest.skip("no bilinear form")
    ref = oracle_outputs(oracle, case)
    out = product_outputs(case, algorithm="rowblock")
    assert np.array_equal(out["A"].indptr, ref["A"].indptr) and np.array_equal(out["A"].indices, ref["A"].indices)
    _close(out["A"].data, ref["A"].data, RTOL_A, f"{case.name} A [matrix={name}]")


@pytest.mark.parametrize("name", _table_names("vector"))
@pytest.mark.parametrize("make", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_dispatch_table_vector_entries(oracle, make, name, monkeypatch):
    monkeypatch.setenv("MPCX_FORCE_KERNEL", f"vector={name}")
    case = make()
    if case.L is None:
        pytest.skip("no linear form")
    ref = oracle_outputs(oracle, case)
    out = product_outputs(case, algorithm=None)
    for k in ("b", "b_lifted"):
        if k in ref:
            _close(out[k], ref[k], RTOL_B, f"{case.name} {k} [vector={name}]")


def test_forced_kernels_are_the_ones_that_run(monkeypatch):
    """MPCX_FORCE_KERNEL really selects the table entry where it applies, and the defaults are the measured ones"""
    import importlib

    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd.la import create_vector
    from problems import case_contact_two_body

    am = importlib.import_module("dolfinx_mpc_amd.assemble_matrix")
    av = importlib.import_module("dolfinx_mpc_amd.assemble_vector")

    def taken(case, which, force=None):
        if force:
            monkeypatch.setenv("MPCX_FORCE_KERNEL", f"{which}={force}")
        else:
            monkeypatch.delenv("MPCX_FORCE_KERNEL", raising=False)
        mpc = product_mpc(case)
        if which == "matrix":
            A = dm.create_matrix(case.a, mpc)
            args, keep = am.matrix_args(case.a, 0, A, mpc, mpc, case.bcs, 2)
        else:
            args, keep = av.vector_args(case.L, 0, create_vector(case.V), mpc, 0)
        return args.kernel_name

    p1 = case_cube_periodic(6, 1, 0.0, reorder=(2, 2, 2))
    assert taken(p1, "matrix") == "cube" and taken(p1, "vector") == "cube_own"
    for name in ("rowblock_lean", "rowblock", "rowpair"):
        assert taken(p1, "matrix", name) == (name if name != "rowpair" else "cube")  # rowpair does not apply to scalar P1
    for name in ("cube_hash", "ownblock", "rowblock", "hash"):
        assert taken(p1, "vector", name) == name
    p2 = case_cube_periodic(4, 2, 0.0, reorder=(2, 2, 2))
    # scalar P2 stiffness: pair records + cached contexts since round 4 (the thread-per-entity row blocks are its fall-back)
    assert taken(p2, "matrix") == "pairs" and taken(p2, "matrix", "rowblock") == "rowblock"
    assert taken(p2, "matrix", "rowpair") == "rowpair" and taken(p2, "matrix", "p2_cube") == "p2_cube"
    assert taken(p2, "vector") == "ownblock"
    el = case_contact_two_body(4, 6, 0.0, reorder=(2, 2, 2))
    # vector P1 elasticity on box meshes: parallelepiped clusters in closed form (leftover cells: rowpair)
    assert taken(el, "matrix") == "cube_el" and taken(el, "matrix", "rowblock") == "rowblock"
    assert taken(el, "matrix", "rowpair") == "rowpair"
    assert taken(el, "vector") == "rowblock"


def test_reproducibility_statement(oracle):
    """What repeated assembly of the same system guarantees (determinism):
    * pattern, plans and the master contributions (one thread per target position, fixed tuple order)
      are deterministic;
    * a row-block value is the sum of <= ~30 fp64 addends added with LDS atomics (ds_add_f64) in an
      order that depends on wave scheduling, the atomic kernel's with device atomics likewise: fp64
      addition is not associative, so two runs may differ in the last bits.  The spread is bounded by
      n * eps * sum|addends|; asserted here as <= 64 ulp of the row's largest entry (measured: 0-2 ulp).
    Bitwise equality run to run is therefore NOT promised by either algorithm (nor by the reference
    under MPI, where PETSc adds off-rank contributions in arrival order)."""
    import dolfinx_mpc_amd as dm

    case = case_cube_periodic(10, 1, 0.0, reorder=(4, 4, 4))
    mpc = product_mpc(case)
    for alg in ("rowblock", "atomic"):
        A = dm.assemble_matrix(case.a, mpc, bcs=case.bcs, algorithm=alg)
        runs = []
        for _ in range(5):
            dm.assemble_matrix(case.a, mpc, bcs=case.bcs, A=A, algorithm=alg)
            runs.append(A.to_scipy().data.copy())
        spread = np.max(np.abs(np.array(runs) - runs[0]), axis=0)
        scale = np.abs(runs[0]).max()
        assert spread.max() <= 64 * np.finfo(np.float64).eps * scale, (alg, spread.max())
    b_runs = [dm.assemble_vector(case.L, mpc).numpy().copy() for _ in range(3)]
    assert np.max(np.abs(b_runs[1] - b_runs[0])) <= 64 * np.finfo(np.float64).eps * np.abs(b_runs[0]).max()


# ---------------------------------------------------------------------------------------------
# values are read live, structure is cached per object (not per id())
# ---------------------------------------------------------------------------------------------
def test_lifting_reads_live_boundary_values(oracle):
    """A boundary Function updated between two apply_lifting calls (time-dependent condition,
    repeated LinearProblem.solve) must reach the kernel: the reference reads bc values on every call
    (cpp/lifting.h:166-180)."""
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import fem

    case = case_cube_periodic(4, 1, 0.0)
    V = case.V
    g = fem.Function(V)
    g.x.array[:] = 1.25
    bc = fem.dirichletbc(g, case.bcs[0].dof_indices()[0], V)
    mpc = product_mpc(case)
    o_mpc = oracle_mpc(oracle, case)
    for value in (1.25, -3.5, 0.75):
        g.x.array[:] = value
        b = dm.assemble_vector(case.L, mpc)
        dm.apply_lifting(b, [case.a], [[bc]], mpc)
        dm.set_bc(b, [bc])
        ref = oracle.assemble_vector(case.L, o_mpc)
        oracle.apply_lifting(ref, [case.a], [[bc]], o_mpc)
        ref[bc.dof_indices()[0]] = value
        _close(b.numpy(), ref, RTOL_B, f"b lifted with g = {value}")


def test_coefficients_and_constants_are_packed_per_call(oracle):
    """dolfinx packs coefficients and constants on every assembly call
    (cpp/assemble_matrix.cpp:583-589): changing them between calls changes the result."""
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import fem
    from problems import case_pipeline

    case = case_pipeline((1, 1))
    V = case.V
    w = fem.Function(V)
    c = fem.Constant(1.5)
    a = fem.form_stiffness(V, constant=c, coefficient=w)
    mpc = product_mpc(case)
    o_mpc = oracle_mpc(oracle, case)
    A = None
    for k, cval in enumerate((1.5, 0.25, 4.0)):
        w.interpolate(lambda x: 1.0 + k + np.sin(x[0] + k) * x[1])
        c.value[:] = cval
        A = dm.assemble_matrix(a, mpc, A=A)
        ref = oracle.assemble_matrix(a, o_mpc)
        _close(A.to_scipy().data, ref.data, RTOL_A, f"A with coefficient/constant set {k}")


def test_coefficient_written_through_a_kept_view_is_seen(oracle):
    """``w = f.x.array; w[:] = 1; assemble; w[:] = 2; assemble`` -- the second assembly must see the
    new values (the reference packs coefficients on every call, cpp/assemble_matrix.cpp:587-589)."""
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import fem
    from problems import case_pipeline

    case = case_pipeline((1, 1))
    V = case.V
    f = fem.Function(V)
    a = fem.form_stiffness(V, coefficient=f)
    L = fem.form_source(V, fem.FN_SIN2D, coefficient=f)
    mpc = product_mpc(case)
    o_mpc = oracle_mpc(oracle, case)
    view = f.x.array
    A, b = None, None
    for val in (1.0, 2.0, -0.5):
        view[:] = val + 0.1 * np.arange(V.num_dofs)
        A = dm.assemble_matrix(a, mpc, A=A)
        b = dm.assemble_vector(L, mpc, b=b)
        _close(A.to_scipy().data, oracle.assemble_matrix(a, o_mpc).data, RTOL_A, f"A, coefficient {val}")
        _close(b.numpy(), oracle.assemble_vector(L, o_mpc), RTOL_B, f"b, coefficient {val}")


@pytest.mark.parametrize("degree", [1, 2])
def test_moved_mesh_is_assembled_on_the_new_coordinates(oracle, degree):
    """cpp/assemble_matrix.cpp:495-501 gathers x on every call: after ``mesh.geometry.x = new`` the cached
    plans stay valid (topology unchanged) and the values follow the new geometry"""
    import dolfinx_mpc_amd as dm

    case = case_cube_periodic(4, degree, 0.0, reorder=(2, 2, 2))
    mpc = product_mpc(case)
    o_mpc = oracle_mpc(oracle, case)
    A = dm.assemble_matrix(case.a, mpc, bcs=case.bcs)
    b = dm.assemble_vector(case.L, mpc)
    x = case.mesh.geometry.x
    # an affine stretch plus a smooth interior wiggle that keeps the periodic faces matched
    new = x * np.array([1.0, 1.5, 0.75])
    new[:, 1] += 0.03 * np.sin(np.pi * x[:, 1]) * np.sin(np.pi * x[:, 2])
    case.mesh.geometry.x = new
    A = dm.assemble_matrix(case.a, mpc, bcs=case.bcs, A=A)
    b = dm.assemble_vector(case.L, mpc, b=b)
    ref_A = oracle.assemble_matrix(case.a, o_mpc, bcs=case.bcs)
    ref_b = oracle.assemble_vector(case.L, o_mpc)
    _close(A.to_scipy().data, ref_A.data, RTOL_A, "A on the moved mesh")
    _close(b.numpy(), ref_b, RTOL_B, "b on the moved mesh")
    assert abs(ref_A.data).max() > 0


def test_new_objects_never_hit_stale_caches(oracle):
    """Fresh DirichletBC / Form objects created in a loop (ids of collected objects get reused by
    CPython) must each be assembled with their OWN markers, masked dofmaps and plans."""
    import gc

    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import fem

    case = case_cube_periodic(5, 1, 0.0, reorder=(2, 2, 2))
    V = case.V
    mpc = product_mpc(case)
    o_mpc = oracle_mpc(oracle, case)
    A = dm.create_matrix(case.a, mpc)
    walls = [lambda x: np.isclose(x[1], 0), lambda x: np.isclose(x[2], 1), lambda x: np.isclose(x[1], 1) | np.isclose(x[2], 0)]
    for rep in range(6):
        marker = walls[rep % 3]
        # slaves may not carry a Dirichlet condition: keep the constraint's face x = 1 and its masters' x = 0 free
        dofs = fem.locate_dofs_geometrical(V, lambda x: marker(x) & ~np.isclose(x[0], 1) & ~np.isclose(x[0], 0))
        bc = fem.dirichletbc(0.5 + rep, dofs, V)
        a = fem.form_stiffness(V)  # a new form object every time
        for alg in ("rowblock", "atomic"):
            dm.assemble_matrix(a, mpc, bcs=[bc], A=A, algorithm=alg)
            ref = oracle.assemble_matrix(a, o_mpc, bcs=[bc], pattern=(A.rowptr, A.cols))
            _close(A.to_scipy().data, ref.data, RTOL_A, f"A rep {rep} {alg}")
        b = dm.assemble_vector(case.L, mpc)
        dm.apply_lifting(b, [a], [[bc]], mpc)
        ref_b = oracle.assemble_vector(case.L, o_mpc)
        oracle.apply_lifting(ref_b, [a], [[bc]], o_mpc)
        _close(b.numpy(), ref_b, RTOL_B, f"b rep {rep}")
        del bc, a
        gc.collect()


def test_reassembly_with_empty_first_integral(oracle):
    """A reused, non-zero matrix and a form whose FIRST integral is empty on this rank: the row-block
    path must still clear the old values (it skips the memset only when the first non-empty integral
    overwrites them)."""
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import fem

    case = case_cube_periodic(4, 1, 0.0)
    V = case.V
    mpc = product_mpc(case)
    A = dm.assemble_matrix(case.a, mpc, bcs=case.bcs, algorithm="rowblock")
    A.vals.fill_(7.0)  # stale values everywhere
    a2 = fem.form_stiffness(V, cells=np.zeros(0, dtype=np.int32)) + fem.form_stiffness(V, constant=2.0)
    dm.assemble_matrix(a2, mpc, bcs=case.bcs, A=A, algorithm="rowblock")
    ref = oracle.assemble_matrix(fem.form_stiffness(V, constant=2.0), oracle_mpc(oracle, case), bcs=case.bcs)
    _close(A.to_scipy().data, ref.data, RTOL_A, "A after an empty first integral")
    # only empty integrals: everything but the diagonal is cleared
    A.vals.fill_(7.0)
    dm.assemble_matrix(fem.form_stiffness(V, cells=np.zeros(0, dtype=np.int32)), mpc, bcs=case.bcs, A=A,
                       algorithm="rowblock")
    assert set(np.unique(A.to_scipy().data)) <= {0.0, 1.0}


def test_backsubstitution_accepts_a_function(oracle):
    """the reference's call shape: mpc.backsubstitution(uh) with a fem.Function
    (python/src/dolfinx_mpc/multipointconstraint.py:586-604)."""
    from dolfinx_mpc_amd import fem
    from problems import case_cube_contact_like

    case = case_cube_contact_like(3)
    mpc = product_mpc(case)
    u = fem.Function(case.V)
    rng = np.random.default_rng(3)
    u.x.array[:] = rng.standard_normal(case.V.num_dofs)
    ref = u.x.array.copy()
    oracle.backsubstitution(oracle_mpc(oracle, case), ref)
    mpc.backsubstitution(u)
    assert np.allclose(u.x.array, ref, rtol=0, atol=1e-14)
    mpc.homogenize(u)
    assert np.all(u.x.array[mpc.slaves] == 0.0)
    with pytest.raises(TypeError):
        mpc.backsubstitution(np.zeros(case.V.num_dofs))


@pytest.mark.parametrize("reorder", [None, (2, 2, 2)])
@pytest.mark.parametrize("scramble", [False, True])
def test_cluster_kernels_with_leftover_cells(oracle, reorder, scramble):
    """MPCX_ALG_CUBE on a mesh where some cubes are NOT clean fans (a few cells appear twice, so seven or eight
    tets sit round their long edge): those cells go through the per-cell kernels, the rest through the cluster
    kernels; matrix, vector and lifting must still match the oracle on the same mesh.  ``scramble``: cells
    shuffled and the local vertices of every cell permuted -- the detection must not depend on either."""
    import torch

    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import fem
    from dolfinx_mpc_amd.clusters import fans_from_topology, mesh_clusters_device
    from dolfinx_mpc_amd.mesh import Mesh, create_unit_cube
    from problems import Case, _walls_yz, periodic_raw

    base = create_unit_cube(6, 6, 6, reorder=reorder)
    cells = base.geometry.dofmap.copy()
    rng = np.random.default_rng(5)
    groups = rng.choice(cells.shape[0] // 6, size=40, replace=False)
    extra = np.concatenate([cells[6 * groups[:20] + 1], cells[6 * groups[20:] + 4], cells[6 * groups[20:30] + 2]], axis=0)
    cells = np.concatenate([cells, extra], axis=0)  # 40 cubes with 7 or 8 tets round their diagonal
    if scramble:
        cells = cells[rng.permutation(cells.shape[0])]
        for c in range(cells.shape[0]):
            cells[c] = cells[c][rng.permutation(4)]
    mesh = Mesh(base.geometry.x, cells, "tetrahedron")
    mesh.node_tile_offsets = base.node_tile_offsets
    verts, left = fans_from_topology(mesh.geometry.x, cells, cells.shape[0])
    assert left.size == 6 * 40 + 50 and verts.shape[0] == base.num_cells // 6 - 40
    d_verts, d_left = mesh_clusters_device(mesh, cells.shape[0])  # the HIP detection finds the same fans
    assert np.array_equal(np.sort(d_left), left)
    assert {tuple(sorted(r)) for r in d_verts.cpu().numpy().tolist()} == {tuple(sorted(r)) for r in verts.tolist()}
    V = fem.functionspace(mesh, ("Lagrange", 1))
    bc = fem.dirichletbc(0.4, fem.locate_dofs_geometrical(V, _walls_yz), V)
    case = Case("cluster_leftover", V, fem.form_stiffness(V, constant=1.7), fem.form_source(V, fem.FN_BENCH_PERIODIC), [bc],
                periodic_raw(V, [bc]))
    ref = oracle_outputs(oracle, case)
    mpc = product_mpc(case)
    A = dm.assemble_matrix(case.a, mpc, bcs=case.bcs, algorithm="rowblock")
    assert ("objcache", "cubes") in A._plans, "the cluster kernel was expected to run"
    S = A.to_scipy()
    assert np.array_equal(S.indptr, ref["A"].indptr) and np.array_equal(S.indices, ref["A"].indices)
    _close(S.data, ref["A"].data, RTOL_A, "A (clusters + leftover cells)")
    out_v = product_outputs(case, algorithm=None)  # vector: "auto" takes the cluster kernel for this form
    _close(out_v["b"], ref["b"], RTOL_B, "b (clusters + leftover cells)")
    _close(out_v["b_lifted"], ref["b_lifted"], RTOL_B, "b_lifted")
    torch.cuda.synchronize()
