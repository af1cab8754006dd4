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
    assert rp_d.dtype == np.int32 and cols_d.dtype == np.int32
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
