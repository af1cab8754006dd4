"""Co-run launches (dolfinx_mpc_amd/corun.py): a row-block matrix launch cut in two sub-ranges of its row blocks -- the
first with an LDS floor (mpcx_matrix_args_t::lds_floor), the second without -- must give the matrix of the single launch,
and a vector launch with a floor the same vector.  The sub-range trick (plan.block_row0 / block_ent_off advanced) is
exercised on every row-block family: entity lists, pair records, node blocks (CSR-valued and block-scalar), row pairs,
imported kernels.  Reference loops: cpp/assemble_matrix.cpp:488-547, cpp/assemble_vector.cpp:65-90."""

import sys

import numpy as np
import pytest

from problems import all_small_cases, oracle_outputs, product_outputs

pytestmark = pytest.mark.gpu

CASES = all_small_cases()


def _close(got, ref, rtol, what):
    scale = max(1.0, abs(ref).max())
    d = abs(got - ref).max()
    assert d <= rtol * scale, f"{what}: max diff {d:.3e} > {rtol * scale:.3e}"


@pytest.fixture
def small_blocks(monkeypatch):
    """row blocks of at most 24 rows so that the small cases have several of them; every launch with >= 2 blocks is cut"""
    import dolfinx_mpc_amd  # noqa: F401

    am = sys.modules["dolfinx_mpc_amd.assemble_matrix"]
    monkeypatch.setattr(am, "ROWBLOCK_MAX_ROWS", 24)
    monkeypatch.setattr(am, "ROWBLOCK_LIGHT_MAX_ROWS", 24)
    monkeypatch.setenv("MPCX_PAIRS_MAX_ROWS", "24")
    monkeypatch.setenv("MPCX_NO_CUBE", "1")
    monkeypatch.setenv("MPCX_CORUN", "1")
    monkeypatch.setenv("MPCX_CORUN_MIN_BLOCKS", "2")
    monkeypatch.setenv("MPCX_CORUN_VECTOR_FLOOR", "50000")


@pytest.mark.parametrize("frac", ["0.3", "0.7"])
@pytest.mark.parametrize("make", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_split_launches_match_oracle(oracle, make, frac, small_blocks, monkeypatch):
    monkeypatch.setenv("MPCX_CORUN_FRAC", frac)
    case = make()
    ref = oracle_outputs(oracle, case)
    out = product_outputs(case, algorithm="rowblock")
    if "A" in ref:
        assert np.array_equal(out["A"].indptr, ref["A"].indptr)
        assert np.array_equal(out["A"].indices, ref["A"].indices)
        _close(out["A"].data, ref["A"].data, 1e-12, case.name + " A")
    for k in ("b", "b_lifted"):
        if k in ref:
            _close(out[k], ref[k], 1e-12, f"{case.name} {k}")


@pytest.mark.parametrize("env", ["MPCX_BLOCK_SCALAR=0", "MPCX_FORCE_KERNEL=matrix=rowpair", "MPCX_FORCE_KERNEL=matrix=pairs",
                                 "MPCX_NO_NODEBLOCK=1"])
@pytest.mark.parametrize("make", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_split_launches_kernel_families(oracle, make, env, small_blocks, monkeypatch):
    k, v = env.split("=", 1)
    monkeypatch.setenv(k, v)
    case = make()
    ref = oracle_outputs(oracle, case)
    out = product_outputs(case, algorithm="rowblock")
    if "A" in ref:
        _close(out["A"].data, ref["A"].data, 1e-12, case.name + " A")


def test_split_really_cuts(small_blocks):
    """the hook is live: a matrix with several row blocks is launched in two parts, the first one with the floor"""
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import _native, corun
    from problems import case_cube_periodic, product_mpc

    case = case_cube_periodic(6, 2, 0.0)
    mpc = product_mpc(case)
    A = dm.create_matrix(case.a, mpc)
    am = sys.modules["dolfinx_mpc_amd.assemble_matrix"]
    a, keep = am.matrix_args(case.a, 0, A, mpc, mpc, case.bcs, 2, 1)
    assert corun.splittable(a)
    parts = corun.split(a, 0.5, corun.matrix_floor(2))
    assert len(parts) == 2 and parts[0].lds_floor > 160 * 1024 // 3 and parts[1].lds_floor == 0
    assert parts[0].plan.num_blocks + parts[1].plan.num_blocks == a.plan.num_blocks
    assert parts[0].n_slave_entities == 0 and parts[1].n_slave_entities == a.n_slave_entities
    assert isinstance(parts[0], _native.MatrixArgs)
    del keep
