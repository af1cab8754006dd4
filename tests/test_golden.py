"""Golden-vector tests.  tests/golden/*.npz were produced by
tests/golden/generate.py (oracle run in the build container; the reference
itself cannot run here).  CPU: the oracle reproduces its committed vectors
(guards against silent changes of the checker).  GPU (-m gpu): the HIP path
matches the same vectors."""

import os

import numpy as np
import pytest

from problems import all_small_cases, case_cube_periodic, oracle_outputs, product_outputs

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = all_small_cases()


def _load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def _check(out, g, rtol):
    if "A_data" in g:
        A = out["A"].tocsr()
        assert np.array_equal(A.indptr, g["A_indptr"]) and np.array_equal(A.indices, g["A_indices"])
        scale = max(1.0, abs(g["A_data"]).max())
        assert abs(A.data - g["A_data"]).max() <= rtol * scale
    for k in ("b", "b_lifted"):
        if k in g:
            assert abs(out[k] - g[k]).max() <= rtol * max(1.0, abs(g[k]).max())


def _check_sums(out, g, rtol):
    A = out["A"].tocsr()
    n = A.shape[0]
    assert A.nnz == int(g["nnz"])
    rng = np.random.default_rng(1234)
    v = rng.standard_normal(n)
    idx = g["sample_idx"]
    assert np.allclose(A.data[idx], g["sample_val"], rtol=0, atol=rtol * abs(g["sample_val"]).max())
    assert np.sqrt((A.data**2).sum()) == pytest.approx(float(g["frob"]), rel=rtol)
    assert np.allclose((A @ v)[:: max(1, n // 256)], g["Av"], rtol=0, atol=rtol * abs(g["Av"]).max() * 10)
    assert A.diagonal().sum() == pytest.approx(float(g["diag_sum"]), rel=rtol)
    for k in ("b", "b_lifted"):
        assert np.linalg.norm(out[k]) == pytest.approx(float(g[k + "_norm"]), rel=rtol)
        assert np.allclose(out[k][:: max(1, n // 256)], g[k + "_sample"], rtol=0, atol=rtol * abs(g[k + "_sample"]).max())


@pytest.mark.parametrize("make", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_oracle_reproduces_golden(oracle, make):
    case = make()
    _check(oracle_outputs(oracle, case), _load(case.name), 1e-14)


def test_oracle_config1_checksums(oracle):
    case = case_cube_periodic(32, 1, 0.0)
    _check_sums(oracle_outputs(oracle, case, fast=True), _load("config1_cube32_checksums"), 1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("alg", ["atomic", "rowblock"])
@pytest.mark.parametrize("make", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_gpu_matches_golden(make, alg):
    case = make()
    _check(product_outputs(case, algorithm=alg), _load(case.name), 1e-12)


@pytest.mark.gpu
def test_gpu_config1_checksums():
    case = case_cube_periodic(32, 1, 0.0)
    _check_sums(product_outputs(case, algorithm="rowblock"), _load("config1_cube32_checksums"), 1e-12)
