"""Imported UFCx kernels in FFCx's shape (tools/ffcx_like.py) on every element the backend has.

CPU: the oracle calls the gcc-compiled text through the function pointer (cpp/assemble_matrix.cpp:438-439) and
must reproduce the built-in operators on all 27 small cases -- this pins generator and tables.
GPU (-m gpu): the product runs the same text inside the LDS row-block kernels (and the thread-per-entity atomic
kernels) and must match the built-in oracle on all cases, both algorithms."""

import numpy as np
import pytest

from problems import all_small_cases, oracle_outputs, product_outputs
from ufcx_twin import num_imported, twin_case

CASES = all_small_cases()


@pytest.mark.parametrize("make", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_oracle_imported_kernels_reproduce_builtin_operators(oracle, make):
    case = make()
    twin = twin_case(case)
    if num_imported(twin) == 0:
        pytest.skip("no cell integral the generator covers")
    ref = oracle_outputs(oracle, case)
    out = oracle_outputs(oracle, twin)
    for k in ref:
        r, o = (ref[k].toarray(), out[k].toarray()) if k == "A" else (ref[k], out[k])
        assert abs(o - r).max() <= 1e-12 * max(1.0, abs(r).max()), f"{case.name} {k}"


@pytest.mark.parametrize("make", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_oracle_reads_generated_kernels_out_of_ffcx_layout_files(oracle, make):
    """the same kernels inside whole FFCx-layout files (codegen.ffcx_file: include block, ufcx_integral / ufcx_form objects,
    alias): the oracle compiles the file against oracle/include/ufcx.h and takes the kernel out of the compiled objects"""
    case = make()
    twin = twin_case(case, layout="ffcx")
    if num_imported(twin) == 0:
        pytest.skip("no cell integral the generator covers")
    ref = oracle_outputs(oracle, case)
    out = oracle_outputs(oracle, twin)
    for k in ref:
        r, o = (ref[k].toarray(), out[k].toarray()) if k == "A" else (ref[k], out[k])
        assert abs(o - r).max() <= 1e-12 * max(1.0, abs(r).max()), f"{case.name} {k}"


def test_include_lines_are_dropped_and_functions_inlined():
    """FFCx output starts with #include <math.h> / <stdint.h> / <ufcx.h>: hipRTC has no such headers, the compile
    step drops the lines; helper functions in the text are fine (everything defined there becomes a device
    function)"""
    from dolfinx_mpc_amd import _native

    src = """#include <math.h>
  #include <stdint.h>
#include <ufcx.h>
static double helper(double v) { return 2.0 * v; }
void tt(double* restrict A, const double* restrict w, const double* restrict c, const double* restrict coordinate_dofs,
        const int* restrict entity_local_index, const uint8_t* restrict quadrature_permutation, void* custom_data)
{ for (int i = 0; i < 3; ++i) A[i] += helper(sqrt(fabs(coordinate_dofs[i]))); }
"""
    L = _native.lib()
    h = L.mpcx_ufcx_compile(_native.UfcxDescT(src.encode(), b"tt", 1, 3, 1, 0, 0, 3))
    assert h, L.mpcx_last_error().decode()
    L.mpcx_ufcx_free(h)


@pytest.mark.gpu
@pytest.mark.parametrize("alg", ["atomic", "rowblock", "auto"])
@pytest.mark.parametrize("make", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_gpu_imported_kernels_match_builtin_oracle(oracle, make, alg):
    """"rowblock" takes the first applicable entry of the matrix table (scalar P1 on tetrahedra: the cluster kernel round the
    imported function, ufcx_cube), "auto" also the vector table's (ufcx_cube_own)"""
    case = make()
    twin = twin_case(case)
    if num_imported(twin) == 0:
        pytest.skip("no cell integral the generator covers")
    ref = oracle_outputs(oracle, case)
    out = product_outputs(twin, algorithm=None if alg == "auto" else alg)
    if "A" in ref:
        assert np.array_equal(out["A"].indptr, ref["A"].indptr) and np.array_equal(out["A"].indices, ref["A"].indices)
        assert abs(out["A"].data - ref["A"].data).max() <= 1e-12 * max(1.0, abs(ref["A"]).max()), case.name + " A"
    for k in ("b", "b_lifted"):
        if k in ref:
            assert abs(out[k] - ref[k]).max() <= 1e-12 * max(1.0, abs(ref[k]).max()), f"{case.name} {k}"


@pytest.mark.gpu
@pytest.mark.parametrize("env", ["MPCX_UFCX_PAIRS=0", "MPCX_UFCX_ROWWISE=0", "MPCX_UFCX_RB_THREADS=128", "MPCX_FORCE_KERNEL=matrix=ufcx_pairs"])
@pytest.mark.parametrize("make", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_gpu_imported_kernels_rowwise_variants(oracle, make, env, monkeypatch):
    """round 6: element tensors of more than 36 entries on simplices run ROW-WISE copies of the text, by default from pair
    records (ufcx_matrix_pairs_kernel); the other routes stay tested: row-wise copies inside the per-cell row blocks
    (MPCX_UFCX_PAIRS=0), the whole tensor per visit (MPCX_UFCX_ROWWISE=0), another workgroup size"""
    k, v = env.split("=", 1)
    monkeypatch.setenv(k, v)
    case = make()
    twin = twin_case(case)
    if num_imported(twin) == 0 or twin.a is None:
        pytest.skip("no bilinear cell integral the generator covers")
    ref = oracle_outputs(oracle, case)
    out = product_outputs(twin, algorithm="rowblock")
    assert abs(out["A"].data - ref["A"].data).max() <= 1e-12 * max(1.0, abs(ref["A"]).max()), case.name + " A"


@pytest.mark.gpu
def test_gpu_pair_records_are_the_default_for_imported_p2(oracle):
    """the dispatch takes ufcx_pairs for a scalar P2 stiffness text (100 entries) and ufcx_cube for P1"""
    import importlib

    import dolfinx_mpc_amd as dm
    from problems import case_cube_periodic, product_mpc

    am = importlib.import_module("dolfinx_mpc_amd.assemble_matrix")
    for degree, want in ((2, "ufcx_pairs"), (1, "ufcx_cube")):
        twin = twin_case(case_cube_periodic(4, degree, 0.0, reorder=(2, 2, 2)))
        mpc = product_mpc(twin)
        A = dm.create_matrix(twin.a, mpc)
        ma, _k = am.matrix_args(twin.a, 0, A, mpc, mpc, twin.bcs, 2, 1)
        assert ma.kernel_name == want, (degree, ma.kernel_name)


@pytest.mark.gpu
@pytest.mark.parametrize("alg", ["atomic", "auto"])
@pytest.mark.parametrize("make", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_gpu_imported_ffcx_layout_files_match_builtin_oracle(oracle, make, alg):
    """whole FFCx-layout files through mpcx_ufcx_compile (objects read, blanked, the alias resolved) on both algorithms"""
    case = make()
    twin = twin_case(case, layout="ffcx")
    if num_imported(twin) == 0:
        pytest.skip("no cell integral the generator covers")
    ref = oracle_outputs(oracle, case)
    out = product_outputs(twin, algorithm=None if alg == "auto" else alg)
    if "A" in ref:
        assert abs(out["A"].data - ref["A"].data).max() <= 1e-12 * max(1.0, abs(ref["A"]).max()), case.name + " A"
    for k in ("b", "b_lifted"):
        if k in ref:
            assert abs(out[k] - ref[k]).max() <= 1e-12 * max(1.0, abs(ref[k]).max()), f"{case.name} {k}"


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["MPCX_NO_MPC_PLAN", "MPCX_VECTOR_OWNER=0", "MPCX_PLAN_LISTS=host", "MPCX_SLAVE_TENSORS=0"])
@pytest.mark.parametrize("idx", [2, 8, 15, 19, 20, 23])
def test_gpu_imported_kernels_plan_variants(oracle, monkeypatch, idx, variant):
    """the imported kernels on the other routes of the row-block path: master contributions without a plan (what a
    bare C-ABI caller gets), halo-recomputing vector row blocks, host-built entity lists"""
    key, _, val = variant.partition("=")
    monkeypatch.setenv(key, val or "1")
    case = CASES[idx]()
    twin = twin_case(case)
    ref = oracle_outputs(oracle, case)
    out = product_outputs(twin, algorithm="rowblock")
    assert abs(out["A"].data - ref["A"].data).max() <= 1e-12 * max(1.0, abs(ref["A"]).max())
    for k in ("b", "b_lifted"):
        if k in ref:
            assert abs(out[k] - ref[k]).max() <= 1e-12 * max(1.0, abs(ref[k]).max()), k


@pytest.mark.gpu
@pytest.mark.parametrize("degree", [1, 2])
def test_stated_builtin_twin_is_checked_then_used(oracle, degree, monkeypatch):
    """fem.form_generated hands the text over WITH the built-in operator it implements: the statement is checked on sample
    cells on the device and the operator then stands in for the text; a wrong statement is dropped and the text runs;
    MPCX_UFCX_BUILTIN=0 never substitutes.  All three give the oracle's result (which runs the text through gcc)."""
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import fem
    from problems import Case, case_cube_periodic, oracle_outputs, product_outputs

    base = case_cube_periodic(4, degree, 0.3, reorder=(2, 2, 2))
    V = base.V

    def forms():
        return fem.form_generated("stiffness", V), fem.form_generated("source", V, fem.FN_BENCH_PERIODIC)

    def check(a, L, want_builtin):
        ref = oracle_outputs(oracle, Case("twin", V, *forms(), base.bcs, base.raw))
        out = product_outputs(Case("twin", V, a, L, base.bcs, base.raw), algorithm="rowblock")
        assert (a.integrals[0].kernel.form != fem.FORM_UFCX) == want_builtin and (L.integrals[0].kernel.form != fem.FORM_UFCX) == want_builtin
        assert abs(out["A"].data - ref["A"].data).max() <= 1e-12 * max(1.0, abs(ref["A"].data).max())
        assert abs(out["b_lifted"] - ref["b_lifted"]).max() <= 1e-12 * max(1.0, abs(ref["b_lifted"]).max())

    a, L = forms()
    check(a, L, True)
    # a wrong statement: the stiffness text with the MASS operator as its twin, the source text with another function
    a, L = forms()
    a.integrals[0].kernel.builtin = fem.form_mass(V).integrals[0].kernel
    L.integrals[0].kernel.builtin = fem.form_source(V, fem.FN_POLY3).integrals[0].kernel
    check(a, L, False)
    monkeypatch.setenv("MPCX_UFCX_BUILTIN", "0")
    a, L = forms()
    check(a, L, False)


def test_code_object_cache(tmp_path, monkeypatch):
    """MPCX_UFCX_CACHE: the second compilation of the same kernel is a file read (hipRTC cross-compiles on the CPU)"""
    import time

    from dolfinx_mpc_amd import _device as D
    from dolfinx_mpc_amd import _native, fem
    from dolfinx_mpc_amd.mesh import create_unit_square

    monkeypatch.setenv("MPCX_UFCX_CACHE", str(tmp_path))
    V = fem.functionspace(create_unit_square(2, 2, "quadrilateral"), ("Lagrange", 2))
    form = fem.form_mass(V, constant=1.2345)
    k = form.integrals[0].kernel
    monkeypatch.setattr(D, "_ufcx_handles", {})
    t0 = time.perf_counter()
    h1 = D.ufcx_compile(k, form)
    t_compile = time.perf_counter() - t0
    files = list(tmp_path.glob("*.co"))
    assert len(files) == 1 and files[0].stat().st_size == _native.lib().mpcx_ufcx_code_size(h1)
    monkeypatch.setattr(D, "_ufcx_handles", {})
    t0 = time.perf_counter()
    h2 = D.ufcx_compile(k, form)
    t_cached = time.perf_counter() - t0
    L = _native.lib()
    import ctypes as C

    n = L.mpcx_ufcx_code_size(h2)
    b1, b2 = C.create_string_buffer(n), C.create_string_buffer(n)
    L.mpcx_ufcx_code(h1, b1)
    L.mpcx_ufcx_code(h2, b2)
    assert b1.raw == b2.raw and t_cached < 0.25 * t_compile, (t_compile, t_cached)
