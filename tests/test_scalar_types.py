"""Scalar types other than fp64-real (VERDICT r3 "missing" item 1): the T-generic oracle (oracle/scalar_oracle.py) pinned on
the CPU, the HIP path (csrc/mpcx_scalar.hip) against it on the GPU.

CPU: (1) in float64 the T-generic restatement IS the C oracle on all 27 small cases (same statements, different
language); (2) in complex128 it satisfies the reference's identity with the HERMITIAN transpose,
A_mpc[free, free] == K^H A K and b_mpc[free] == K^H b (python/src/dolfinx_mpc/utils/test.py:202-265 with K complex):
a transposed-but-not-conjugated row side fails it.
GPU (-m gpu): all 27 cases in complex128, complex64 and float32, both algorithms, against the T-generic oracle.  Tolerances: complex128 as
fp64 (1e-12 of the largest entry); float32 / complex64: the tensor is computed in fp64 and every scatter-add rounds to
fp32 -- <= ~30 addends per entry -- 2e-5 of the largest entry (fp32 epsilon 1.2e-7 x addends x safety)."""

import numpy as np
import pytest

from problems import all_small_cases, oracle_mpc, oracle_outputs
from scalar_cases import oracle_outputs_scalar, product_outputs_scalar, retype, scalar_mpc

CASES = all_small_cases()
IDS = [f"case{i}" for i in range(len(CASES))]


@pytest.fixture(scope="module")
def so():
    from oracle import scalar_oracle

    return scalar_oracle


@pytest.mark.parametrize("make", CASES, ids=IDS)
def test_generic_oracle_equals_c_oracle_in_float64(oracle, so, make):
    case = make()
    ref = oracle_outputs(oracle, case)
    got = oracle_outputs_scalar(so, retype(make(), np.float64))
    if "A" in ref:
        assert np.array_equal(got["A"].indptr, ref["A"].indptr) and np.array_equal(got["A"].indices, ref["A"].indices)
        assert abs(got["A"].data - ref["A"].data).max() <= 1e-13 * max(1.0, abs(ref["A"].data).max())
    for k in ("b", "b_lifted"):
        if k in ref:
            assert abs(got[k] - ref[k]).max() <= 1e-13 * max(1.0, abs(ref[k]).max()), k


def _K(mpc, n):
    """global transformation matrix K (n x n_free columns kept as n x n with slave columns empty): u = K u_free"""
    import scipy.sparse

    rows, cols, vals = [], [], []
    for d in range(n):
        if mpc.is_slave[d]:
            for (m, c) in mpc.links(d):
                rows.append(d), cols.append(m), vals.append(c)
        else:
            rows.append(d), cols.append(d), vals.append(1.0)
    return scipy.sparse.csr_matrix((np.array(vals, dtype=np.complex128), (rows, cols)), shape=(n, n))


@pytest.mark.parametrize("make", [CASES[i] for i in (0, 4, 6, 9, 15, 17, 20, 21, 24)],
                         ids=[IDS[i] for i in (0, 4, 6, 9, 15, 17, 20, 21, 24)])
def test_complex_oracle_satisfies_the_hermitian_identity(so, make):
    case = retype(make(), np.complex128)
    if case.a is None or case.a.function_spaces[0] is not case.a.function_spaces[1]:
        pytest.skip("square forms")
    mpc = scalar_mpc(so, case)
    empty = so.ScalarMPC(case.V, np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int64), np.zeros(0), np.zeros(0, dtype=np.int32),
                         np.zeros(1, dtype=np.int32), np.complex128)
    n = case.V.num_dofs
    K = _K(mpc, n)
    A = so.assemble_matrix(case.a, mpc, bcs=case.bcs, diagval=case.diagval)
    A0 = so.assemble_matrix(case.a, empty, bcs=case.bcs, diagval=case.diagval)
    red = (K.conj().T @ A0 @ K).toarray()
    free = ~mpc.is_slave
    got = A.toarray()
    scale = max(1.0, abs(red).max())
    assert abs(got[np.ix_(free, free)] - red[np.ix_(free, free)]).max() <= 5e-12 * scale
    # the plain transpose is NOT what the path computes when the coefficients are complex
    wrong = (K.T @ A0 @ K).toarray()
    if any(abs(c.imag) > 0 for d in mpc.slaves for (_m, c) in mpc.links(d)):
        assert abs(got[np.ix_(free, free)] - wrong[np.ix_(free, free)]).max() > 1e-6 * scale
    if case.L is not None:
        b = so.assemble_vector(case.L, mpc)
        b0 = so.assemble_vector(case.L, empty)
        assert abs(b[free] - (K.conj().T @ b0)[free]).max() <= 5e-12 * max(1.0, abs(b0).max())
        assert abs(b[mpc.slaves]).max() == 0 if mpc.slaves.size else True


TOL = {"complex128": 1e-12, "complex64": 2e-5, "float32": 2e-5}


@pytest.mark.gpu
@pytest.mark.parametrize("alg", ["atomic", "rowblock"])
@pytest.mark.parametrize("dtype", ["complex128", "complex64", "float32"])
@pytest.mark.parametrize("make", CASES, ids=IDS)
def test_gpu_scalar_types_match_the_generic_oracle(so, make, dtype, alg):
    """both algorithms of csrc/mpcx_scalar.hip: per-entity device atomics and LDS row blocks"""
    case = retype(make(), np.dtype(dtype))
    ref = oracle_outputs_scalar(so, retype(make(), np.complex128 if dtype.startswith("complex") else np.float64))
    out = product_outputs_scalar(case, algorithm=alg)
    tol = TOL[dtype]
    if "A" in ref:
        assert out["A"].dtype == np.dtype(dtype)
        assert np.array_equal(out["A"].indptr, ref["A"].indptr) and np.array_equal(out["A"].indices, ref["A"].indices)
        assert abs(out["A"].data - ref["A"].data).max() <= tol * max(1.0, abs(ref["A"].data).max())
    for k in ("b", "b_lifted"):
        if k in ref:
            assert out[k].dtype == np.dtype(dtype)
            assert abs(out[k] - ref[k]).max() <= tol * max(1.0, abs(ref[k]).max()), k


@pytest.mark.gpu
def test_gpu_complex_backsubstitution_and_errors(so):
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import fem
    from dolfinx_mpc_amd.la import Vector
    import torch

    case = retype(CASES[15](), np.complex128)
    out = product_outputs_scalar(case)
    mpc = out["mpc"]
    rng = np.random.default_rng(0)
    u = rng.standard_normal(case.V.num_dofs) + 1j * rng.standard_normal(case.V.num_dofs)
    v = Vector(u.size, dtype=np.complex128)
    v.array.copy_(torch.from_numpy(u))
    mpc.backsubstitution(v)
    ref = so.backsubstitution(scalar_mpc(so, case), u.copy())
    assert abs(v.numpy() - ref).max() < 1e-14
    mpc.homogenize(v)
    assert np.all(v.numpy()[mpc.slaves] == 0)
    # a real constraint with a complex form, and the fp64-only algorithm, are refused
    real = dm.MultiPointConstraint(case.V)
    real.finalize()
    with pytest.raises(ValueError):
        dm.assemble_matrix(case.a, real, bcs=case.bcs)
    gen = fem.form_generated("stiffness", case.V)
    k = gen.integrals[0].kernel
    fu = fem.form_ufcx([case.V, case.V], k.ufcx_source, k.ufcx_name).set_dtype(np.complex128)  # imported text: fp64-real only
    with pytest.raises((NotImplementedError, RuntimeError)):
        dm.assemble_matrix(fu, mpc, bcs=case.bcs)
    with pytest.raises(NotImplementedError):
        dm.MultiPointConstraint(case.V, dtype=np.int32)
