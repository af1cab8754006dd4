"""Pin the oracle with the reference's own verification identities
(python/src/dolfinx_mpc/utils/test.py:202-265):

    A_mpc[free, free] == K^T A_unconstrained K        (atol 5e3 * resolution = 5e-12)
    b_mpc[slaves] == 0,  b_mpc[free] == K^T b_unconstrained

on the configurations of the reference test-suite (tests/problems.py), plus the
scipy solve / back-substitution recipe of python/tests/test_mpc_pipeline.py:99-110.
CPU only.
"""

import numpy as np
import pytest
import scipy.sparse.linalg

from problems import all_small_cases, oracle_mpc, oracle_outputs

CASES = all_small_cases()


@pytest.mark.parametrize("make", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_lhs_rhs_identities(oracle, make):
    po = oracle
    case = make()
    mpc = oracle_mpc(po, case)
    emp = po.OracleMPC.empty(case.V)
    out = oracle_outputs(po, case)
    scale = 1.0
    if case.a is not None:
        A_org = po.assemble_matrix(case.a, emp, bcs=case.bcs, diagval=case.diagval)
        scale = max(1.0, abs(A_org).max())
        po.compare_mpc_lhs(A_org, out["A"], mpc, atol=5e3 * np.finfo(np.float64).resolution * scale)
        # slave rows/cols hold only diagval on the diagonal
        sl = mpc.slaves[: mpc.num_local_slaves]
        A = out["A"].tocsr()
        assert np.allclose(A.diagonal()[sl], case.diagval)
        assert abs(A[sl]).sum() == pytest.approx(abs(case.diagval) * sl.size)
        assert abs(A[:, sl]).sum() == pytest.approx(abs(case.diagval) * sl.size)
    if case.L is not None:
        b_org = po.assemble_vector(case.L, emp)
        po.compare_mpc_rhs(b_org, out["b"], mpc)
        if "b_lifted" in out:
            x0 = None if case.x0 is None else [case.x0]
            po.apply_lifting(b_org, [case.a], [case.bcs], emp, x0=x0, scale=case.scale)
            po.compare_mpc_rhs(b_org, out["b_lifted"], mpc)


@pytest.mark.parametrize("make", [CASES[6], CASES[7], CASES[9], CASES[12], CASES[15], CASES[17], CASES[19], CASES[20]],
                         ids=["lifting", "pipeline", "vector_poisson", "surface", "cube", "cube_bc", "slip", "contact"])
def test_solution_matches_reduced_system(oracle, make):
    """u_mpc (solve A_mpc u = b_mpc, back-substitute) == K (K^T A K)^-1 K^T b
    (python/tests/test_mpc_pipeline.py:99-110, test_lifting.py:103-122)."""
    po = oracle
    case = make()
    mpc = oracle_mpc(po, case)
    emp = po.OracleMPC.empty(case.V)
    out = oracle_outputs(po, case)
    b = out.get("b_lifted", out["b"]).copy()
    g = np.zeros(case.V.num_dofs)
    for bc in case.bcs:  # set_bc
        vals = np.zeros_like(b)
        bc.set(vals)
        d = bc.dof_indices()[0]
        b[d] = vals[d]
        g[d] = vals[d]
    if not case.bcs and case.name.startswith(("pipeline", "square")):
        pytest.skip("pure Neumann problem is singular")
    u = scipy.sparse.linalg.spsolve(out["A"].tocsc(), b)
    po.backsubstitution(mpc, u)

    A_org = po.assemble_matrix(case.a, emp, bcs=case.bcs)
    b_org = po.assemble_vector(case.L, emp)
    if case.bcs:
        po.apply_lifting(b_org, [case.a], [case.bcs], emp)
        for bc in case.bcs:
            d = bc.dof_indices()[0]
            b_org[d] = g[d]
    K = po.gather_transformation_matrix(mpc)
    d_red = scipy.sparse.linalg.spsolve((K.T @ A_org @ K).tocsc(), K.T @ b_org)
    u_ref = K @ d_red
    assert np.allclose(u, u_ref, rtol=500 * np.finfo(np.float64).resolution, atol=1e-10 * max(1, abs(u_ref).max()))


def test_backsubstitution_and_homogenize(oracle):
    po = oracle
    from problems import case_cube_contact_like

    case = case_cube_contact_like(3)
    mpc = oracle_mpc(po, case)
    rng = np.random.default_rng(0)
    u = rng.standard_normal(case.V.num_dofs)
    v = u.copy()
    po.backsubstitution(mpc, v)
    K = po.gather_transformation_matrix(mpc)
    free = np.flatnonzero(mpc.is_slave == 0)
    assert np.allclose(v, K @ u[free])
    po.homogenize(mpc, u)
    assert np.all(u[mpc.slaves] == 0)
