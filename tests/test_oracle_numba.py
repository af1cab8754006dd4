"""The two restatements of the reference's constrained assembly -- oracle/mpc_oracle.c after the C++
assemblers (cpp/assemble_matrix.cpp:99-268, 417-548), oracle/numba_oracle.py after the numba ones
(python/src/dolfinx_mpc/numba/assemble_matrix.py:216-449: unconstrained pass + correction over the slave
entities) -- must produce the same matrix and vector on every configuration of tests/problems.py.
north_star names both assemblers as parity targets; agreement of two differently structured statements
rules out a shared misreading of modify_mpc_cell (it does not replace a run of the reference)."""

import numpy as np
import pytest

from problems import all_small_cases, oracle_mpc

CASES = all_small_cases()


@pytest.mark.parametrize("make", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_numba_structure_equals_cpp_structure(oracle, make):
    from oracle import numba_oracle as no

    case = make()
    mpc = oracle_mpc(oracle, case)
    if case.a is not None:
        A_cpp = oracle.assemble_matrix(case.a, mpc, bcs=case.bcs, diagval=case.diagval)
        A0, corr = no.assemble_matrix(case.a, mpc, bcs=case.bcs, diagval=case.diagval)
        A_nb = (A0 + corr).tocsr()
        scale = max(1.0, abs(A_cpp).max())
        assert abs(A_nb - A_cpp).max() <= 1e-13 * scale
        # whatever the correction pass inserts outside the MPC pattern is an explicit zero: the numba version writes
        # 0-valued (other slave, master) entries (numba/assemble_matrix.py:396-409, mpc_dofs keeps the other slaves)
        import scipy.sparse

        pat = oracle.create_pattern(case.a, mpc, mpc)
        P = scipy.sparse.csr_matrix((np.ones(pat[1].size), pat[1], pat[0]), shape=A_cpp.shape)
        outside = corr.tocsr() - corr.tocsr().multiply(P)
        assert outside.nnz == 0 or abs(outside).max() == 0.0
        # slave rows / columns end up with the diagonal only: step 2 removed what step 1 had put there
        sl = mpc.slaves[: mpc.num_local_slaves]
        if sl.size:
            assert abs(A_nb[sl]).sum() == pytest.approx(abs(case.diagval) * sl.size, rel=1e-12)
            assert abs(A_nb[:, sl]).sum() == pytest.approx(abs(case.diagval) * sl.size, rel=1e-12)
    if case.L is not None:
        b_cpp = oracle.assemble_vector(case.L, mpc)
        b_nb = no.assemble_vector(case.L, mpc)
        assert abs(b_nb - b_cpp).max() <= 1e-13 * max(1.0, abs(b_cpp).max())
