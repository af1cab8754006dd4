"""Taylor-Hood Stokes blocks with a slip constraint on the velocity space
(BASELINE config 3 at toy size; python/tests/test_stokes_channelflow.py:77-81 forms,
python/tests/test_rectangular_assembly.py nest assembly with (mpc_i, mpc_j)):

    a00 = inner(grad u, grad v) dx   (P2^d x P2^d, component-diagonal 30x30 / 12x12)
    a01 = -p div v dx                (P2^d x P1, rectangular)
    a10 = -div u q dx                (P1 x P2^d)

CPU: the oracle's rectangular blocks obey  A_ij_mpc[free_i, free_j] == K_i^T A_ij K_j.
GPU (-m gpu): HIP kernels == oracle for every block, both scatter algorithms,
and for the lifted right-hand sides.
"""

import numpy as np
import pytest

from dolfinx_mpc_amd import fem
from dolfinx_mpc_amd.mesh import create_unit_cube, create_unit_square
from problems import empty_raw, stokes_slip_problem


def _stokes(dim, n):
    return stokes_slip_problem(dim, n)


def _oracle_blocks(po, dim, n):
    V, Q, bcs, raw_v, forms, L0 = _stokes(dim, n)
    mv = po.OracleMPC.from_raw(V, *raw_v)
    mq = po.OracleMPC.from_raw(Q, *empty_raw())
    mpcs = [mv, mq]
    out = {}
    for (i, j), f in forms.items():
        out[(i, j)] = po.assemble_matrix(f, mpcs[i], mpcs[j], bcs=bcs)
    b0 = po.assemble_vector(L0, mv)
    po.apply_lifting(b0, [forms[(0, 0)]], [bcs], mv)
    b1 = np.zeros(Q.num_dofs)
    po.apply_lifting(b1, [forms[(1, 0)]], [bcs], mq)  # b1 -= A10 g
    out["b0"], out["b1"] = b0, b1
    return (V, Q, bcs, raw_v, forms, L0), mpcs, out


@pytest.mark.parametrize("dim,n", [(2, 3), (3, 2)])
def test_oracle_rectangular_identities(oracle, dim, n):
    po = oracle
    (V, Q, bcs, raw_v, forms, L0), (mv, mq), out = _oracle_blocks(po, dim, n)
    ev, eq = po.OracleMPC.empty(V), po.OracleMPC.empty(Q)
    K = po.gather_transformation_matrix(mv)
    free = np.flatnonzero(mv.is_slave == 0)
    # a00
    A00 = po.assemble_matrix(forms[(0, 0)], ev, ev, bcs=bcs)
    po.compare_mpc_lhs(A00, out[(0, 0)], mv, atol=5e-12 * max(1, abs(A00).max()))
    # a01: rows constrained, columns not -> K^T A01
    A01 = po.assemble_matrix(forms[(0, 1)], ev, eq, bcs=bcs)
    assert abs(K.T @ A01 - out[(0, 1)].tocsr()[free, :]).max() < 5e-12
    assert abs(out[(0, 1)].tocsr()[mv.slaves]).sum() == 0  # slave rows empty, no diagonal in an off-diagonal block
    # a10: columns constrained -> A10 K
    A10 = po.assemble_matrix(forms[(1, 0)], eq, ev, bcs=bcs)
    assert abs(A10 @ K - out[(1, 0)].tocsr()[:, free]).max() < 5e-12
    assert abs(out[(1, 0)].tocsr()[:, mv.slaves]).sum() == 0
    # a10 == a01^T without constraints and bcs
    assert abs(po.assemble_matrix(forms[(1, 0)], eq, ev) - po.assemble_matrix(forms[(0, 1)], ev, eq).T).max() < 1e-13
    # div of a linear field: sum_q A10[q, :] u = -int div(u) = -trace * |domain|
    x = V.tabulate_dof_coordinates()
    u = np.zeros(V.num_dofs)
    grad = np.array([[0.3, -1.0, 0.2], [0.5, 0.7, 0.1], [-0.4, 0.6, -1.1]])[:dim, :dim]
    for k in range(dim):
        u[k::dim] = x[:, :dim] @ grad[k]
    A10_free = po.assemble_matrix(forms[(1, 0)], eq, ev)
    assert (A10_free @ u).sum() == pytest.approx(-np.trace(grad), rel=1e-12)
    # right-hand sides
    b0 = po.assemble_vector(L0, ev)
    po.apply_lifting(b0, [forms[(0, 0)]], [bcs], ev)
    po.compare_mpc_rhs(b0, out["b0"], mv)


def _imported(problem, layout):
    """the same problem with every cell integral as IMPORTED text (tests/ufcx_twin.py: codegen.generate / generate_div; layout
    "ffcx": inside whole FFCx-layout files, the kernels named by the form aliases)"""
    from ufcx_twin import twin_form

    V, Q, bcs, raw_v, forms, L0 = problem
    return V, Q, bcs, raw_v, {k: twin_form(f, layout) for k, f in forms.items()}, twin_form(L0, layout)


@pytest.mark.parametrize("layout", ["function", "ffcx"])
@pytest.mark.parametrize("dim,n", [(2, 3), (3, 2)])
def test_oracle_imported_taylor_hood_blocks_reproduce_builtin_blocks(oracle, dim, n, layout):
    """the Taylor-Hood blocks as FFCx-shaped text (round 6: generate_div for p div(v) / div(u) q) through the oracle's function
    pointer / ufcx objects == the built-in operators"""
    po = oracle
    problem = _stokes(dim, n)
    V, Q, bcs, raw_v, forms, L0 = problem
    _, _, _, _, tforms, tL0 = _imported(problem, layout)
    assert all(f.integrals[0].kernel.form == fem.FORM_UFCX for f in tforms.values()) and tL0.integrals[0].kernel.form == fem.FORM_UFCX
    mv = po.OracleMPC.from_raw(V, *raw_v)
    mq = po.OracleMPC.from_raw(Q, *empty_raw())
    mpcs = [mv, mq]
    for (i, j), f in forms.items():
        want = po.assemble_matrix(f, mpcs[i], mpcs[j], bcs=bcs)
        got = po.assemble_matrix(tforms[(i, j)], mpcs[i], mpcs[j], bcs=bcs)
        assert abs(got - want).max() <= 1e-12 * max(1.0, abs(want).max()), (i, j)
    want, got = po.assemble_vector(L0, mv), po.assemble_vector(tL0, mv)
    assert abs(got - want).max() <= 1e-12 * max(1.0, abs(want).max())


def _product_blocks(dim, n, alg, layout=None):
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd.la import create_vector

    V, Q, bcs, raw_v, forms, L0 = _stokes(dim, n) if layout is None else _imported(_stokes(dim, n), layout)
    mv = dm.MultiPointConstraint(V)
    mv.add_constraint(V, *raw_v)
    mv.finalize()
    mq = dm.MultiPointConstraint(Q)
    mq.finalize()
    mpcs = [mv, mq]
    a = [[forms.get((i, j)) for j in range(2)] for i in range(2)]
    A = dm.create_matrix_nest(a, mpcs)
    # rectangular blocks with two different constraints: device pattern == host pattern
    for i in range(2):
        for j in range(2):
            if a[i][j] is not None:
                rp, cols = dm.create_sparsity_pattern(a[i][j], (mpcs[i], mpcs[j]), where="device")
                assert np.array_equal(rp, A[i][j].rowptr) and np.array_equal(cols, A[i][j].cols), (i, j)
    for i in range(2):
        for j in range(2):
            if a[i][j] is not None:
                dm.assemble_matrix(a[i][j], (mpcs[i], mpcs[j]), bcs=bcs, A=A[i][j], algorithm=alg)
    out = {(i, j): A[i][j].to_scipy() for i in range(2) for j in range(2) if A[i][j] is not None}
    b0 = dm.assemble_vector(L0, mv)
    dm.apply_lifting(b0, [forms[(0, 0)]], [bcs], mv)
    b1 = create_vector(Q)
    dm.apply_lifting(b1, [forms[(1, 0)]], [bcs], mq)
    out["b0"], out["b1"] = b0.numpy(), b1.numpy()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("alg", ["atomic", "rowblock"])
@pytest.mark.parametrize("dim,n", [(2, 3), (3, 2)])
def test_gpu_stokes_blocks_match_oracle(oracle, dim, n, alg):
    _, _, ref = _oracle_blocks(oracle, dim, n)
    out = _product_blocks(dim, n, alg)
    for key in [(0, 0), (0, 1), (1, 0)]:
        assert np.array_equal(out[key].indptr, ref[key].indptr) and np.array_equal(out[key].indices, ref[key].indices)
        scale = max(1.0, abs(ref[key]).max())
        assert abs(out[key].data - ref[key].data).max() <= 1e-12 * scale, key
    for key in ("b0", "b1"):
        assert abs(out[key] - ref[key]).max() <= 1e-12 * max(1.0, abs(ref[key]).max()), key


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["function", "ffcx"])
@pytest.mark.parametrize("alg", ["atomic", "rowblock"])
@pytest.mark.parametrize("dim,n", [(2, 3), (3, 2)])
def test_gpu_imported_taylor_hood_blocks_match_oracle(oracle, dim, n, alg, layout):
    """config 3's blocks as imported text (N1: the path north_star names, beyond scalar P1): P2^d stiffness, p div(v), div(u) q
    and the P2^d source through the imported-kernel row blocks / per-entity kernels, against the built-in oracle"""
    _, _, ref = _oracle_blocks(oracle, dim, n)
    out = _product_blocks(dim, n, alg, layout)
    for key in [(0, 0), (0, 1), (1, 0)]:
        assert np.array_equal(out[key].indptr, ref[key].indptr) and np.array_equal(out[key].indices, ref[key].indices)
        scale = max(1.0, abs(ref[key]).max())
        assert abs(out[key].data - ref[key].data).max() <= 1e-12 * scale, key
    for key in ("b0", "b1"):
        assert abs(out[key] - ref[key]).max() <= 1e-12 * max(1.0, abs(ref[key]).max()), key


@pytest.mark.gpu
@pytest.mark.parametrize("dim,n", [(2, 4), (3, 3)])
def test_gpu_block_scalar_storage_of_the_velocity_block(oracle, dim, n, monkeypatch):
    """a00 = inner(grad u, grad v) on P2^d is S (x) I except for masked entries and constraint couplings: the node-block
    kernel leaves ONE value per d x d block + an overlay (include/mpcx.h mpcx_matrix_args_t::block_vals).  The
    expanded values, the block-scalar SpMV and the scalar-CSR path (MPCX_BLOCK_SCALAR=0) must all be the same matrix
    as the oracle's, with a diagonal value other than 1 and across repeated assemblies into one matrix."""
    import torch

    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd.la import Vector
    from dolfinx_mpc_amd.problem import spmv
    from problems import stokes_slip_problem

    V, Q, bcs, raw_v, forms, L0 = stokes_slip_problem(dim, n)
    mv = dm.MultiPointConstraint(V)
    mv.add_constraint(V, *raw_v)
    mv.finalize()
    om = oracle.OracleMPC.from_raw(V, *raw_v)
    a00 = forms[(0, 0)]
    A = None
    for diagval in (1.0, 2.5):
        ref = oracle.assemble_matrix(a00, om, bcs=bcs, diagval=diagval)
        A = dm.assemble_matrix(a00, mv, bcs=bcs, diagval=diagval, A=A)
        assert A.is_block_scalar, "block-scalar storage expected for a component-diagonal form on a blocked space"
        x = Vector(V.num_dofs)
        x.array.copy_(torch.from_numpy(np.random.default_rng(3).standard_normal(V.num_dofs)))
        y = spmv(A, x).numpy()  # straight from the block-scalar layout (A still unexpanded)
        assert A.is_block_scalar
        S = A.to_scipy()  # expands
        assert not A.is_block_scalar
        assert np.array_equal(S.indptr, ref.indptr) and np.array_equal(S.indices, ref.indices)
        assert abs(S.data - ref.data).max() <= 1e-12 * max(1.0, abs(ref).max())
        yref = ref @ x.numpy()
        assert abs(y - yref).max() <= 1e-12 * max(1.0, abs(yref).max())
    monkeypatch.setenv("MPCX_BLOCK_SCALAR", "0")
    B = dm.assemble_matrix(a00, mv, bcs=bcs, diagval=2.5)
    assert not B.is_block_scalar and B._compact is None
    assert abs(B.to_scipy().data - S.data).max() <= 1e-13 * abs(S.data).max()
    # a matrix that was block-scalar can be assembled into through the scalar path and back
    dm.assemble_matrix(a00, mv, bcs=bcs, diagval=2.5, A=A)
    assert abs(A.to_scipy().data - S.data).max() <= 1e-13 * abs(S.data).max()
    monkeypatch.delenv("MPCX_BLOCK_SCALAR")
    dm.assemble_matrix(a00, mv, bcs=bcs, diagval=2.5, A=A)
    assert A.is_block_scalar and abs(A.to_scipy().data - S.data).max() <= 1e-13 * abs(S.data).max()
