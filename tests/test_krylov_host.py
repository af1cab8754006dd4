"""The Krylov recurrences of dolfinx_mpc_amd/krylov.py on CPU tensors (they are device-agnostic torch programs):
PCG on a Laplacian, preconditioned MINRES on a saddle-point system with an empty (2, 2) block, zero rows (pressure
slaves of a nest system without a11, python/tests/test_stokes_channelflow.py:89-125) and the constant-pressure null
space.  Tolerance: the true residual the solver reports, and 1e-8 against a direct solve where the system is regular."""

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def _lap(n):
    T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(n, n))
    return (sp.kron(sp.eye(n), T) + sp.kron(T, sp.eye(n))).tocsr()


def _mv(A):
    import torch

    At = torch.sparse_csr_tensor(torch.from_numpy(A.indptr.astype(np.int64)), torch.from_numpy(A.indices.astype(np.int64)),
                                 torch.from_numpy(A.data.astype(np.float64)), size=A.shape)
    return lambda v: At @ v


@pytest.mark.parametrize("check_every", [1, 4])
def test_pcg_laplacian(check_every):
    import torch

    from dolfinx_mpc_amd.krylov import pcg

    A = _lap(24)
    rng = np.random.default_rng(0)
    b = rng.standard_normal(A.shape[0])
    dinv = torch.from_numpy(1.0 / A.diagonal())
    x, info = pcg(_mv(A), lambda r: dinv * r, torch.from_numpy(b), rtol=1e-12, max_it=2000, check_every=check_every)
    assert info["converged"], info
    assert np.linalg.norm(A @ x.numpy() - b) <= 1.01e-12 * np.linalg.norm(b)
    assert abs(x.numpy() - spla.spsolve(A.tocsc(), b)).max() < 1e-8


def test_pcg_zero_rhs():
    import torch

    from dolfinx_mpc_amd.krylov import pcg

    A = _lap(5)
    x, info = pcg(_mv(A), lambda r: r, torch.zeros(A.shape[0], dtype=torch.float64))
    assert info["converged"] and info["iterations"] == 0 and float(x.abs().max()) == 0.0


def _saddle(n, empty_rows):
    """[[K, B^T], [B, 0]] with K SPD (n^2 x n^2 Laplacian + I), B a discrete divergence-like full-rank map with the
    constants in the null space of B^T, plus ``empty_rows`` trailing zero rows / columns"""
    K = (_lap(n) + sp.eye(n * n)).tocsr()
    m = n * n // 3
    rng = np.random.default_rng(1)
    B = sp.random(m, n * n, density=0.05, random_state=2, format="csr")
    B = B - sp.csr_matrix(np.outer(np.ones(m), np.asarray(B.sum(axis=0)).ravel() / m))  # columns sum to zero: B^T 1 = 0
    B = sp.csr_matrix(B)
    Z = sp.csr_matrix((empty_rows, n * n))
    A = sp.bmat([[K, B.T, Z.T], [B, None, None], [Z, None, sp.csr_matrix((empty_rows, empty_rows))]], format="csr")
    f = rng.standard_normal(n * n)
    b = np.concatenate([f, np.zeros(m + empty_rows)])  # consistent: the constraint rows have zero data
    return A, K, b, n * n, m


@pytest.mark.parametrize("empty_rows", [0, 3])
def test_minres_saddle_point(empty_rows):
    import torch

    from dolfinx_mpc_amd.krylov import minres

    A, K, b, nu, m = _saddle(12, empty_rows)
    Kinv = spla.factorized(K.tocsc())

    def M_inv(r):  # additive field split: exact K solve on the first block, identity on the second
        out = r.clone()
        out[:nu] = torch.from_numpy(Kinv(r[:nu].numpy()))
        return out

    x, info = minres(_mv(A), M_inv, torch.from_numpy(b), rtol=1e-11, max_it=500, check_every=5)
    assert info["converged"], info
    xn = x.numpy()
    assert np.linalg.norm(A @ xn - b) <= 1.01e-11 * np.linalg.norm(b)
    if empty_rows:
        assert abs(xn[nu + m:]).max() == 0.0  # rows nothing couples to are never touched
    # the velocity part is unique (the multiplier is unique up to the constant): compare with a regularised direct solve
    keep = np.ones(A.shape[0], dtype=bool)
    keep[nu] = False  # pin one multiplier
    keep[nu + m:] = False
    idx = np.flatnonzero(keep)
    ref = np.zeros(A.shape[0])
    ref[idx] = spla.spsolve(A[idx][:, idx].tocsc(), b[idx])
    assert abs(xn[:nu] - ref[:nu]).max() < 1e-8 * max(1.0, abs(ref[:nu]).max())


def test_minres_matches_cg_on_spd():
    import torch

    from dolfinx_mpc_amd.krylov import minres

    A = _lap(16)
    b = np.random.default_rng(5).standard_normal(A.shape[0])
    dinv = torch.from_numpy(1.0 / A.diagonal())
    x, info = minres(_mv(A), lambda r: dinv * r, torch.from_numpy(b), rtol=1e-12, max_it=1000)
    assert info["converged"], info
    assert abs(x.numpy() - spla.spsolve(A.tocsc(), b)).max() < 1e-8


def test_bicgstab_nonsymmetric():
    import torch

    from dolfinx_mpc_amd.krylov import bicgstab

    n = 20
    A = (_lap(n) + 0.4 * sp.kron(sp.eye(n), sp.diags([-1.0, 1.0], [-1, 1], shape=(n, n)))).tocsr()  # convection term
    b = np.random.default_rng(7).standard_normal(A.shape[0])
    dinv = torch.from_numpy(1.0 / A.diagonal())
    x, info = bicgstab(_mv(A), lambda r: dinv * r, torch.from_numpy(b), rtol=1e-11, max_it=2000)
    assert info["converged"], info
    assert np.linalg.norm(A @ x.numpy() - b) <= 1.01e-11 * np.linalg.norm(b)
    assert abs(x.numpy() - spla.spsolve(A.tocsc(), b)).max() < 1e-8
