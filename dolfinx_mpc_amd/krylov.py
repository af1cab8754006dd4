"""Krylov solvers on device tensors for the systems the hot path assembles (the solve that
follows the assembly in the reference's drivers).  The operator and the preconditioner are callables on device tensors;
every scalar of the recurrences stays on the device as a 0-d tensor and the host reads a residual norm only every
``check_every`` iterations, so an iteration is a chain of asynchronous launches.

  * ``pcg``    -- preconditioned conjugate gradients (symmetric positive definite systems: Poisson, elasticity;
                  ``ksp_type cg`` of python/benchmarks/bench_periodic.py:112-149, bench_contact_3D.py:290-316)
  * ``bicgstab`` -- right-preconditioned BiCGStab for nonsymmetric systems (the Jacobians of ``NonlinearProblem``)
  * ``minres`` -- preconditioned MINRES for symmetric indefinite systems with a symmetric positive definite
                  preconditioner: the Stokes nest system (``ksp.setType("minres")`` with an additive field split,
                  python/tests/test_stokes_channelflow.py:107-125, python/demos/demo_stokes_nest.py:231-252)
"""

from __future__ import annotations

import numpy as np


def pcg(A_mv, M_inv, b, x0=None, rtol: float = 1e-10, atol: float = 0.0, max_it: int = 500, check_every: int = 4):
    """x with |b - A x| <= max(rtol |b|, atol); returns (x, info)"""
    import torch

    x = torch.zeros_like(b) if x0 is None else x0.clone()
    r = b - A_mv(x) if x0 is not None else b.clone()
    z = M_inv(r)
    p = z.clone()
    rz = torch.dot(r, z)
    bb = float(torch.dot(b, b))
    tol2 = max(rtol * rtol * bb, atol * atol)
    rr_d = torch.dot(r, r)
    rr = float(rr_d)
    k = 0
    while rr > tol2 and k < max_it:
        for _ in range(min(check_every, max_it - k)):
            Ap = A_mv(p)
            pAp = torch.dot(p, Ap)
            alpha = rz / torch.where(pAp != 0, pAp, torch.ones_like(pAp))  # (an exactly zero residual: nothing moves)
            x = torch.addcmul(x, alpha, p)
            r = torch.addcmul(r, -alpha, Ap)
            rr_d = torch.dot(r, r)
            z = M_inv(r)
            rz_new = torch.dot(r, z)
            p = torch.addcmul(z, rz_new / torch.where(rz != 0, rz, torch.ones_like(rz)), p)
            rz = rz_new
            k += 1
        rr = float(rr_d)
        if not np.isfinite(rr):
            raise RuntimeError("pcg: the residual is not finite (operator or preconditioner not positive definite?)")
    return x, {"iterations": k, "residual_norm": float(np.sqrt(rr)), "b_norm": float(np.sqrt(bb)), "converged": bool(rr <= tol2)}


def minres(A_mv, M_inv, b, x0=None, rtol: float = 1e-10, atol: float = 0.0, max_it: int = 2000, check_every: int = 10):
    """Preconditioned MINRES (Paige and Saunders; the Lanczos form with the preconditioner applied to the unnormalised
    Lanczos vector, Elman / Silvester / Wathen alg. 4.1).  ``A_mv`` symmetric, possibly indefinite or singular with a
    consistent right-hand side; ``M_inv`` symmetric positive definite.  The recurrence carries the residual norm in the
    ``M^-1`` norm (eta); the stopping test is on the true residual |b - A x| <= max(rtol |b|, atol), computed when the
    estimate says so.  Returns (x, info)."""
    import torch

    x = torch.zeros_like(b) if x0 is None else x0.clone()
    v = b - A_mv(x) if x0 is not None else b.clone()
    bnorm = float(torch.linalg.vector_norm(b))
    tol = max(rtol * bnorm, atol)
    z = M_inv(v)
    gamma = torch.sqrt(torch.dot(z, v))
    eta0 = float(gamma)
    info = {"iterations": 0, "residual_norm": float(torch.linalg.vector_norm(v)), "b_norm": bnorm, "converged": False}
    if info["residual_norm"] <= tol:
        info["converged"] = True
        return x, info
    if not np.isfinite(eta0) or eta0 == 0.0:
        raise RuntimeError("minres: the preconditioner is not positive definite on the initial residual")
    one = torch.ones((), dtype=b.dtype, device=b.device)
    zero = torch.zeros((), dtype=b.dtype, device=b.device)
    eta = gamma.clone()
    gamma_old = one
    s_old, s_cur, c_old, c_cur = zero, zero, one, one
    v_old = torch.zeros_like(b)
    w_old, w_cur = torch.zeros_like(b), torch.zeros_like(b)
    k = 0
    est_scale = info["residual_norm"] / eta0  # |r|_2 per unit of |r|_{M^-1} at the start: turns eta into a 2-norm guess
    true_checks = False
    while k < max_it:
        for _ in range(min(check_every, max_it - k)):
            zn = z / gamma
            Az = A_mv(zn)
            delta = torch.dot(Az, zn)
            v_new = Az - (delta / gamma) * v - (gamma / gamma_old) * v_old
            z_new = M_inv(v_new)
            # (a breakdown gamma_new = 0 means the Krylov space is exhausted: the update below is then the last one)
            gamma_new = torch.sqrt(torch.clamp(torch.dot(z_new, v_new), min=0.0))
            a0 = c_cur * delta - c_old * s_cur * gamma
            a1 = torch.sqrt(a0 * a0 + gamma_new * gamma_new)
            a2 = s_cur * delta + c_old * c_cur * gamma
            a3 = s_old * gamma
            a1s = torch.where(a1 > 0, a1, one)
            c_new, s_new = a0 / a1s, gamma_new / a1s
            w_new = (zn - a3 * w_old - a2 * w_cur) / a1s
            x = torch.addcmul(x, c_new * eta, w_new)
            eta = -s_new * eta
            v_old, v, z = v, v_new, z_new
            gamma_old, gamma = gamma, torch.where(gamma_new > 0, gamma_new, one)
            s_old, s_cur, c_old, c_cur = s_cur, s_new, c_cur, c_new
            w_old, w_cur = w_cur, w_new
            k += 1
        est = abs(float(eta)) * est_scale
        if not np.isfinite(est):
            raise RuntimeError("minres: the recurrence is not finite (preconditioner not positive definite?)")
        # (the M^-1-norm estimate may be off the 2-norm by the preconditioner's conditioning: a true residual every eighth
        # check regardless, every check once the estimate is within 1e3 of the target)
        if true_checks or est <= 1e3 * tol or (k // check_every) % 8 == 0:
            true_checks = true_checks or est <= 1e3 * tol
            rn = float(torch.linalg.vector_norm(b - A_mv(x)))
            info["residual_norm"] = rn
            if rn <= tol:
                info["converged"] = True
                break
            if abs(float(eta)) <= 1e-15 * eta0:  # the recurrence has nothing left to give
                break
    else:
        info["residual_norm"] = float(torch.linalg.vector_norm(b - A_mv(x)))
        info["converged"] = info["residual_norm"] <= tol
    info["iterations"] = k
    return x, info


def bicgstab(A_mv, M_inv, b, x0=None, rtol: float = 1e-10, atol: float = 0.0, max_it: int = 2000, check_every: int = 5):
    """Right-preconditioned BiCGStab (van der Vorst) for the nonsymmetric Jacobians of a Newton iteration
    (python/src/dolfinx_mpc/problem.py:26-85: the SNES linear solves); same conventions as ``pcg``."""
    import torch

    x = torch.zeros_like(b) if x0 is None else x0.clone()
    r = b - A_mv(x) if x0 is not None else b.clone()
    rhat = r.clone()
    bnorm = float(torch.linalg.vector_norm(b))
    tol = max(rtol * bnorm, atol)
    one = torch.ones((), dtype=b.dtype, device=b.device)

    def safe(d):
        return torch.where(d != 0, d, one)

    rho_old, alpha, omega = one, one, one
    v = torch.zeros_like(b)
    p = torch.zeros_like(b)
    rn = float(torch.linalg.vector_norm(r))
    k = 0
    while rn > tol and k < max_it:
        for _ in range(min(check_every, max_it - k)):
            rho = torch.dot(rhat, r)
            beta = (rho / safe(rho_old)) * (alpha / safe(omega))
            p = r + beta * (p - omega * v)
            y = M_inv(p)
            v = A_mv(y)
            alpha = rho / safe(torch.dot(rhat, v))
            s = r - alpha * v
            z = M_inv(s)
            t = A_mv(z)
            omega = torch.dot(t, s) / safe(torch.dot(t, t))
            x = x + alpha * y + omega * z
            r = s - omega * t
            rho_old = rho
            k += 1
        r = b - A_mv(x)  # the true residual replaces the recurrence's (which drifts) at every check
        rn = float(torch.linalg.vector_norm(r))
        if not np.isfinite(rn):
            raise RuntimeError("bicgstab: the residual is not finite")
    return x, {"iterations": k, "residual_norm": rn, "b_norm": bnorm, "converged": bool(rn <= tol)}
