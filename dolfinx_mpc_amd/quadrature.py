"""Quadrature rules on the reference simplices.

In the reference the rule is chosen by UFL/FFCx at JIT time and baked into the
generated ``tabulate_tensor`` (third party, absent here), so the rule is part
of the *kernel data*: the same table is handed to the HIP kernels and to the
test oracle.  Our stated choices:

* tetrahedron: degree<=1 centroid, degree 2 the 4-point rule, degree 3..5 the
  14-point degree-5 rule, above that a collapsed Gauss-Legendre rule;
* triangle: degree<=1 centroid, degree 2 the 3-point rule, degree 3..4 the
  6-point degree-4 rule, above that collapsed Gauss-Legendre;
* interval (facets of triangles): Gauss-Legendre on [0, 1].

Weights sum to the reference cell volume (1/6, 1/2, 1).
"""

from __future__ import annotations

import numpy as np


def _gauss01(n: int):
    p, w = np.polynomial.legendre.leggauss(n)
    return 0.5 * (p + 1.0), 0.5 * w


def _collapsed_triangle(degree: int):
    n = (degree + 2 + 1) // 2
    p, w = _gauss01(n)
    U, V = np.meshgrid(p, p, indexing="ij")
    WU, WV = np.meshgrid(w, w, indexing="ij")
    pts = np.stack([U.ravel(), (V * (1 - U)).ravel()], axis=1)
    wts = (WU * WV * (1 - U)).ravel()
    return pts, wts


def _collapsed_tet(degree: int):
    n = (degree + 3 + 1) // 2
    p, w = _gauss01(n)
    U, V, W = np.meshgrid(p, p, p, indexing="ij")
    WU, WV, WW = np.meshgrid(w, w, w, indexing="ij")
    pts = np.stack([U.ravel(), (V * (1 - U)).ravel(), (W * (1 - U) * (1 - V)).ravel()], axis=1)
    wts = (WU * WV * WW * (1 - U) ** 2 * (1 - V)).ravel()
    return pts, wts


def _tet14():
    # 14-point, degree-5 rule (weights for the reference tet of volume 1/6)
    a1, w1 = 0.31088591926330060980, 0.11268792571801585080
    a2, w2 = 0.092735250310891226402, 0.073493043116361949544
    b, w3 = 0.045503704125649649492, 0.042546020777081466438
    pts, wts = [], []
    for a, w in ((a1, w1), (a2, w2)):
        c = 1.0 - 3.0 * a
        for p in ((a, a, a), (c, a, a), (a, c, a), (a, a, c)):
            pts.append(p)
            wts.append(w / 6.0)
    c = 0.5 - b
    for p in ((b, b, c), (b, c, b), (c, b, b), (b, c, c), (c, b, c), (c, c, b)):
        pts.append(p)
        wts.append(w3 / 6.0)
    return np.array(pts), np.array(wts)


def _tet24():
    """Keast's 24-point rule of degree 6 (positive weights): three orbits (a, a, a, 1-3a) and
    one orbit (a, a, b, 1-2a-b); exact to 4e-16 on every monomial of degree <= 6
    (tests/test_oracle_kernels.py)."""
    import itertools

    pts, wts = [], []
    for a, w in ((0.2146028712591521, 0.0399227502581679), (0.0406739585346113, 0.0100772110553207),
                 (0.3223378901422757, 0.0553571815436544)):
        for perm in sorted(set(itertools.permutations((a, a, a, 1.0 - 3.0 * a)))):
            pts.append(perm[1:])
            wts.append(w)
    a, b, w = 0.0636610018750175, 0.2696723314583159, 0.0482142857142857
    for perm in sorted(set(itertools.permutations((a, a, b, 1.0 - 2.0 * a - b)))):
        pts.append(perm[1:])
        wts.append(w)
    return np.array(pts), np.array(wts) / 6.0


def make_quadrature(cell_name: str, degree: int):
    """Return (points (nq, tdim), weights (nq,)) exact for polynomials of ``degree``."""
    degree = max(int(degree), 0)
    if cell_name == "tetrahedron":
        if degree <= 1:
            return np.array([[0.25, 0.25, 0.25]]), np.array([1.0 / 6.0])
        if degree == 2:
            a, b = 0.1381966011250105, 0.5854101966249685
            return np.array([[a, a, a], [b, a, a], [a, b, a], [a, a, b]]), np.full(4, 1.0 / 24.0)
        if degree <= 5:
            return _tet14()
        if degree == 6:
            return _tet24()
        return _collapsed_tet(degree)
    if cell_name == "triangle":
        if degree <= 1:
            return np.array([[1.0 / 3.0, 1.0 / 3.0]]), np.array([0.5])
        if degree == 2:
            return np.array([[1 / 6, 1 / 6], [2 / 3, 1 / 6], [1 / 6, 2 / 3]]), np.full(3, 1.0 / 6.0)
        if degree <= 4:
            a, wa = 0.445948490915965, 0.223381589678011
            b, wb = 0.091576213509771, 0.109951743655322
            pts = [[a, a], [1 - 2 * a, a], [a, 1 - 2 * a], [b, b], [1 - 2 * b, b], [b, 1 - 2 * b]]
            return np.array(pts), np.array([wa, wa, wa, wb, wb, wb]) * 0.5
        return _collapsed_triangle(degree)
    if cell_name == "interval":
        p, w = _gauss01(max((degree + 2) // 2, 1))
        return p.reshape(-1, 1), w
    raise ValueError(f"unsupported cell {cell_name}")


def facet_cell_name(cell_name: str) -> str:
    return {"tetrahedron": "triangle", "triangle": "interval"}[cell_name]


def lagrange_basis(cell: str, degree: int, pts: np.ndarray) -> np.ndarray:
    """values (npts, nd) of the scalar Lagrange P1/P2 basis at reference points, in the dof order of the element
    kernels: vertices, then (P2) edges -- tets (2,3)(1,3)(1,2)(0,3)(0,2)(0,1), triangles (1,2)(0,2)(0,1)"""
    pts = np.asarray(pts, dtype=np.float64).reshape(-1, 3 if cell == "tetrahedron" else 2)
    lam = np.concatenate([1.0 - pts.sum(axis=1, keepdims=True), pts], axis=1)
    if degree == 1:
        return lam
    edges = [(2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1)] if cell == "tetrahedron" else [(1, 2), (0, 2), (0, 1)]
    return np.concatenate([lam * (2.0 * lam - 1.0)] + [4.0 * lam[:, [a]] * lam[:, [b]] for a, b in edges], axis=1)
