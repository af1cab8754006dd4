"""General Lagrange elements of the host stand-in layer (degree 4 on all four cell types since round 5: the degree
python/tests/test_multispace_mpc.py:12-16 sweeps): degree 1-3 on triangles and quadrilaterals (what the
reference's own assembly tests sweep -- python/tests/test_matrix_assembly.py:23-26, test_vector_assembly.py:22-24:
``degree in range(1, 4)``, ``celltype in [triangle, quadrilateral]``), P3 on tetrahedra and Q2 / Q3 on hexahedra
(python/tests/test_stokes_channelflow.py:21-23: Taylor-Hood of order 2 and 3 on both cell types).  DOLFINx / Basix are absent here, so this module defines

  * the reference nodes and their association with sub-entities (vertices, edges, faces, interior), in the order of the
    element's local dofs: vertices, then the interior nodes of every local edge (``mesh.local_edges`` order, counted from
    the edge's first local vertex), then (hexahedra) of every local face, then the cell interior -- for degree <= 2 on
    simplices this is the order the built-in operators use;
  * the nodal basis through the Vandermonde matrix of the monomials that span P_p (simplices) / Q_p (tensor cells);
  * the global dof numbering: a node on a shared edge is ONE dof, numbered along the edge from its lower to its higher
    global vertex, so the two cells of an edge agree whatever their local orientations (Lagrange elements need only this
    permutation, no sign change -- what DOLFINx's dof transformations do for these elements); the (p - 1)^2 nodes of a
    shared quadrilateral face are numbered in the frame its GLOBAL vertex numbers define (origin = the lowest vertex, first
    direction = towards its lower face neighbour), so the two hexahedra of a face agree under any of the eight relative
    orientations; a triangular face of a P3 tetrahedron carries one node.

Imported (generated) kernels tabulate this basis (codegen.generate_general); the oracle compiles the same text."""

from __future__ import annotations

from functools import lru_cache

import numpy as np

from .mesh import local_edges

_VERTS = {
    "triangle": np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]]),
    "quadrilateral": np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0], [1.0, 1.0]]),
    "tetrahedron": np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]),
    "hexahedron": np.array([[float(v & 1), float(v >> 1 & 1), float(v >> 2 & 1)] for v in range(8)]),
}
# local faces of a hexahedron (mesh.HEX_FACETS order); the face's interior nodes are laid out on its first three
# vertices (origin, first direction, second direction)
_HEX_FACES = np.array([[0, 1, 2, 3], [0, 1, 4, 5], [0, 2, 4, 6], [1, 3, 5, 7], [2, 3, 6, 7], [4, 5, 6, 7]])


def tdim(cell: str) -> int:
    return _VERTS[cell].shape[1]


def is_simplex(cell: str) -> bool:
    return cell in ("triangle", "tetrahedron")


@lru_cache(maxsize=None)
def reference_nodes(cell: str, degree: int):
    """(points (nd, tdim), entity (nd, 2): (dim, local entity index) of every node, position (nd,) of the node among the
    interior nodes of its entity)"""
    p = int(degree)
    V = _VERTS[cell]
    pts, ent, pos = [v for v in V], [(0, i) for i in range(V.shape[0])], [0] * V.shape[0]
    for e, (a, b) in enumerate(local_edges(cell)):
        for k in range(1, p):
            pts.append(V[a] + (V[b] - V[a]) * (k / p))
            ent.append((1, e))
            pos.append(k - 1)
    d = V.shape[1]
    if cell == "tetrahedron" and p >= 3:
        if p > 4:
            raise NotImplementedError("tetrahedra: degree 1-4")
        from .mesh import TET_FACETS

        for f, fv in enumerate(TET_FACETS):
            if p == 3:
                pts.append(V[list(fv)].mean(axis=0))
                ent.append((2, f))
                pos.append(0)
            else:
                # P4: three nodes inside a triangular face, node k the one nearest to the face's k-th vertex (barycentric
                # weights 2/4, 1/4, 1/4); two tets sharing the face agree on them through the GLOBAL ids of the face's
                # vertices (build_dofmap: position = rank of the node's vertex among the three)
                P = V[list(fv)]
                for k in range(3):
                    pts.append((P.sum(axis=0) + P[k]) / 4.0)
                    ent.append((2, f))
                    pos.append(k)
    if cell == "hexahedron":
        for f, fv in enumerate(_HEX_FACES):
            n = 0
            for j in range(1, p):
                for i in range(1, p):
                    pts.append(V[fv[0]] + (V[fv[1]] - V[fv[0]]) * (i / p) + (V[fv[2]] - V[fv[0]]) * (j / p))
                    ent.append((2, f))
                    pos.append(n)
                    n += 1
    # cell interior
    n = 0
    if cell == "triangle":
        for j in range(1, p):
            for i in range(1, p - j):
                pts.append(np.array([i / p, j / p]))
                ent.append((2, 0))
                pos.append(n)
                n += 1
    elif cell == "tetrahedron":
        for k in range(1, p):
            for j in range(1, p - k):
                for i in range(1, p - k - j):
                    pts.append(np.array([i / p, j / p, k / p]))
                    ent.append((3, 0))
                    pos.append(n)
                    n += 1
    elif cell == "quadrilateral":
        for j in range(1, p):
            for i in range(1, p):
                pts.append(np.array([i / p, j / p]))
                ent.append((2, 0))
                pos.append(n)
                n += 1
    else:
        for k in range(1, p):
            for j in range(1, p):
                for i in range(1, p):
                    pts.append(np.array([i / p, j / p, k / p]))
                    ent.append((3, 0))
                    pos.append(n)
                    n += 1
    return np.array(pts).reshape(-1, d), np.array(ent, dtype=np.int64).reshape(-1, 2), np.array(pos, dtype=np.int64)


def num_dofs(cell: str, degree: int) -> int:
    return reference_nodes(cell, degree)[0].shape[0]


@lru_cache(maxsize=None)
def _exponents(cell: str, degree: int):
    d, p = tdim(cell), int(degree)
    rng = range(p + 1)
    if d == 2:
        ex = [(a, b) for b in rng for a in rng]
    else:
        ex = [(a, b, c) for c in rng for b in rng for a in rng]
    if is_simplex(cell):
        ex = [e for e in ex if sum(e) <= p]
    return np.array(ex, dtype=np.int64)


def _monomials(ex: np.ndarray, pts: np.ndarray, deriv: int = -1) -> np.ndarray:
    """values (npts, nmono) of the monomials (deriv = -1) or of their derivative in direction ``deriv``"""
    out = np.ones((pts.shape[0], ex.shape[0]))
    for d in range(ex.shape[1]):
        e = ex[:, d]
        if d == deriv:
            with np.errstate(divide="ignore", invalid="ignore"):
                out *= np.where(e > 0, e * pts[:, [d]] ** np.maximum(e - 1, 0), 0.0)
        else:
            out *= pts[:, [d]] ** e
    return out


def _centre(cell: str) -> np.ndarray:
    """the monomials are taken about the centroid of the reference cell: the Vandermonde matrix of Q3 on a hexahedron has
    condition 3e3 there instead of 2e6 about the origin (nodal basis exact to 1e-14 instead of 2e-12)"""
    return _VERTS[cell].mean(axis=0)


@lru_cache(maxsize=None)
def _coefficients(cell: str, degree: int) -> np.ndarray:
    nodes = reference_nodes(cell, degree)[0]
    Vm = _monomials(_exponents(cell, degree), nodes - _centre(cell))
    return np.linalg.inv(Vm)  # column j: (centred) monomial coefficients of basis function j


def tabulate(cell: str, degree: int, pts) -> tuple[np.ndarray, np.ndarray]:
    """phi (npts, nd) and reference derivatives dphi (tdim, npts, nd) of the nodal basis at reference points"""
    pts = np.asarray(pts, dtype=np.float64).reshape(-1, tdim(cell))
    ex, Cm = _exponents(cell, degree), _coefficients(cell, degree)
    pts = pts - _centre(cell)
    phi = _monomials(ex, pts) @ Cm
    dphi = np.stack([_monomials(ex, pts, d) @ Cm for d in range(tdim(cell))], axis=0)
    return phi, dphi


def build_dofmap(mesh, degree: int):
    """(cell_dofs (nc, nd) int32, number of dofs, dof coordinates (ndofs, 3)): vertex dofs = the mesh nodes, then the
    edge dofs (degree - 1 per edge, numbered from the edge's lower global vertex), then face dofs (hexahedra), then
    the interior dofs of every cell."""
    cell, p = mesh.cell_name, int(degree)
    pts, ent, pos = reference_nodes(cell, p)
    cells = mesh.geometry.dofmap.astype(np.int64)
    nc, nv = cells.shape
    nn = mesh.num_nodes
    nd = pts.shape[0]
    out = np.empty((nc, nd), dtype=np.int64)
    out[:, :nv] = cells
    off = nn
    le = local_edges(cell)
    ne_nodes = p - 1
    if ne_nodes > 0:
        cell_edges, ev = mesh.edges()
        for e, (a, b) in enumerate(le):
            flip = cells[:, a] > cells[:, b]  # the local direction a -> b runs against the global one
            g = cell_edges[:, e].astype(np.int64)
            for k in range(ne_nodes):
                col = nv + e * ne_nodes + k
                kk = np.where(flip, ne_nodes - 1 - k, k)
                out[:, col] = off + g * ne_nodes + kk
        off += ev.shape[0] * ne_nodes
    col = nv + le.shape[0] * ne_nodes
    if cell == "tetrahedron" and p in (3, 4):
        from .mesh import TET_FACETS

        fv = np.sort(cells[:, TET_FACETS], axis=2).reshape(nc * 4, 3)
        _, inv = np.unique(fv, axis=0, return_inverse=True)
        inv = inv.reshape(nc, 4)
        nf = 1 if p == 3 else 3  # nodes inside a triangular face
        for f in range(4):
            if p == 3:
                out[:, col + f] = off + inv[:, f]
            else:
                g = cells[:, TET_FACETS[f]]  # (nc, 3) global vertices of the face in local order
                rank = np.argsort(np.argsort(g, axis=1), axis=1)  # rank of every local face vertex among the three
                for k in range(3):  # local face node k sits next to local face vertex k
                    out[:, col + f * 3 + k] = off + inv[:, f].astype(np.int64) * 3 + rank[:, k]
        off += (int(inv.max()) + 1) * nf
        col += 4 * nf
    if cell == "hexahedron" and p >= 2:
        fv = np.sort(cells[:, _HEX_FACES], axis=2).reshape(nc * 6, 4)
        _, inv = np.unique(fv, axis=0, return_inverse=True)
        inv = inv.reshape(nc, 6)
        nfaces = int(inv.max()) + 1
        m = p - 1  # nodes per direction inside a face
        for f in range(6):
            g = cells[:, _HEX_FACES[f]]  # global vertices of the face in local tensor order: (0,0) (1,0) (0,1) (1,1)
            o = np.argmin(g, axis=1)  # the face's lowest global vertex: origin of the shared frame
            io, jo = o & 1, o >> 1
            rows = np.arange(nc)
            first_is_i = g[rows, o ^ 1] < g[rows, o ^ 2]  # the frame's first direction: towards the lower neighbour
            n = 0
            for j in range(1, p):
                for i in range(1, p):
                    di = np.where(io == 1, p - i, i)  # distance from the origin along the local directions
                    dj = np.where(jo == 1, p - j, j)
                    sfirst = np.where(first_is_i, di, dj)
                    ssecond = np.where(first_is_i, dj, di)
                    out[:, col + f * m * m + n] = off + inv[:, f].astype(np.int64) * (m * m) + (ssecond - 1) * m + (sfirst - 1)
                    n += 1
        off += nfaces * m * m
        col += 6 * m * m
    nint = nd - col
    for k in range(nint):
        out[:, col + k] = off + np.arange(nc) * nint + k
    off += nc * nint
    # coordinates: push the reference nodes forward with the (P1 / Q1) geometry of every cell
    gphi, _ = tabulate(cell, 1, pts)  # (nd, nv)
    xc = mesh.geometry.x[cells]  # (nc, nv, 3)
    xd = np.einsum("dv,cvk->cdk", gphi, xc)
    coords = np.empty((off, 3))
    coords[out.reshape(-1)] = xd.reshape(-1, 3)
    return out.astype(np.int32), int(off), coords


def facet_closure_dofs(cell: str, degree: int):
    """for every local facet: the local dofs in its closure (its vertices, the interior nodes of its edges, its own
    interior nodes) -- ``locate_dofs_topological``"""
    from .mesh import local_facets

    pts, ent, _pos = reference_nodes(cell, degree)
    V = _VERTS[cell]
    out = []
    for fv in local_facets(cell):
        # a node lies in the closure of the facet iff it lies in the affine hull of the facet's vertices AND inside it:
        # on these reference cells every facet is a coordinate-aligned or the x + y (+ z) = 1 plane
        P = V[list(fv)]
        if tdim(cell) == 2:
            t = P[1] - P[0]
            nrm = np.array([-t[1], t[0]])
        else:
            nrm = np.cross(P[1] - P[0], P[2] - P[0])
        dist = (pts - P[0]) @ nrm
        out.append(np.flatnonzero(np.abs(dist) < 1e-12).astype(np.int64))
    return out
