"""Structured simplicial meshes as flat arrays.

The reference receives these arrays from DOLFINx (``mesh.geometry.x``,
``mesh.geometry.dofmaps[0]``; read at cpp/assemble_matrix.cpp:465-470 and
python/src/dolfinx_mpc/numba/assemble_matrix.py:85-86).  DOLFINx is not
available here, so the generators below define our own connectivity and
numbering; geometry is always stored with 3 components
per node (cpp/assemble_matrix.cpp:473, 499).
"""

from __future__ import annotations

import numpy as np

# local facets: facet i is opposite vertex i (Basix/UFC convention)
TET_FACETS = np.array([[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]], dtype=np.int32)
TRI_FACETS = np.array([[1, 2], [0, 2], [0, 1]], dtype=np.int32)
# local edges
TET_EDGES = np.array([[2, 3], [1, 3], [1, 2], [0, 3], [0, 2], [0, 1]], dtype=np.int32)
TRI_EDGES = np.array([[1, 2], [0, 2], [0, 1]], dtype=np.int32)

# Kuhn split of the unit cube into 6 tets sharing the diagonal v0-v7;
# cube corner b has bit0 = x, bit1 = y, bit2 = z.
_KUHN = np.array(
    [[0, 1, 3, 7], [0, 1, 7, 5], [0, 5, 7, 4], [0, 3, 2, 7], [0, 6, 4, 7], [0, 2, 6, 7]], dtype=np.int64
)


# hexahedron, DOLFINx (tensor-product) vertex order: vertex v has (x, y, z) bits (v & 1, v >> 1 & 1, v >> 2 & 1)
HEX_FACETS = np.array([[0, 1, 2, 3], [0, 1, 4, 5], [0, 2, 4, 6], [1, 3, 5, 7], [2, 3, 6, 7], [4, 5, 6, 7]], dtype=np.int64)
HEX_EDGES = np.array(
    [[0, 1], [0, 2], [0, 4], [1, 3], [1, 5], [2, 3], [2, 6], [3, 7], [4, 5], [4, 6], [5, 7], [6, 7]], dtype=np.int64
)


# quadrilateral, DOLFINx (tensor-product) vertex order: vertex v has (x, y) bits (v & 1, v >> 1 & 1); facets = edges
QUAD_EDGES = np.array([[0, 1], [0, 2], [1, 3], [2, 3]], dtype=np.int64)


def local_facets(cell_name: str) -> np.ndarray:
    return {"tetrahedron": TET_FACETS, "triangle": TRI_FACETS, "hexahedron": HEX_FACETS, "quadrilateral": QUAD_EDGES}[cell_name]


def local_edges(cell_name: str) -> np.ndarray:
    return {"tetrahedron": TET_EDGES, "triangle": TRI_EDGES, "hexahedron": HEX_EDGES, "quadrilateral": QUAD_EDGES}[cell_name]


class Geometry:
    """``x`` (num_nodes, 3) float64 and ``dofmap`` (num_cells, nv) int32.

    The reference gathers the coordinates from ``mesh.geometry.x`` on every assembly call
    (cpp/assemble_matrix.cpp:495-501), so a moved mesh is picked up automatically.  Here the kernels read
    a device mirror, and comparing 400 MB of host coordinates per call would cost more than the assembly:
    ``x`` is therefore handed out READ-ONLY (an in-place write raises instead of silently assembling on
    stale coordinates) and a moved mesh is stated explicitly, ``mesh.geometry.x = new_x`` (or
    ``set_x``), which bumps ``version``; the device mirror is refreshed when the version differs."""

    def __init__(self, x: np.ndarray, dofmap: np.ndarray):
        self._x = np.array(x, dtype=np.float64, order="C", copy=True)
        self._x.flags.writeable = False
        self.dofmap = dofmap
        self.version = 0

    @property
    def x(self) -> np.ndarray:
        return self._x

    @x.setter
    def x(self, value):
        self.set_x(value)

    def set_x(self, value):
        value = np.asarray(value, dtype=np.float64)
        if value.shape != self._x.shape:
            raise ValueError(f"geometry.x has shape {self._x.shape}, got {value.shape}")
        new = np.array(value, dtype=np.float64, order="C", copy=True)
        new.flags.writeable = False
        self._x = new
        self.version += 1


class Mesh:
    """Single-process mesh, one geometry dofmap: affine simplices, or trilinear hexahedra (Q1 spaces, element
    kernels generated as UFCx C text -- dolfinx_mpc_amd/codegen.py)."""

    def __init__(self, x: np.ndarray, cells: np.ndarray, cell_name: str):
        assert cell_name in ("triangle", "tetrahedron", "hexahedron", "quadrilateral")
        self.geometry = Geometry(np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(cells, dtype=np.int32))
        self.cell_name = cell_name
        self.tdim = 2 if cell_name in ("triangle", "quadrilateral") else 3
        self.geometry.dim = self.tdim  # (planar meshes live in the z = 0 plane of the 3-column coordinate array)
        self._exterior_facets = None
        self._edges = None
        self._device = {}
        # single process by default; dolfinx_mpc_amd.distributed.create_slab_mesh overrides
        self.num_owned_nodes = self.geometry.x.shape[0]
        self.num_owned_cells = self.geometry.dofmap.shape[0]
        self.node_global = None
        # first node of every tile when the numbering is tiled (hint for row blocks)
        self.node_tile_offsets = None

    @property
    def num_cells(self) -> int:
        return self.geometry.dofmap.shape[0]

    @property
    def num_nodes(self) -> int:
        return self.geometry.x.shape[0]

    # -- topology helpers (small meshes only) -------------------------------
    def exterior_facets(self) -> np.ndarray:
        """(cell, local_facet) pairs of all boundary facets, stride-2 layout of
        the reference's exterior-facet domains (cpp/assemble_matrix.cpp:343-348)."""
        if self._exterior_facets is None:
            cells = self.geometry.dofmap.astype(np.int64)
            lf = local_facets(self.cell_name)
            nc, nf = cells.shape[0], lf.shape[0]
            fv = np.sort(cells[:, lf], axis=2).reshape(nc * nf, -1)  # (nc*nf, vertices per facet)
            nn = self.num_nodes
            if float(nn) ** fv.shape[1] < 2.0**62:
                key = fv[:, 0]
                for k in range(1, fv.shape[1]):
                    key = key * nn + fv[:, k]
                _, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
            else:
                _, inv, cnt = np.unique(fv, axis=0, return_inverse=True, return_counts=True)
                inv = inv.reshape(-1)
            ext = np.flatnonzero(cnt[inv] == 1)
            out = np.empty((ext.size, 2), dtype=np.int32)
            out[:, 0] = ext // nf
            out[:, 1] = ext % nf
            self._exterior_facets = out
        return self._exterior_facets

    def facet_midpoints(self, facets: np.ndarray) -> np.ndarray:
        lf = local_facets(self.cell_name)
        verts = self.geometry.dofmap[facets[:, 0]][np.arange(facets.shape[0])[:, None], lf[facets[:, 1]]]
        return self.geometry.x[verts].mean(axis=1)

    def locate_exterior_facets(self, marker) -> np.ndarray:
        """Boundary facets whose midpoint satisfies ``marker(x)`` (x is (3, n))."""
        f = self.exterior_facets()
        mid = self.facet_midpoints(f)
        return np.ascontiguousarray(f[np.asarray(marker(mid.T), dtype=bool)])

    def edges(self):
        """Global edge numbering: returns (cell_edges (nc, ne) int32, edge_vertices (nE, 2))."""
        if self._edges is None:
            cells = self.geometry.dofmap.astype(np.int64)
            le = local_edges(self.cell_name)
            ev = np.sort(cells[:, le], axis=2).reshape(-1, 2)
            key = ev[:, 0] * self.num_nodes + ev[:, 1]
            uniq, inv = np.unique(key, return_inverse=True)
            edge_vertices = np.stack([uniq // self.num_nodes, uniq % self.num_nodes], axis=1)
            self._edges = (inv.reshape(cells.shape[0], le.shape[0]).astype(np.int32), edge_vertices)
        return self._edges


def _tile_permutation(n1: tuple, tile: tuple, shift: tuple = (0, 0, 0)) -> np.ndarray:
    """old node index -> new node index, numbering nodes tile by tile
    (x fastest inside a tile, tiles x fastest).  ``shift``: grid index at which the tiling starts along every axis (the
    first OWNED plane of a slab: the ghost planes below it form a thin tile layer of their own, so that the owned nodes
    fall into whole tiles)."""
    nx1, ny1, nz1 = n1
    tx, ty, tz = tile
    k, j, i = np.meshgrid(np.arange(nz1), np.arange(ny1), np.arange(nx1), indexing="ij")
    i, j, k = i - shift[0], j - shift[1], k - shift[2]
    o = [1 if sh > 0 else 0 for sh in shift]  # (the layer of the indices below the start)
    ntx, nty = -(-(nx1 - shift[0]) // tx) + o[0], -(-(ny1 - shift[1]) // ty) + o[1]
    tid = ((k // tz + o[2]) * nty + (j // ty + o[1])) * ntx + (i // tx + o[0])
    loc = ((k % tz) * ty + (j % ty)) * tx + (i % tx)
    key = tid.astype(np.int64) * (tx * ty * tz) + loc
    order = np.argsort(key.ravel(), kind="stable")  # new -> old
    perm = np.empty_like(order)
    perm[order] = np.arange(order.size)
    return perm  # old -> new


def _tile_ids(n1: tuple, tile: tuple, shift: tuple = (0, 0, 0)) -> np.ndarray:
    """tile id of every node of the (nx1, ny1, nz1) grid, lexicographic node order (``shift``: see _tile_permutation)"""
    nx1, ny1, nz1 = n1
    tx, ty, tz = tile
    k, j, i = np.meshgrid(np.arange(nz1), np.arange(ny1), np.arange(nx1), indexing="ij")
    i, j, k = i - shift[0], j - shift[1], k - shift[2]
    o = [1 if sh > 0 else 0 for sh in shift]
    ntx, nty = -(-(nx1 - shift[0]) // tx) + o[0], -(-(ny1 - shift[1]) // ty) + o[1]
    return (((k // tz + o[2]) * nty + (j // ty + o[1])) * ntx + (i // tx + o[0])).ravel()


def _tile_starts(tid_new: np.ndarray) -> np.ndarray:
    """first node of every run of equal tile id (numbering hint for row blocks)"""
    return np.concatenate([[0], np.flatnonzero(np.diff(tid_new) != 0) + 1]).astype(np.int32)


def create_box(p0, p1, n, cell_type: str = "tetrahedron", reorder: tuple | None = None) -> Mesh:
    """Box mesh of ``n = (nx, ny, nz)`` cubes, each split into 6 tets (``cell_type="tetrahedron"``) or kept as one
    hexahedron (``"hexahedron"``, the default cell of python/benchmarks/bench_periodic.py:38,199-200).

    Node index ``(k*(ny+1) + j)*(nx+1) + i`` (x fastest); cells follow cube
    order (x fastest), 6 consecutive tets per cube.  ``reorder=(tx,ty,tz)``
    renumbers nodes and cells tile by tile (DOLFINx also reorders for locality;
    numbering is not part of the reference's contract).
    """
    assert cell_type in ("tetrahedron", "hexahedron")
    per_cube, nvc = (6, 4) if cell_type == "tetrahedron" else (1, 8)
    nx, ny, nz = n
    xs = np.linspace(p0[0], p1[0], nx + 1)
    ys = np.linspace(p0[1], p1[1], ny + 1)
    zs = np.linspace(p0[2], p1[2], nz + 1)
    Z, Y, X = np.meshgrid(zs, ys, xs, indexing="ij")
    x = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    base = ((k * (ny + 1) + j) * (nx + 1) + i).ravel().astype(np.int64)
    corner = np.empty((base.size, 8), dtype=np.int64)
    for b in range(8):
        corner[:, b] = base + (b & 1) + ((b >> 1) & 1) * (nx + 1) + ((b >> 2) & 1) * (nx + 1) * (ny + 1)
    cells = corner[:, _KUHN].reshape(-1, 4) if cell_type == "tetrahedron" else corner
    if reorder is not None:
        perm = _tile_permutation((nx + 1, ny + 1, nz + 1), reorder)
        xn = np.empty_like(x)
        xn[perm] = x
        x = xn
        cells = perm[cells]
        # cells tile by tile as well (tile of the cube's lower corner)
        cperm = _tile_permutation((nx, ny, nz), reorder)  # old cube -> new cube
        order = np.argsort(cperm, kind="stable")  # new cube -> old cube
        cells = cells.reshape(-1, per_cube, nvc)[order].reshape(-1, nvc)
        tid_new = np.empty(perm.size, dtype=np.int64)
        tid_new[perm] = _tile_ids((nx + 1, ny + 1, nz + 1), reorder)
    mesh = Mesh(x, cells.astype(np.int32), cell_type)
    if reorder is not None:
        mesh.node_tile_offsets = _tile_starts(tid_new)
    return mesh


def create_unit_cube(nx: int, ny: int, nz: int, cell_type: str = "tetrahedron", reorder=None) -> Mesh:
    """python/benchmarks/bench_periodic.py:43 ``create_unit_cube(comm, N, N, N, ct)``."""
    return create_box((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), (nx, ny, nz), cell_type, reorder)


def create_rectangle(p0, p1, n, cell_type: str = "triangle") -> Mesh:
    """rectangle of nx x ny squares: two triangles each, or one quadrilateral (DOLFINx tensor-product vertex order) --
    python/tests/test_matrix_assembly.py:25-30 sweeps both cell types"""
    assert cell_type in ("triangle", "quadrilateral")
    nx, ny = n
    xs = np.linspace(p0[0], p1[0], nx + 1)
    ys = np.linspace(p0[1], p1[1], ny + 1)
    Y, X = np.meshgrid(ys, xs, indexing="ij")
    x = np.stack([X.ravel(), Y.ravel(), np.zeros(X.size)], axis=1)
    j, i = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    v0 = (j * (nx + 1) + i).ravel().astype(np.int64)
    v1, v2, v3 = v0 + 1, v0 + nx + 1, v0 + nx + 2
    if cell_type == "quadrilateral":
        return Mesh(x, np.stack([v0, v1, v2, v3], axis=1).astype(np.int32), "quadrilateral")
    cells = np.stack([np.stack([v0, v1, v3], axis=1), np.stack([v0, v3, v2], axis=1)], axis=1).reshape(-1, 3)
    return Mesh(x, cells.astype(np.int32), "triangle")


def create_unit_square(nx: int, ny: int, cell_type: str = "triangle") -> Mesh:
    """python/tests/test_matrix_assembly.py:30 ``create_unit_square(comm, 5, 3, ct)``."""
    return create_rectangle((0.0, 0.0), (1.0, 1.0), (nx, ny), cell_type)


# ---------------------------------------------------------------------------------------------
# facet tags and the two-body mesh of the contact benchmark
# ---------------------------------------------------------------------------------------------
class MeshTags:
    """``dolfinx.mesh.MeshTags`` stand-in for FACETS: entities are (cell, local_facet) pairs -- the
    stride-2 layout the reference's exterior-facet integrals use
    (cpp/assemble_matrix.cpp:343-348) -- with one integer value each."""

    def __init__(self, mesh: Mesh, dim: int, entities: np.ndarray, values: np.ndarray):
        assert dim == mesh.tdim - 1, "facet tags only"
        self.mesh = mesh
        self.dim = dim
        self.entities = np.ascontiguousarray(entities, dtype=np.int32).reshape(-1, 2)
        self.values = np.ascontiguousarray(values, dtype=np.int32)
        assert self.entities.shape[0] == self.values.size

    def find(self, value: int) -> np.ndarray:
        """(cell, local_facet) pairs tagged ``value`` (python/benchmarks/bench_contact_3D.py:222 ``mt.find(5)``)"""
        return np.ascontiguousarray(self.entities[self.values == value])


def renumber(mesh: Mesh, node_new_of_old: np.ndarray, cell_old_of_new: np.ndarray, tags: "MeshTags | None" = None):
    """The same mesh with nodes renumbered (``node_new_of_old``) and cells reordered (``cell_old_of_new``); local
    vertex order inside every cell is kept, so (cell, local_facet) pairs only change their cell index.  Returns the
    new mesh (and the re-indexed facet tags if ``tags`` is given)."""
    perm = np.asarray(node_new_of_old, dtype=np.int64)
    order = np.asarray(cell_old_of_new, dtype=np.int64)
    out = None
    if mesh.num_cells >= 1_000_000:
        # big meshes: the gathers on the device when there is one (include/mpcx.h mpcx_renumber_mesh: the host's fancy indexing
        # of 4 x 10^8 node ids takes 12 s at 256^3, the device 0.4 s with the transfers)
        try:
            from . import locality

            if locality._gpu():
                out, _inv = locality._renumber(mesh, perm, order)
        except Exception:  # noqa: BLE001  (no library / no device: the numpy path below)
            out = None
    if out is None:
        x = np.empty_like(mesh.geometry.x)
        x[perm] = mesh.geometry.x
        cells = perm[mesh.geometry.dofmap.astype(np.int64)][order]
        out = Mesh(x, cells.astype(np.int32), mesh.cell_name)
    if tags is None:
        return out
    cell_new_of_old = np.empty(order.size, dtype=np.int64)
    cell_new_of_old[order] = np.arange(order.size)
    ents = tags.entities.astype(np.int64).copy()
    ents[:, 0] = cell_new_of_old[ents[:, 0]]
    return out, MeshTags(out, tags.dim, ents, tags.values.copy())


def reorder_spatial(mesh: Mesh, tags: "MeshTags | None" = None, tile_nodes: int = 512):
    """Renumber an arbitrarily numbered mesh for locality: nodes along a Z-order (Morton) curve through their
    coordinates, cells by their lowest node.  The row-block kernels keep contiguous CSR row ranges in LDS and
    evaluate every cell that touches a range, so they need numberings in which contiguous ranges are compact in
    space -- the generators' tile-wise numbering is, a mesh read from a file may not be (DOLFINx reorders dofs for
    locality too; the numbering is not part of the reference's contract).  ``node_tile_offsets`` of the result
    marks every ``tile_nodes``-th node: along a Morton curve any aligned run is a compact brick."""
    x = mesh.geometry.x
    lo, hi = x.min(axis=0), x.max(axis=0)
    span = np.where(hi > lo, hi - lo, 1.0)
    bits = 21
    q = np.minimum(((x - lo) / span * (1 << bits)).astype(np.int64), (1 << bits) - 1)
    code = np.zeros(x.shape[0], dtype=np.int64)
    for b in range(bits):
        for d in range(3):
            code |= ((q[:, d] >> b) & 1) << (3 * b + d)
    order = np.argsort(code, kind="stable")  # new -> old
    perm = np.empty_like(order)
    perm[order] = np.arange(order.size)
    cell_key = perm[mesh.geometry.dofmap.astype(np.int64)].min(axis=1)
    cell_order = np.argsort(cell_key, kind="stable")
    res = renumber(mesh, perm, cell_order, tags)
    out = res[0] if tags is not None else res
    out.node_tile_offsets = np.arange(0, out.num_nodes, tile_nodes, dtype=np.int32)
    return res


def facet_vertices(mesh: Mesh, facets: np.ndarray) -> np.ndarray:
    """geometry nodes of the given (cell, local_facet) pairs, shape (n, tdim)"""
    lf = TET_FACETS if mesh.tdim == 3 else TRI_FACETS
    facets = np.asarray(facets).reshape(-1, 2)
    return mesh.geometry.dofmap[facets[:, 0]][np.arange(facets.shape[0])[:, None], lf[facets[:, 1]]]


def rotation_matrix(axis, angle: float) -> np.ndarray:
    """rotation about ``axis`` by ``angle`` (Rodrigues; python/src/dolfinx_mpc/utils/mpc_utils.py:35-48)"""
    n = np.asarray(axis, dtype=np.float64)
    n = n / np.sqrt(n @ n)
    K = np.array([[0, -n[2], n[1]], [n[2], 0, -n[0]], [-n[1], n[0], 0]])
    return np.sin(angle) * K + np.cos(angle) * np.eye(3) + (1 - np.cos(angle)) * np.outer(n, n)


def _facets_on_plane(cells: np.ndarray, cell_ids: np.ndarray, on_plane: np.ndarray) -> np.ndarray:
    """(cell, local_facet) of the facets of ``cell_ids`` whose three vertices all satisfy ``on_plane``"""
    hit = on_plane[cells[cell_ids]]  # (n, 4) vertex flags
    out = []
    for f in range(4):
        sel = hit[:, TET_FACETS[f]].all(axis=1)
        if sel.any():
            out.append(np.stack([cell_ids[sel], np.full(int(sel.sum()), f, dtype=np.int64)], axis=1))
    return np.concatenate(out, axis=0).astype(np.int32) if out else np.zeros((0, 2), dtype=np.int32)


def merge_meshes(meshes) -> Mesh:
    """Disjoint union: points and cells concatenated in the given order (what
    python/benchmarks/bench_contact_3D.py:108-110 does with ``np.vstack``)."""
    xs, cs, hints = [], [], []
    off = 0
    for m in meshes:
        xs.append(m.geometry.x)
        cs.append(m.geometry.dofmap.astype(np.int64) + off)
        if m.node_tile_offsets is not None:
            hints.append(m.node_tile_offsets.astype(np.int64) + off)
        else:
            hints.append(np.array([off], dtype=np.int64))
        off += m.num_nodes
    out = Mesh(np.concatenate(xs, axis=0), np.concatenate(cs, axis=0).astype(np.int32), meshes[0].cell_name)
    if any(m.node_tile_offsets is not None for m in meshes):
        out.node_tile_offsets = np.concatenate(hints).astype(np.int32)
    return out


# facet markers of the contact benchmark (python/benchmarks/bench_contact_3D.py:163-168)
CONTACT_TOP, CONTACT_BOTTOM_INTERFACE, CONTACT_TOP_INTERFACE, CONTACT_BOTTOM = 3, 4, 9, 5


def create_stacked_cubes(n_top: int, n_bottom: int | None = None, theta: float = 0.0, reorder=None):
    """The two-body mesh of python/benchmarks/bench_contact_3D.py:62-195 (``mesh_3D_dolfin``): a unit cube
    with ``n_top``^3 cubes on z in [1, 2] stacked on a unit cube with ``n_bottom``^3 (default 2 n_top) on
    z in [0, 1]; points and cells of the top body first, then the bottom body; every cube split into 6
    tets; the whole rotated about (1, 1, 0)/sqrt(2) by -theta.  The two bodies share no node: they touch
    along the plane z = 1 (before rotation).

    Returns (mesh, facet_tags, cell_tags) with the reference's markers: 3 top (z = 2), 4 bottom
    interface (the bottom body's facets on z = 1), 9 top interface (the top body's), 5 bottom
    (z = 0); cell_tags[c] = 2 for cells of the top body, 0 otherwise."""
    n_bottom = 2 * n_top if n_bottom is None else n_bottom
    top = create_box((0.0, 0.0, 1.0), (1.0, 1.0, 2.0), (n_top,) * 3, "tetrahedron", reorder)
    bot = create_box((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), (n_bottom,) * 3, "tetrahedron", reorder)
    return _stack_bodies(top, bot, theta)


def _stack_bodies(top: Mesh, bot: Mesh, theta: float = 0.0):
    """two tetrahedral bodies on z in [1, 2] and [0, 1] merged (top first) with the contact benchmark's facet markers"""
    mesh = merge_meshes([top, bot])
    x = mesh.geometry.x
    z = x[:, 2]
    cells = mesh.geometry.dofmap
    nct = top.num_cells
    is_top_node = np.arange(mesh.num_nodes) < top.num_nodes
    # candidate cells: those with a vertex on the plane (cheap: vertex flags, no global facet sort)
    def tagged(on_plane, body_cells):
        cand = body_cells[on_plane[cells[body_cells]].any(axis=1)]
        return _facets_on_plane(cells, cand, on_plane)

    top_cells = np.arange(nct, dtype=np.int64)
    bot_cells = np.arange(nct, mesh.num_cells, dtype=np.int64)
    f_top = tagged(np.isclose(z, 2.0), top_cells)
    f_tif = tagged(np.isclose(z, 1.0) & is_top_node, top_cells)
    f_bif = tagged(np.isclose(z, 1.0) & ~is_top_node, bot_cells)
    f_bot = tagged(np.isclose(z, 0.0), bot_cells)
    ents = np.concatenate([f_top, f_bif, f_tif, f_bot], axis=0)
    vals = np.concatenate([np.full(f.shape[0], v, dtype=np.int32) for f, v in
                           ((f_top, CONTACT_TOP), (f_bif, CONTACT_BOTTOM_INTERFACE), (f_tif, CONTACT_TOP_INTERFACE),
                            (f_bot, CONTACT_BOTTOM))])
    if theta != 0.0:
        R = rotation_matrix([1 / np.sqrt(2), 1 / np.sqrt(2), 0], -theta)
        mesh.geometry.x = x @ R.T
    cell_tags = np.zeros(mesh.num_cells, dtype=np.int32)
    cell_tags[:nct] = 2
    return mesh, MeshTags(mesh, 2, ents, vals), cell_tags


def _delaunay_points(n, p0, p1, rng, jitter: float) -> np.ndarray:
    """lattice of (n_d + 1) points per direction on the box, every point moved by up to ``jitter`` lattice widths -- but
    only inside the face / edge of the box it lies on (corners stay): the boundary of the box stays the boundary"""
    n = np.asarray(n, dtype=np.int64)
    dim = n.size
    p0, p1 = np.asarray(p0, dtype=np.float64)[:dim], np.asarray(p1, dtype=np.float64)[:dim]
    idx = np.stack(np.meshgrid(*[np.arange(k + 1) for k in n], indexing="ij"), axis=-1).reshape(-1, dim)
    h = (p1 - p0) / n
    pts = p0 + idx * h
    move = (rng.random(pts.shape) * 2.0 - 1.0) * jitter * h
    move[(idx == 0) | (idx == n)] = 0.0
    return pts + move


def create_delaunay_box(p0, p1, n, seed: int = 0, jitter: float = 0.4) -> Mesh:
    """An IRREGULAR simplicial mesh of the box [p0, p1] (2D: triangles, 3D: tetrahedra): jittered lattice points
    triangulated by ``scipy.spatial.Delaunay`` -- variable vertex valence, no six-tet fans, no numbering locality (the
    points are shuffled): what a mesh generator such as gmsh hands the reference (python/tests/test_cube_contact.py:15-160).
    Flat simplices among the coplanar points of the box faces are dropped (they have no volume; the mesh stays
    conforming: such a simplex only touches the boundary).  The point sets of opposite faces do NOT match."""
    from scipy.spatial import Delaunay

    n = tuple(int(k) for k in n)
    dim = len(n)
    assert dim in (2, 3)
    rng = np.random.default_rng(seed)
    pts = _delaunay_points(n, p0, p1, rng, jitter)
    pts = pts[rng.permutation(pts.shape[0])]
    tri = Delaunay(pts)
    cells = tri.simplices.astype(np.int64)
    xv = pts[cells]
    vol = np.abs(np.linalg.det(xv[:, 1:] - xv[:, :1])) / (6.0 if dim == 3 else 2.0)
    box = float(np.prod(np.asarray(p1, dtype=np.float64)[:dim] - np.asarray(p0, dtype=np.float64)[:dim]))
    keep = vol > 1e-12 * box / max(cells.shape[0], 1)
    cells = cells[keep]
    assert abs(vol[keep].sum() - box) <= 1e-10 * box, "Delaunay mesh does not fill the box"
    cells = cells[rng.permutation(cells.shape[0])]
    x = np.zeros((pts.shape[0], 3))
    x[:, :dim] = pts
    return Mesh(x, cells.astype(np.int32), "tetrahedron" if dim == 3 else "triangle")


def create_stacked_delaunay(n_top: int, n_bottom: int, seed: int = 0, theta: float = 0.0):
    """the two-body contact mesh (``create_stacked_cubes``) with irregular bodies: two Delaunay boxes whose interface
    point sets do not match.  Returns (mesh, facet_tags, cell_tags) with the same markers."""
    top = create_delaunay_box((0.0, 0.0, 1.0), (1.0, 1.0, 2.0), (n_top,) * 3, seed)
    bot = create_delaunay_box((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), (n_bottom,) * 3, seed + 1)
    return _stack_bodies(top, bot, theta)
