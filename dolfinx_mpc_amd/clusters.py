"""Cell clusters for the P1 cluster kernels (include/mpcx.h MPCX_ALG_CUBE, csrc/mpcx_cubes.hip).

A cluster ("fan") is a set of six tetrahedra round one shared edge whose other vertices form a closed ring
of six -- what a box generator (ours, DOLFINx create_box) emits per cube.  In the local numbering of the
cluster kernels the shared edge is (0, 7) and the ring 1-3-2-6-4-5:

    (0,1,3,7) (0,1,7,5) (0,5,7,4) (0,3,2,7) (0,6,4,7) (0,2,6,7)

Detection (``fans_from_topology`` / the HIP kernels behind ``mesh_clusters_device``) does not depend on the
order of the cells nor on the local vertex order inside a cell -- DOLFINx reorders both: every tet names its
longest edge, tets are sorted by that key, runs of six whose other vertices close into a ring are fans.  The
kernels compute every tet's geometry from its own coordinates.  Cells that are in no fan are returned as
leftovers and go through the per-cell kernels.

``kuhn_fans`` is the older detector for meshes that keep the generator's cell order (six consecutive cells
with the pattern's local vertex order); kept as an independent check of the general one."""

from __future__ import annotations

import numpy as np


def kuhn_fans(cells: np.ndarray, ncells: int):
    """cells: (>= ncells, 4) int geometry dofmap; only cells [0, ncells) are considered, in groups
    starting at multiples of 6.  Returns (cube_verts (n, 8) int32, leftover cell ids int32)."""
    ng = ncells // 6
    c = np.asarray(cells[: ng * 6]).reshape(ng, 6, 4)
    v0, v1, v3, v7 = c[:, 0, 0], c[:, 0, 1], c[:, 0, 2], c[:, 0, 3]
    ok = (c[:, 1, 0] == v0) & (c[:, 1, 1] == v1) & (c[:, 1, 2] == v7)
    v5 = c[:, 1, 3]
    ok &= (c[:, 2, 0] == v0) & (c[:, 2, 1] == v5) & (c[:, 2, 2] == v7)
    v4 = c[:, 2, 3]
    ok &= (c[:, 3, 0] == v0) & (c[:, 3, 1] == v3) & (c[:, 3, 3] == v7)
    v2 = c[:, 3, 2]
    ok &= (c[:, 4, 0] == v0) & (c[:, 4, 2] == v4) & (c[:, 4, 3] == v7)
    v6 = c[:, 4, 1]
    ok &= (c[:, 5, 0] == v0) & (c[:, 5, 1] == v2) & (c[:, 5, 2] == v6) & (c[:, 5, 3] == v7)
    verts = np.stack([v0, v1, v2, v3, v4, v5, v6, v7], axis=1)
    # eight distinct vertices
    s = np.sort(verts, axis=1)
    ok &= (s[:, 1:] != s[:, :-1]).all(axis=1)
    groups = np.flatnonzero(~ok)
    left = (groups[:, None] * 6 + np.arange(6)[None, :]).reshape(-1)
    left = np.concatenate([left, np.arange(ng * 6, ncells)]).astype(np.int32)
    return np.ascontiguousarray(verts[ok], dtype=np.int32), left


_RING_LOCAL = (1, 3, 2, 6, 4, 5)  # local ids of the ring vertices in walking order


def long_edge_keys(x: np.ndarray, cells: np.ndarray) -> np.ndarray:
    """(vmin << 32) | vmax of every tet's longest edge (ties: the smaller pair) -- numpy restatement of
    ``tet_long_edge_kernel``"""
    c = np.asarray(cells, dtype=np.int64)
    best = np.full(c.shape[0], -1.0)
    key = np.full(c.shape[0], -1, dtype=np.int64)
    for a in range(4):
        for b in range(a + 1, 4):
            d = x[c[:, a]] - x[c[:, b]]
            ln = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            k2 = (np.minimum(c[:, a], c[:, b]) << 32) | np.maximum(c[:, a], c[:, b])
            take = (ln > best) | ((ln == best) & (k2 < key))
            best = np.where(take, ln, best)
            key = np.where(take, k2, key)
    return key


def fans_from_topology(x: np.ndarray, cells: np.ndarray, ncells: int):
    """Fans among cells [0, ncells) of any order / local vertex order (host restatement of the device detection,
    plain loops over the candidate runs: tests and small meshes).  Returns (cube_verts (n, 8) int32, leftover
    cell ids int32, ascending)."""
    cells = np.asarray(cells[:ncells], dtype=np.int64)
    keys = long_edge_keys(np.asarray(x), cells)
    order = np.argsort(keys, kind="stable")
    ks = keys[order]
    starts = np.flatnonzero(np.concatenate([[True], ks[1:] != ks[:-1]]))
    lens = np.diff(np.concatenate([starts, [ks.size]]))
    in_fan = np.zeros(ncells, dtype=bool)
    verts = []
    for p in starts[lens == 6]:
        v0, v7 = int(ks[p] >> 32), int(ks[p] & 0xFFFFFFFF)
        edges = []
        for c in order[p:p + 6]:
            o = [int(u) for u in cells[c] if u != v0 and u != v7]
            if len(o) != 2 or o[0] == o[1]:
                break
            edges.append(o)
        if len(edges) != 6:
            continue
        ring, used, good = [edges[0][0], edges[0][1]], {0}, True
        for k in range(2, 7):
            cur = ring[-1] if k < 7 else None
            nxt = None
            for t in range(1, 6):
                if t not in used and cur in edges[t]:
                    nxt = edges[t][0] if edges[t][1] == cur else edges[t][1]
                    used.add(t)
                    break
            if nxt is None:
                good = False
                break
            if k < 6:
                ring.append(nxt)
            elif nxt != ring[0]:
                good = False
        if not good or len(set(ring)) != 6 or v0 in ring or v7 in ring:
            continue
        v = [0] * 8
        v[0], v[7] = v0, v7
        for loc, r in zip(_RING_LOCAL, ring):
            v[loc] = r
        verts.append(v)
        in_fan[order[p:p + 6]] = True
    verts = np.array(verts, dtype=np.int32).reshape(-1, 8)
    return np.ascontiguousarray(verts), np.flatnonzero(~in_fan).astype(np.int32)


def mesh_clusters(mesh, ncells: int):
    """cached per (mesh, ncells): (cube_verts, leftover cells)"""
    key = ("kuhn_fans", int(ncells))
    if key not in mesh._device:
        mesh._device[key] = kuhn_fans(mesh.geometry.dofmap, int(ncells))
    return mesh._device[key]


def mesh_clusters_device(mesh, ncells: int, parallelepipeds_only: bool = False, with_cells: bool = False):
    """(cube_verts device tensor (n, 8) int32, leftover cells (host int32, ascending, possibly empty)) of cells
    [0, ncells): detected on the device from topology + edge lengths, whatever the order of the cells and of their
    local vertices (``tet_long_edge_kernel`` -> sort by key (torch: plumbing) -> ``fan_build_kernel`` -> compaction).
    MPCX_CLUSTER_DETECT=consecutive selects the older detector (six consecutive cells in the generator's pattern).
    Cached per (mesh, ncells, geometry version): the longest edge is a property of the coordinates.
    ``parallelepipeds_only``: clusters whose eight vertices are not an affine image of the cube (mpcx_cell_shapes) are
    dropped and their cells returned among the leftover ones -- for kernels that only know the closed form.
    ``with_cells``: a third result, the six cells of every cluster (device tensor (n, 6) int32)."""
    import os

    import torch

    from . import _device as D
    from . import _native

    def build():
        md = D.mesh_device(mesh)
        dm = md["x_dofmap"]
        dev = dm.device
        n = int(ncells)
        if dm.shape[1] != 4 or n < 6:
            return torch.zeros((0, 8), dtype=torch.int32, device=dev), np.arange(n, dtype=np.int32)
        L = _native.lib()
        st = D.stream_ptr()
        if os.environ.get("MPCX_CLUSTER_DETECT", "topology") == "consecutive":
            ng = n // 6
            verts = torch.empty((ng, 8), dtype=torch.int32, device=dev)
            ok = torch.empty(ng, dtype=torch.int8, device=dev)
            _native.check(L.mpcx_cube_detect(dm.data_ptr(), ng, verts.data_ptr(), ok.data_ptr(), st), "mpcx_cube_detect")
            if int(ok.sum(dtype=torch.int64).item()) == ng and n % 6 == 0:
                return verts, np.zeros(0, dtype=np.int32)
            v, left = kuhn_fans(mesh.geometry.dofmap, n)
            return D._to_dev(v, dev), left
        keys = torch.empty(n, dtype=torch.int64, device=dev)
        _native.check(L.mpcx_cluster_keys(md["x"].data_ptr(), dm.data_ptr(), n, keys.data_ptr(), st), "mpcx_cluster_keys")
        from . import _prims

        # stable radix sort on the key bits in use (rocPRIM behind the C ABI): vmin in the high word, vmax in the low one
        keys, order = _prims.sort_pairs(keys, torch.arange(n, dtype=torch.int32, device=dev), 32 + int(mesh.num_nodes).bit_length())
        verts = torch.empty((n, 8), dtype=torch.int32, device=dev)
        ok = torch.empty(n, dtype=torch.int8, device=dev)
        in_fan = torch.zeros(n, dtype=torch.int8, device=dev)
        _native.check(L.mpcx_cluster_build(n, keys.data_ptr(), order.data_ptr(), dm.data_ptr(), verts.data_ptr(), ok.data_ptr(),
                                           in_fan.data_ptr(), st), "mpcx_cluster_build")
        # corner numbering for the fans that are parallelepipeds (the ring walk starts at an arbitrary ring vertex)
        _native.check(L.mpcx_cluster_canonical(n, verts.data_ptr(), ok.data_ptr(), md["x"].data_ptr(), st), "mpcx_cluster_canonical")
        if parallelepipeds_only:
            general = torch.empty(n, dtype=torch.uint8, device=dev)  # (positions that start no fan hold garbage vertices: masked by ok)
            vsafe = torch.where(ok[:, None] != 0, verts, torch.zeros_like(verts))
            _native.check(L.mpcx_cell_shapes(n, vsafe.data_ptr(), md["x"].data_ptr(), general.data_ptr(), st), "mpcx_cell_shapes")
            bad = torch.nonzero((ok != 0) & (general != 0)).reshape(-1)
            if bad.numel() > 0:
                cells = order.long()[(bad[:, None] + torch.arange(6, device=dev)[None, :]).reshape(-1)]
                in_fan[cells] = 0
                ok[bad] = 0
            del general, vsafe
        sel = torch.nonzero(ok).reshape(-1)
        fan_cells = None
        if with_cells:
            fan_cells = order[(sel[:, None] + torch.arange(6, device=dev)[None, :]).reshape(-1)].view(-1, 6).contiguous()
        del keys, order
        verts = verts[sel].contiguous()
        if verts.shape[0] * 6 == n:
            left = np.zeros(0, dtype=np.int32)
        else:
            left = torch.nonzero(in_fan == 0).reshape(-1).to(torch.int32).cpu().numpy()
        return (verts, left, fan_cells) if with_cells else (verts, left)

    return D.cached(mesh._device, "fans_dev", (), (int(ncells), mesh.geometry.version, bool(parallelepipeds_only), bool(with_cells)), build, maxsize=3)


def mesh_clusters_ordered_device(mesh, ncells: int):
    """Clusters for IMPORTED element kernels: (cube_verts (n, 8), leftover cells (host int32, ascending), cluster cells
    (n, 6) device int32).  An imported ``tabulate_tensor`` must see every cell with the vertex order the mesh lists
    (a quadrature rule need not be symmetric), and the cluster kernels hand it tet t of a cluster as local vertices
    (0,1,3,7) (0,1,7,5) (0,5,7,4) (0,3,2,7) (0,6,4,7) (0,2,6,7): a fan found from topology qualifies when some numbering of
    its eight vertices makes its six cells read exactly so (``mpcx_cluster_ordered``; what a Kuhn box generator emits).
    The others -- cells listed in another local order -- are returned among the leftover cells and keep the per-cell
    kernels.  ``cluster cells[p][t]`` is the cell of table row t (the index of its packed coefficients)."""
    import torch

    from . import _device as D
    from . import _native

    def build():
        verts, left, cells = mesh_clusters_device(mesh, ncells, with_cells=True)
        n = int(verts.shape[0])
        if n == 0:
            return verts, left, torch.zeros((0, 6), dtype=torch.int32, device=verts.device)
        md = D.mesh_device(mesh)
        verts, cells = verts.clone(), cells.clone()
        ok = torch.empty(n, dtype=torch.int8, device=verts.device)
        _native.check(_native.lib().mpcx_cluster_ordered(n, verts.data_ptr(), cells.data_ptr(), md["x_dofmap"].data_ptr(),
                                                         ok.data_ptr(), D.stream_ptr()), "mpcx_cluster_ordered")
        nok = int(ok.sum(dtype=torch.int64).item())
        if nok < n:
            bad = cells[ok == 0].reshape(-1).cpu().numpy().astype(np.int32)
            left = np.union1d(left, bad).astype(np.int32)
            keep = torch.nonzero(ok).reshape(-1)
            verts, cells = verts[keep].contiguous(), cells[keep].contiguous()
        return verts, left, cells

    return D.cached(mesh._device, "fans_ordered", (), (int(ncells), mesh.geometry.version), build, maxsize=2)
