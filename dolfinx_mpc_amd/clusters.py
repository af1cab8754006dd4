"""Cell clusters for the P1 cluster kernels (include/mpcx.h MPCX_ALG_CUBE, csrc/mpcx_cubes.hip).

A cluster ("Kuhn fan") is a run of six consecutive tetrahedra that share one edge and have eight
vertices in the pattern every structured box generator emits per cube:

    (0,1,3,7) (0,1,7,5) (0,5,7,4) (0,3,2,7) (0,6,4,7) (0,2,6,7)      local vertex b: bit0 = x, bit1 = y, bit2 = z

Detection is purely topological (vertex ids of consecutive cells); the kernels compute every tet's
geometry from its own coordinates.  Cells that are not part of such a run are returned as leftovers
and keep going through the per-cell kernels."""

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


def mesh_clusters(mesh, ncells: int):
    """cached per (mesh, ncells): (cube_verts, leftover cells)"""
    key = ("kuhn_fans", int(ncells))
    if key not in mesh._device:
        mesh._device[key] = kuhn_fans(mesh.geometry.dofmap, int(ncells))
    return mesh._device[key]


def mesh_clusters_device(mesh, ncells: int):
    """(cube_verts device tensor (n, 8) int32, leftover cells (host int32, possibly empty)) -- detection by the HIP
    kernel ``cube_detect_kernel`` on the device-resident geometry dofmap; only when some group is not a fan does
    the (slower) host path run to compact the clusters and list the leftovers.  Cached per (mesh, ncells)."""
    import torch

    from . import _device as D
    from . import _native

    def build():
        dm = D.mesh_device(mesh)["x_dofmap"]
        dev = dm.device
        ng = int(ncells) // 6
        if dm.shape[1] != 4 or ng == 0:
            return torch.zeros((0, 8), dtype=torch.int32, device=dev), np.arange(ncells, dtype=np.int32)
        verts = torch.empty((ng, 8), dtype=torch.int32, device=dev)
        ok = torch.empty(ng, dtype=torch.int8, device=dev)
        _native.check(_native.lib().mpcx_cube_detect(dm.data_ptr(), ng, verts.data_ptr(), ok.data_ptr(), D.stream_ptr()),
                      "mpcx_cube_detect")
        if int(ok.sum(dtype=torch.int64).item()) == ng and ncells % 6 == 0:
            return verts, np.zeros(0, dtype=np.int32)
        v, left = kuhn_fans(mesh.geometry.dofmap, int(ncells))
        return D._to_dev(v, dev), left

    return D.cached(mesh._device, "kuhn_fans_dev", (), int(ncells), build)
