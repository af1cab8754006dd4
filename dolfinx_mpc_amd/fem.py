"""Flat-array stand-ins for the DOLFINx objects the reference's hot path reads.

``FunctionSpace``/``DofMap``/``IndexMap``/``DirichletBC``/``Form`` expose only
what cpp/assemble_matrix.cpp, cpp/assemble_vector.cpp and cpp/lifting.h consume
(``dofmap.cell_dofs``, ``index_map.size_local``, ``bs``, bc markers/values,
integration domains, packed coefficients and constants).  Single process:
``num_ghosts == 0`` and local == global numbering.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Optional, Sequence

import numpy as np

from .mesh import Mesh
from .quadrature import facet_cell_name, make_quadrature

# form kinds (shared numbering with include/mpcx.h)
FORM_STIFFNESS = 0
FORM_MASS = 1
FORM_SOURCE = 2
FORM_ELASTICITY = 3
FORM_FACET_MASS = 4
FORM_FACET_SOURCE = 5
FORM_DIV_TEST = 6  # c * p div(v): vector test space, scalar trial space
FORM_DIV_TRIAL = 7  # c * div(u) q: scalar test space, vector trial space
FORM_UFCX = 100  # an imported UFCx tabulate_tensor (C source), include/mpcx.h mpcx_ufcx_compile

CELL_TRIANGLE = 1
CELL_TETRAHEDRON = 2
CELL_HEXAHEDRON = 3  # Q1 (built-in kernels + generated text), Q2 (generated text)
CELL_QUADRILATERAL = 4  # Q1-Q3, element kernels imported as generated UFCx C text (codegen.generate_general)
_CELL_ID = {"triangle": CELL_TRIANGLE, "tetrahedron": CELL_TETRAHEDRON, "hexahedron": CELL_HEXAHEDRON,
            "quadrilateral": CELL_QUADRILATERAL}

# analytic right-hand sides (evaluated at physical quadrature points)
FN_ONE = 0
FN_BENCH_PERIODIC = 1  # python/benchmarks/bench_periodic.py:85-89
FN_SIN2D = 2
FN_POLY3 = 3
FN_LINEAR = 4
FN_CONSTANT_VEC = 5  # f = constants[1:1+bs] (constants[0] is the scale)

_FN_DEGREE = {FN_ONE: 0, FN_BENCH_PERIODIC: 4, FN_SIN2D: 4, FN_POLY3: 3, FN_LINEAR: 1, FN_CONSTANT_VEC: 0}


class IndexMap:
    def __init__(self, size_local: int, num_ghosts: int = 0):
        self.size_local = int(size_local)
        self.num_ghosts = int(num_ghosts)
        self.size_global = int(size_local)
        self.local_range = (0, int(size_local))


class DofMap:
    def __init__(self, cell_dofs: np.ndarray, num_blocks: int, bs: int, num_ghosts: int = 0):
        self.list = np.ascontiguousarray(cell_dofs, dtype=np.int32)  # (num_cells, nd) blocked dofs
        self.index_map = IndexMap(num_blocks, num_ghosts)
        self.index_map_bs = int(bs)
        self.bs = int(bs)

    def cell_dofs(self, c: int) -> np.ndarray:
        return self.list[c]


def _lagrange_ndofs(cell_name: str, degree: int) -> int:
    return {("tetrahedron", 1): 4, ("tetrahedron", 2): 10, ("triangle", 1): 3, ("triangle", 2): 6,
            ("hexahedron", 1): 8}[(cell_name, degree)]


def kuhn_edge_global_ids(ga: np.ndarray, gb: np.ndarray, n1, num_global_nodes: int) -> np.ndarray:
    """Global id of the edges (ga, gb) of the structured Kuhn mesh whose global node id is
    (k*ny1 + j)*nx1 + i (``n1`` = nx1 = ny1, or the pair (nx1, ny1)): every edge runs from its lower
    node in one of 7 directions (1,0,0) (0,1,0) (0,0,1) (1,1,0) (0,1,1) (1,0,1) (1,1,1), so
    id = Gn + 7*lower + direction."""
    g0, g1 = np.minimum(ga, gb).astype(np.int64), np.maximum(ga, gb).astype(np.int64)
    d = g1 - g0
    nx1, ny1 = (n1, n1) if np.isscalar(n1) else n1
    sx, sy, sz = 1, nx1, nx1 * ny1
    code = {sx: 0, sy: 1, sz: 2, sy + sx: 3, sz + sy: 4, sz + sx: 5, sz + sy + sx: 6}
    ud = np.unique(d)
    direction = np.array([code[int(v)] for v in ud])[np.searchsorted(ud, d)]
    return num_global_nodes + 7 * g0 + direction


class FunctionSpace:
    """Lagrange P1/P2 space on simplices, Q1 on hexahedra, optionally blocked (``shape=(bs,)``)."""

    def __init__(self, mesh: Mesh, element=("Lagrange", 1), shape: Optional[tuple] = None):
        family, degree = element[0], int(element[1])
        if family not in ("Lagrange", "CG", "P"):
            raise NotImplementedError(f"element family {family}")
        # general Lagrange elements (elements.py): degree 3 on triangles / tetrahedra, Q1-Q3 on quadrilaterals, Q2 / Q3 on hexahedra -- the
        # cell / degree sweep of python/tests/test_matrix_assembly.py:23-26; their forms run generated (imported) kernels
        self.general = (mesh.cell_name == "quadrilateral" and degree in (1, 2, 3, 4)) or (mesh.cell_name in ("triangle", "tetrahedron") and degree in (3, 4)) \
            or (mesh.cell_name == "hexahedron" and degree in (2, 3, 4))
        if not self.general:
            if degree not in (1, 2):
                raise NotImplementedError("Lagrange degree 1-4")
            if mesh.cell_name == "hexahedron" and degree != 1:
                raise NotImplementedError("hexahedra: Q1-Q4")
        self.mesh = mesh
        self.degree = degree
        bs = 1 if not shape else int(np.prod(shape))  # vector (d,) and tensor (d, d) valued spaces: blocked dofs
        self.value_shape = tuple(int(k) for k in shape) if shape else ()
        nghost = 0
        self._dof_coords = None
        # partitioned meshes (dolfinx_mpc_amd.distributed): global id and lowest node
        # plane of every dof block, used by the interface exchange
        self.dof_global = None
        self.dof_plane = None
        self.dof_tile_offsets = None  # first dof block of every tile of a tiled numbering (hint for row blocks)
        if self.general:
            from . import elements

            if mesh.num_owned_nodes != mesh.num_nodes:
                raise NotImplementedError("general Lagrange elements on partitioned meshes")
            cell_dofs, nblocks, self._dof_coords = elements.build_dofmap(mesh, degree)
            self._dof_coords_version = mesh.geometry.version
        elif degree == 1:
            cell_dofs = mesh.geometry.dofmap.copy()
            self.dof_tile_offsets = mesh.node_tile_offsets
            nblocks = mesh.num_owned_nodes
            nghost = mesh.num_nodes - mesh.num_owned_nodes
            if mesh.node_global is not None:
                self.dof_global = mesh.node_global
                self.dof_plane = getattr(mesh, "node_plane", None)
        elif mesh.num_owned_nodes == mesh.num_nodes:
            cell_edges, ev = mesh.edges()
            cell_dofs = np.concatenate([mesh.geometry.dofmap, cell_edges + mesh.num_nodes], axis=1)
            nblocks = mesh.num_nodes + int(cell_edges.max()) + 1
            if mesh.node_tile_offsets is not None:
                # tiled numbering: an edge dof is numbered next to its lower end node instead of after
                # all nodes, so that the dofs of a spatial tile form one contiguous range (row blocks of
                # the assembly kernels, distinct dofs per workgroup) -- numbering is not part of the
                # reference's contract (DOLFINx reorders dofs for locality as well)
                nn = mesh.num_nodes
                owner = np.concatenate([np.arange(nn, dtype=np.int64), np.minimum(ev[:, 0], ev[:, 1]).astype(np.int64)])
                kind = np.concatenate([np.zeros(nn, dtype=np.int8), np.ones(ev.shape[0], dtype=np.int8)])
                order = np.lexsort((kind, owner))  # old ids in new order
                new_of_old = np.empty(order.size, dtype=np.int64)
                new_of_old[order] = np.arange(order.size)
                cell_dofs = new_of_old[cell_dofs]
                x = mesh.geometry.x
                self._dof_coords = np.concatenate([x, 0.5 * (x[ev[:, 0]] + x[ev[:, 1]])], axis=0)[order]
                self.dof_tile_offsets = new_of_old[mesh.node_tile_offsets]
        else:
            cell_dofs, nblocks, nghost = self._p2_on_slab(mesh)
        self.dofmap = DofMap(cell_dofs, nblocks, bs, nghost)
        self._device = {}

    def _p2_on_slab(self, mesh: Mesh):
        """P2 dofs on a z-slab mesh: an edge dof belongs to the rank that owns the lower of
        its two node planes (the only rank whose cells touch a vertical edge; the same rule as
        for nodes when the edge lies in a plane).  Numbering: owned nodes, owned edges, ghost
        nodes, ghost edges.  Global edge id = Gn + 7 * (lower global node) + direction."""
        part = mesh.partition
        if part["axis"] != 2 or part.get("bodies", 1) != 1:
            raise NotImplementedError("P2 on partitioned meshes: single box cut along z only")
        (nx, ny, nz), (l0, l1) = part["n"], part["layers"]
        rank, world = part["rank"], part["world"]
        cell_edges, ev = mesh.edges()
        g = mesh.node_global
        plane = g // ((nx + 1) * (ny + 1))
        edge_global = kuhn_edge_global_ids(g[ev[:, 0]], g[ev[:, 1]], (nx + 1, ny + 1), (nx + 1) * (ny + 1) * (nz + 1))
        edge_plane = np.minimum(plane[ev[:, 0]], plane[ev[:, 1]])
        last = rank == world - 1

        def owned(p):
            return (p >= l0) & ((p < l1) | (last & (p == l1)))

        eo = owned(edge_plane)
        n_on, n_nodes = mesh.num_owned_nodes, mesh.num_nodes
        n_oe = int(eo.sum())
        edge_new = np.empty(ev.shape[0], dtype=np.int64)
        node_new = np.arange(n_nodes, dtype=np.int64)
        if mesh.node_tile_offsets is not None:
            # tiled numbering, as on an unpartitioned mesh: an owned edge dof is numbered next to its lowest OWNED end node,
            # so that the owned dofs of a spatial tile form one contiguous range (without it the P2 kernels of a slab ran at
            # half speed: 8.6 instead of 5.4 ms for the matrix of half of config 5)
            lo, hi = np.minimum(ev[eo, 0], ev[eo, 1]).astype(np.int64), np.maximum(ev[eo, 0], ev[eo, 1]).astype(np.int64)
            own_node = np.where(lo < n_on, lo, hi)  # (the end node in the owned planes; the lower index if both are)
            assert (own_node < n_on).all()
            owner = np.concatenate([np.arange(n_on, dtype=np.int64), own_node])
            kind = np.concatenate([np.zeros(n_on, dtype=np.int8), np.ones(n_oe, dtype=np.int8)])
            order = np.lexsort((kind, owner))  # old owned ids (nodes, then owned edges) in new order
            pos = np.empty(order.size, dtype=np.int64)
            pos[order] = np.arange(order.size)
            node_new[:n_on] = pos[:n_on]
            edge_new[eo] = pos[n_on:]
            t_off = np.asarray(mesh.node_tile_offsets, dtype=np.int64)
            self.dof_tile_offsets = np.where(t_off < n_on, node_new[np.minimum(t_off, n_on - 1)], t_off + n_oe)
        else:
            edge_new[eo] = n_on + np.arange(n_oe)
        edge_new[~eo] = n_on + n_oe + (n_nodes - n_on) + np.arange(ev.shape[0] - n_oe)
        node_new[n_on:] += n_oe
        cell_dofs = np.concatenate([node_new[mesh.geometry.dofmap], edge_new[cell_edges]], axis=1)
        ntot = n_nodes + ev.shape[0]
        self.dof_global = np.empty(ntot, dtype=np.int64)
        self.dof_plane = np.empty(ntot, dtype=np.int64)
        self.dof_global[node_new], self.dof_plane[node_new] = g, plane
        self.dof_global[edge_new], self.dof_plane[edge_new] = edge_global, edge_plane
        x = mesh.geometry.x
        self._dof_coords = np.empty((ntot, 3))
        self._dof_coords[node_new] = x
        self._dof_coords[edge_new] = 0.5 * (x[ev[:, 0]] + x[ev[:, 1]])
        # dofs of the upper interface plane (ghosts whose partial sums go to rank + 1)
        self.dof_send_up = (self.dof_plane == l1) & (np.arange(ntot) >= n_on + n_oe) if not last else np.zeros(ntot, bool)
        return cell_dofs, n_on + n_oe, ntot - n_on - n_oe

    @property
    def element_ndofs(self) -> int:
        return self.dofmap.list.shape[1]

    @property
    def num_dofs(self) -> int:
        """unrolled local (+ghost) dofs"""
        m = self.dofmap.index_map
        return (m.size_local + m.num_ghosts) * self.dofmap.index_map_bs

    def tabulate_dof_coordinates(self) -> np.ndarray:
        gv = self.mesh.geometry.version
        if self.general and getattr(self, "_dof_coords_version", 0) != gv:
            from . import elements

            self._dof_coords = elements.build_dofmap(self.mesh, self.degree)[2]
            self._dof_coords_version = gv
        if self._dof_coords is not None and getattr(self, "_dof_coords_version", 0) != gv:
            # the mesh was moved: recompute from the dofmap (P1: the nodes; P2: nodes and edge midpoints)
            x = self.mesh.geometry.x
            if self.degree == 1 and self.dofmap.list.shape == self.mesh.geometry.dofmap.shape \
                    and np.array_equal(self.dofmap.list, self.mesh.geometry.dofmap):
                self._dof_coords = x
            else:
                from .mesh import local_edges

                le = local_edges(self.mesh.cell_name)
                nv = self.mesh.geometry.dofmap.shape[1]
                xc = x[self.mesh.geometry.dofmap]  # (nc, nv, 3)
                out = np.empty((self.dofmap.list.max() + 1, 3))
                out[self.dofmap.list[:, :nv]] = xc
                if self.degree == 2:
                    out[self.dofmap.list[:, nv:]] = 0.5 * (xc[:, le[:, 0]] + xc[:, le[:, 1]])
                self._dof_coords = out
            self._dof_coords_version = gv
        if self._dof_coords is None:
            x = self.mesh.geometry.x
            if self.degree == 1:
                self._dof_coords = x
            else:
                _, ev = self.mesh.edges()
                self._dof_coords = np.concatenate([x, 0.5 * (x[ev[:, 0]] + x[ev[:, 1]])], axis=0)
            self._dof_coords_version = gv
        return self._dof_coords

    def contains(self, other: "FunctionSpace") -> bool:
        return other is self

    def clone(self) -> "FunctionSpace":
        """a distinct space object with the same mesh, element and dof numbering (dolfinx ``FunctionSpace.clone``,
        python/tests/test_multispace_mpc.py:31): constraints and boundary conditions of the two do not mix"""
        return FunctionSpace(self.mesh, ("Lagrange", self.degree), self.value_shape or None)


def functionspace(mesh: Mesh, element, shape: Optional[tuple] = None) -> FunctionSpace:
    if len(element) == 3 and shape is None:
        shape = element[2]
    return FunctionSpace(mesh, element[:2], shape)


class _Vector:
    """Host dof array of a ``Function``.  ``array`` hands out the writable numpy array, and a caller may keep
    that view and write through it at any time, so nothing here tries to track changes: consumers that
    keep a packed / device copy (coefficients, Dirichlet values) COMPARE the current values with the copy
    they uploaded on every assembly call and refresh when they differ -- the reference packs coefficients
    and reads boundary values on every call (cpp/assemble_matrix.cpp:587-589, cpp/lifting.h:166-180)."""

    def __init__(self, n: int, dtype=np.float64):
        self._data = np.zeros(n, dtype=dtype)

    @property
    def array(self) -> np.ndarray:
        return self._data

    @array.setter
    def array(self, value):
        self._data[:] = value


class Constant:
    """``dolfinx.fem.Constant`` stand-in: ``value`` may be changed between assemblies; the
    assemblers pack it on every call (cpp/assemble_matrix.cpp:583-585)."""

    def __init__(self, value):
        v = np.atleast_1d(np.asarray(value))
        self.value = v.astype(np.complex128 if np.iscomplexobj(v) else np.float64).copy()


class Function:
    """Nodal coefficient (packed per cell like dolfinx ``pack_coefficients``,
    cpp/assemble_matrix.cpp:587-589)."""

    def __init__(self, V: FunctionSpace, dtype=np.float64):
        self.function_space = V
        self.x = _Vector(V.num_dofs, dtype)

    @property
    def dtype(self):
        return self.x._data.dtype

    def interpolate(self, f: Callable[[np.ndarray], np.ndarray]):
        V = self.function_space
        vals = np.asarray(f(V.tabulate_dof_coordinates().T), dtype=self.x._data.dtype)
        bs = V.dofmap.bs
        if bs == 1:
            self.x.array[:] = vals.reshape(-1)
        else:
            self.x.array[:] = vals.reshape(bs, -1).T.reshape(-1)


class DirichletBC:
    """Dirichlet condition on blocked dofs (all components, or one ``component``).

    Provides what cpp/assemble_matrix.cpp:688-705 and cpp/lifting.h:166-180 use:
    ``mark_dofs`` and ``set``.
    """

    def __init__(self, value, dofs: np.ndarray, V: FunctionSpace, component: Optional[int] = None):
        self.function_space = V
        bs = V.dofmap.bs
        dofs = np.asarray(dofs, dtype=np.int32).reshape(-1)
        if component is None:
            self._dofs = (dofs[:, None] * bs + np.arange(bs, dtype=np.int32)[None, :]).reshape(-1)
        else:
            self._dofs = dofs * bs + int(component)
        self._dofs = np.ascontiguousarray(self._dofs, dtype=np.int32)
        self.value = value

    def dof_indices(self):
        """(unrolled dofs, number of owned ones first) like dolfinx DirichletBC.dof_indices"""
        V = self.function_space
        nowned = V.dofmap.index_map.size_local * V.dofmap.index_map_bs
        owned = self._dofs < nowned
        if not owned.all():
            self._dofs = np.concatenate([self._dofs[owned], self._dofs[~owned]])
        return self._dofs, int(owned.sum())

    def mark_dofs(self, markers: np.ndarray):
        markers[self._dofs] = 1

    def values_at_dofs(self) -> np.ndarray:
        """current boundary value at each of ``dof_indices()[0]`` (read from the live ``value``)"""
        v = self.value
        if isinstance(v, Function):
            return v.x._data[self._dofs]
        if isinstance(v, Constant):
            v = v.value
        v = np.asarray(v)
        v = v.astype(np.complex128 if np.iscomplexobj(v) else np.float64, copy=False)
        if v.ndim == 0 or v.size == 1:
            return np.full(self._dofs.size, v.reshape(-1)[0])
        bs = self.function_space.dofmap.bs
        return v.reshape(-1)[self._dofs % bs] if v.size == bs else v.reshape(-1)[self._dofs]

    def set(self, values: np.ndarray, x0=None, alpha: float = 1.0):
        values[self._dofs] = alpha * self.values_at_dofs()


def dirichletbc(value, dofs, V: FunctionSpace, component: Optional[int] = None) -> DirichletBC:
    return DirichletBC(value, dofs, V, component)


def locate_dofs_geometrical(V: FunctionSpace, marker: Callable[[np.ndarray], np.ndarray]) -> np.ndarray:
    """Blocked dofs whose coordinate satisfies ``marker(x)``, x of shape (3, n)
    (python/benchmarks/bench_periodic.py:57)."""
    x = V.tabulate_dof_coordinates()
    return np.flatnonzero(np.asarray(marker(x.T), dtype=bool)).astype(np.int32)


def locate_dofs_topological(V: FunctionSpace, entity_dim: int, entities: np.ndarray) -> np.ndarray:
    """Blocked dofs in the closure of the given facets, (cell, local_facet) pairs
    (python/benchmarks/bench_contact_3D.py:222 ``locate_dofs_topological(V, fdim, mt.find(5))``):
    the facet's vertices and, for P2, the edges between them."""
    from .mesh import local_edges, local_facets

    mesh = V.mesh
    assert entity_dim == mesh.tdim - 1, "facets only"
    ents = np.asarray(entities, dtype=np.int64).reshape(-1, 2)
    if getattr(V, "general", False):
        from . import elements

        closure = elements.facet_closure_dofs(mesh.cell_name, V.degree)
        out = [V.dofmap.list[ents[ents[:, 1] == f, 0]][:, loc].reshape(-1) for f, loc in enumerate(closure) if (ents[:, 1] == f).any()]
        return np.unique(np.concatenate(out)).astype(np.int32) if out else np.zeros(0, dtype=np.int32)
    lf, le = local_facets(mesh.cell_name), local_edges(mesh.cell_name)
    nv = mesh.geometry.dofmap.shape[1]
    out = []
    for f in range(lf.shape[0]):
        sel = ents[ents[:, 1] == f, 0]
        if sel.size == 0:
            continue
        loc = list(lf[f])
        if V.degree == 2:  # edges whose two end vertices lie on the facet
            loc += [nv + e for e in range(le.shape[0]) if le[e][0] in lf[f] and le[e][1] in lf[f]]
        out.append(V.dofmap.list[sel][:, loc].reshape(-1))
    if not out:
        return np.zeros(0, dtype=np.int32)
    return np.unique(np.concatenate(out)).astype(np.int32)


@dataclass
class KernelSpec:
    """Description of one element kernel (stands in for an FFCx ``tabulate_tensor``)."""

    form: int
    celltype: int
    degree: int
    bs: int
    fn_id: int = 0
    coeff_degree: int = 0
    qpts: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))
    qwts: np.ndarray = field(default_factory=lambda: np.zeros(0))
    fqpts: np.ndarray = field(default_factory=lambda: np.zeros((0, 2)))
    fqwts: np.ndarray = field(default_factory=lambda: np.zeros(0))
    degree1: int = 0  # trial space (0 = same as test space)
    bs1: int = 0
    ufcx_source: str = ""  # FORM_UFCX: C source of the tabulate_tensor function and its name
    ufcx_name: str = ""
    # FORM_UFCX: the same integral as a built-in operator of the library, where it has one (hexahedra: the cluster
    # kernels of MPCX_ALG_CUBE take the bulk of the cells; constrained cells, lifting and the plan-free algorithm
    # keep running the imported kernel)
    builtin: Optional["KernelSpec"] = None
    # FORM_UFCX: names of the element's dof transformations defined in ufcx_source, (test space, trial space transposed), or
    # None (every Lagrange element) -- include/mpcx.h mpcx_ufcx_desc_t::transform0_name; cell_info comes from the mesh
    # (Mesh.cell_permutation_info)
    ufcx_transforms: Optional[tuple] = None


class Integral:
    """One integral of a form: kind, integration entities, element kernel, and the SOURCES of the
    packed data the kernel reads -- ``coefficient`` (a ``Function``, a list of Functions, an already
    packed array, or None) and ``constant`` (a ``Constant``, raw numbers, or None).  ``coeffs`` /
    ``constants`` pack them when read, like dolfinx ``pack_coefficients`` / ``pack_constants`` do on every
    assembly call (cpp/assemble_matrix.cpp:583-589): for every entity the cell dofs of every coefficient
    in turn, unrolled ``dof * bs + k`` for blocked coefficient spaces."""

    def __init__(self, itype: str, entities: np.ndarray, kernel: KernelSpec, coefficient=None, constant=None):
        self.itype = itype  # "cell" | "exterior_facet"
        self.entities = entities  # cells int32[n] or (cell, local_facet) int32[n, 2]
        self.kernel = kernel
        self.coefficient = coefficient
        self.constant = constant

    @property
    def coefficient_functions(self) -> list:
        """the ``Function`` objects behind the packed coefficients (empty for none / a pre-packed array)"""
        f = self.coefficient
        if f is None or isinstance(f, np.ndarray):
            return []
        return list(f) if isinstance(f, (list, tuple)) else [f]

    @property
    def cstride(self) -> int:
        """packed coefficient values per entity (``coeffs.shape[1]``), without packing"""
        f = self.coefficient
        if f is None:
            return 0
        if isinstance(f, np.ndarray):
            return int(f.shape[1])
        return int(sum(g.function_space.element_ndofs * g.function_space.dofmap.bs for g in self.coefficient_functions))

    @property
    def coeffs(self) -> Optional[np.ndarray]:
        """float64[n, cstride] packed coefficient dofs of the entities' cells, or None; packed from the
        CURRENT dof values on every read"""
        f = self.coefficient
        if f is None:
            return None
        if isinstance(f, np.ndarray):  # already packed by the caller
            return f
        cells = self.cells
        parts = []
        for g in self.coefficient_functions:
            Vc = g.function_space
            bs = Vc.dofmap.bs
            dofs = Vc.dofmap.list[cells].astype(np.int64)
            if bs > 1:
                dofs = (dofs[:, :, None] * bs + np.arange(bs)[None, None, :]).reshape(dofs.shape[0], -1)
            parts.append(g.x._data[dofs])
        return np.ascontiguousarray(parts[0] if len(parts) == 1 else np.concatenate(parts, axis=1))

    @property
    def constants(self) -> Optional[np.ndarray]:
        c = self.constant
        if c is None:
            return None
        v = np.atleast_1d(np.asarray(c.value if isinstance(c, Constant) else c))
        return v.astype(np.complex128 if np.iscomplexobj(v) else np.float64, copy=False)

    @property
    def estride(self) -> int:
        return 1 if self.itype == "cell" else 2

    @property
    def num_entities(self) -> int:
        return self.entities.shape[0]

    @property
    def cells(self) -> np.ndarray:
        return self.entities if self.itype == "cell" else self.entities[:, 0]


class Form:
    """Compiled-form stand-in: ``rank``, ``function_spaces``, integrals."""

    def __init__(self, function_spaces: Sequence[FunctionSpace], integrals: Sequence[Integral], dtype=np.float64):
        self.function_spaces = list(function_spaces)
        self.rank = len(self.function_spaces)
        self.integrals = list(integrals)
        self.mesh = self.function_spaces[0].mesh
        self._device = {}
        # scalar type of the assembled tensor (``dolfinx.fem.form(a, dtype=...)``): float64 runs the tuned kernels,
        # float32 / complex64 / complex128 the general ones (``set_dtype`` / the ``dtype`` argument of ``fem.form``)
        self.dtype = np.dtype(dtype)

    def set_dtype(self, dtype) -> "Form":
        if np.dtype(dtype) != self.dtype:
            self.dtype = np.dtype(dtype)
            self._device.clear()
        return self

    def __add__(self, other: "Form") -> "Form":
        assert self.rank == other.rank
        for a, b in zip(self.function_spaces, other.function_spaces):
            assert a is b
        return Form(self.function_spaces, self.integrals + other.integrals, np.result_type(self.dtype, other.dtype))


def form(f: "Form", dtype=np.float64) -> "Form":
    """``dolfinx.fem.form(a, dtype=...)``: the scalar type the form is assembled in"""
    return f.set_dtype(dtype)


def _coefficient_degree(coefficient: Optional[Function]) -> int:
    if coefficient is None:
        return 0
    Vc = coefficient.function_space
    assert Vc.dofmap.bs == 1, "only scalar coefficients"
    return Vc.degree


def _cells_or_all(mesh: Mesh, cells) -> np.ndarray:
    if cells is None:
        # owned cells only, like a DOLFINx Form's default cell domain: ONE read-only array 0..n-1 per mesh -- consumers
        # recognise the object (``entities is mesh._all_cells``) and skip the entity indirection without comparing
        # 100 M indices
        n = mesh.num_owned_cells
        cached = getattr(mesh, "_all_cells", None)
        if cached is None or cached.shape[0] != n:
            cached = np.arange(n, dtype=np.int32)
            cached.flags.writeable = False
            mesh._all_cells = cached
        return cached
    return np.ascontiguousarray(cells, dtype=np.int32)


def _constants(c):
    """a ``Constant`` is kept by reference (live value); raw numbers are frozen here"""
    if c is None or isinstance(c, Constant):
        return c
    v = np.atleast_1d(np.asarray(c))
    return v.astype(np.complex128 if np.iscomplexobj(v) else np.float64).copy()


def _cell_kernel(V: FunctionSpace, form: int, qdeg: int, fn_id: int = 0, coeff_degree: int = 0) -> KernelSpec:
    name = V.mesh.cell_name
    q, w = make_quadrature(name, qdeg)
    return KernelSpec(form, _CELL_ID[name], V.degree, V.dofmap.bs, fn_id, coeff_degree, q, w)


def _facet_kernel(V: FunctionSpace, form: int, qdeg: int, fn_id: int = 0) -> KernelSpec:
    name = V.mesh.cell_name
    q, w = make_quadrature(facet_cell_name(name), qdeg)
    return KernelSpec(form, _CELL_ID[name], V.degree, V.dofmap.bs, fn_id, 0, fqpts=q, fqwts=w)


def fn_c_expression(fn_id: int) -> str:
    """C text of the analytic right-hand side ``fn_id`` in x[0..2], the component k and the constants c -- the same
    functions as the built-in ``eval_fn`` (csrc/mpcx_elements.hpp), every cell type, one table (the generated
    hexahedron kernels and the imported-kernel twins of the tests both read it)"""
    from .codegen import BENCH_PERIODIC_F

    pi = "3.14159265358979323846"
    table = {
        FN_ONE: "1.0",
        FN_BENCH_PERIODIC: BENCH_PERIODIC_F,
        FN_SIN2D: f"sin(2.0 * {pi} * x[0]) * sin({pi} * x[1]) + 0.3 * (k + 1)",
        FN_POLY3: "1.0 + 2.0 * x[0] + 3.0 * x[1] * x[1] - x[2] * x[2] * x[2] + x[0] * x[1] * x[2] + 0.5 * k * x[0]",
        FN_LINEAR: "(k + 1) * (1.0 + x[0] - 2.0 * x[1] + 0.5 * x[2])",
        FN_CONSTANT_VEC: "c[1 + k]",
    }
    if fn_id not in table:
        raise NotImplementedError(f"analytic source function {fn_id}")
    return table[fn_id]


def _hex_form(V, kind: str, constant=None, coefficient: Optional[Function] = None, cells=None, fn_id: int = FN_ONE,
              quadrature_degree: Optional[int] = None) -> Form:
    """Forms on hexahedra: the element kernel is generated as UFCx C text (codegen.generate_hex: Q1, trilinear
    geometry, tensor Gauss rule) and imported like an FFCx kernel -- there is no built-in hexahedron kernel."""
    from .codegen import gauss_hex, generate_hex

    if coefficient is not None:
        Vc = coefficient.function_space
        assert Vc.mesh is V.mesh and Vc.degree == 1 and Vc.dofmap.bs == 1, "hexahedra: scalar Q1 coefficients"
    if kind == "source":
        if fn_id == FN_CONSTANT_VEC:
            raise NotImplementedError("hexahedra: FN_CONSTANT_VEC sources (the generated kernel takes c[0] as the scale only)")
        fexpr = fn_c_expression(fn_id)
        qdeg = (1 + _FN_DEGREE[fn_id]) if quadrature_degree is None else quadrature_degree
    else:
        fexpr, qdeg = "1.0", (2 if quadrature_degree is None else quadrature_degree)
    if coefficient is not None:
        qdeg += 1
    src, name = generate_hex(kind, V.dofmap.bs, gauss_hex(qdeg), use_constant=constant is not None and kind != "elasticity",
                             fexpr=fexpr, coefficient=coefficient is not None)
    name_q = f"{name}_f{fn_id}"
    src = src.replace(name, name_q)
    form = form_ufcx([V] if kind == "source" else [V, V], src, name_q, "cell", cells, coefficient, constant)
    # the integrals the library also knows as built-in hexahedron operators (csrc/mpcx_cubes.hip: matrix_hex_kernel,
    # vector_hex_own_kernel): scalar stiffness with the 2 x 2 x 2 rule, scalar sources with f = 1 or the benchmark's f
    if V.dofmap.bs == 1 and coefficient is None:
        pts, wts = gauss_hex(qdeg)
        if kind == "stiffness" and wts.size == 8:
            form.integrals[0].kernel.builtin = KernelSpec(FORM_STIFFNESS, CELL_HEXAHEDRON, 1, 1, 0, 0, pts, wts)
        elif kind == "source" and wts.size in (1, 8, 27) and fn_id in (FN_ONE, FN_BENCH_PERIODIC):
            form.integrals[0].kernel.builtin = KernelSpec(FORM_SOURCE, CELL_HEXAHEDRON, 1, 1, fn_id, 0, pts, wts)
    return form


def _general_form(V, kind: str, constant=None, coefficient: Optional[Function] = None, cells=None, fn_id: int = FN_ONE,
                  quadrature_degree: Optional[int] = None) -> Form:
    """Forms on the general Lagrange elements (elements.py): the kernel is generated as UFCx C text
    (codegen.generate_general: basis from baked tables, geometry of degree 1 evaluated at every point) and imported like
    an FFCx kernel.  Rule: exact for the integrand on affine cells -- 2 (p - 1) for stiffness on simplices, 2 p on tensor
    cells, 2 p for mass, p + deg(f) for sources (+ the coefficient's degree); Gauss per variable on tensor cells."""
    from . import elements
    from .codegen import gauss_tensor, generate_general

    cell, p = V.mesh.cell_name, V.degree
    cd = 0
    if coefficient is not None:
        Vc = coefficient.function_space
        assert Vc.mesh is V.mesh and Vc.dofmap.bs == 1, "scalar coefficients on the same mesh"
        cd = Vc.degree
    simplex = elements.is_simplex(cell)
    if kind == "source":
        fexpr, base = fn_c_expression(fn_id), p + _FN_DEGREE[fn_id]
    else:
        # per-variable degree of the integrand on an affine cell: a derivative lowers the degree only in ITS variable on
        # tensor cells (Q1 stiffness needs the 2 x 2 rule: one point leaves the hourglass modes in the kernel)
        fexpr, base = "1.0", (2 * (p - 1) if (kind == "stiffness" and simplex) else 2 * p)
    qdeg = base + cd if quadrature_degree is None else quadrature_degree
    rule = make_quadrature(cell, qdeg) if simplex else gauss_tensor(elements.tdim(cell), qdeg)
    src, name = generate_general(kind, cell, p, V.dofmap.bs, rule, use_constant=constant is not None, fexpr=fexpr,
                                 coefficient_degree=cd)
    name_q = f"{name}_f{fn_id}"
    src = src.replace(name, name_q)
    return form_ufcx([V] if kind == "source" else [V, V], src, name_q, "cell", cells, coefficient, constant)


def form_stiffness(V, constant=None, coefficient: Optional[Function] = None, cells=None) -> Form:
    """a(u, v) = c * w * inner(grad(u), grad(v)) dx  (bench_periodic.py:84;
    test_mpc_pipeline.py:45 with coefficient and constant)."""
    if getattr(V, "general", False):
        return _general_form(V, "stiffness", constant, coefficient, cells)
    if V.mesh.cell_name == "hexahedron":
        return _hex_form(V, "stiffness", constant, coefficient, cells)
    cells = _cells_or_all(V.mesh, cells)
    cd = _coefficient_degree(coefficient)
    k = _cell_kernel(V, FORM_STIFFNESS, 2 * (V.degree - 1) + cd, coeff_degree=cd)
    return Form([V, V], [Integral("cell", cells, k, coefficient, _constants(constant))])


def form_mass(V, constant=None, coefficient: Optional[Function] = None, cells=None) -> Form:
    if getattr(V, "general", False):
        return _general_form(V, "mass", constant, coefficient, cells)
    if V.mesh.cell_name == "hexahedron":
        return _hex_form(V, "mass", constant, coefficient, cells)
    cells = _cells_or_all(V.mesh, cells)
    cd = _coefficient_degree(coefficient)
    k = _cell_kernel(V, FORM_MASS, 2 * V.degree + cd, coeff_degree=cd)
    return Form([V, V], [Integral("cell", cells, k, coefficient, _constants(constant))])


def form_elasticity(V, mu: float, lmbda: float, cells=None) -> Form:
    """a(u, v) = inner(sigma(u), grad(v)) dx, sigma = 2 mu eps(u) + lambda tr(eps(u)) I
    (python/benchmarks/bench_contact_3D.py:257-269)."""
    assert V.dofmap.bs == V.mesh.tdim
    if V.mesh.cell_name == "hexahedron":
        return _hex_form(V, "elasticity", np.array([mu, lmbda], dtype=np.float64), None, cells)
    cells = _cells_or_all(V.mesh, cells)
    k = _cell_kernel(V, FORM_ELASTICITY, 2 * (V.degree - 1))
    return Form([V, V], [Integral("cell", cells, k, None, np.array([mu, lmbda], dtype=np.float64))])


def _general_div(V, Q, test: bool, constant, cells) -> Form:
    """the Taylor-Hood coupling blocks on general elements (any pair of degrees on one cell type): generated kernels"""
    from . import elements
    from .codegen import gauss_tensor, generate_general

    cell = V.mesh.cell_name
    simplex = elements.is_simplex(cell)
    qdeg = (V.degree - 1 + Q.degree) if simplex else (V.degree + Q.degree)
    rule = make_quadrature(cell, qdeg) if simplex else gauss_tensor(elements.tdim(cell), qdeg)
    if test:
        src, name = generate_general("div_test", cell, V.degree, V.dofmap.bs, rule, use_constant=True, degree1=Q.degree)
        return form_ufcx([V, Q], src, name, "cell", cells, None, constant)
    src, name = generate_general("div_trial", cell, Q.degree, 1, rule, use_constant=True, degree1=V.degree)
    return form_ufcx([Q, V], src, name, "cell", cells, None, constant)


def _is_general_pair(V, Q) -> bool:
    return getattr(V, "general", False) or getattr(Q, "general", False) or V.mesh.cell_name in ("hexahedron", "quadrilateral")


def form_div_test(V, Q, constant=-1.0, cells=None) -> Form:
    """a(p, v) = c * p div(v) dx, rows = V (vector), cols = Q (scalar): the ``a01`` block
    of python/tests/test_stokes_channelflow.py:77-80 with c = -1."""
    assert V.mesh is Q.mesh and V.dofmap.bs == V.mesh.tdim and Q.dofmap.bs == 1
    if _is_general_pair(V, Q):
        return _general_div(V, Q, True, constant, cells)
    cells = _cells_or_all(V.mesh, cells)
    k = _cell_kernel(V, FORM_DIV_TEST, V.degree - 1 + Q.degree)
    k.degree1, k.bs1 = Q.degree, 1
    return Form([V, Q], [Integral("cell", cells, k, None, _constants(constant))])


def form_div_trial(Q, V, constant=-1.0, cells=None) -> Form:
    """a(u, q) = c * div(u) q dx, rows = Q (scalar), cols = V (vector): the ``a10`` block."""
    assert V.mesh is Q.mesh and V.dofmap.bs == V.mesh.tdim and Q.dofmap.bs == 1
    if _is_general_pair(V, Q):
        return _general_div(V, Q, False, constant, cells)
    cells = _cells_or_all(Q.mesh, cells)
    k = _cell_kernel(Q, FORM_DIV_TRIAL, V.degree - 1 + Q.degree)
    k.degree1, k.bs1 = V.degree, V.dofmap.bs
    return Form([Q, V], [Integral("cell", cells, k, None, _constants(constant))])


def form_source(V, fn_id: int = FN_ONE, constant=None, coefficient: Optional[Function] = None, cells=None,
                quadrature_degree: Optional[int] = None) -> Form:
    """L(v) = c * w * inner(f, v) dx with analytic f (bench_periodic.py:85-91).
    Non-polynomial f: estimated degree +2 per UFL's rule -> P1: 5."""
    if getattr(V, "general", False):
        return _general_form(V, "source", constant, coefficient, cells, fn_id, quadrature_degree)
    if V.mesh.cell_name == "hexahedron":
        return _hex_form(V, "source", constant, coefficient, cells, fn_id, quadrature_degree)
    cells = _cells_or_all(V.mesh, cells)
    cd = _coefficient_degree(coefficient)
    fdeg = _FN_DEGREE[fn_id]
    qdeg = V.degree + fdeg + cd if quadrature_degree is None else quadrature_degree
    k = _cell_kernel(V, FORM_SOURCE, qdeg, fn_id, cd)
    return Form([V], [Integral("cell", cells, k, coefficient, _constants(constant))])


def form_ufcx(spaces: Sequence[FunctionSpace], source: str, function_name: Optional[str] = None, itype: str = "cell", entities=None,
              coefficient=None, constant=None, builtin: Optional[KernelSpec] = None, dof_transformations=None) -> Form:
    """A form whose element kernel is an imported UFCx ``tabulate_tensor`` given as C SOURCE (what FFCx
    writes to disk; the reference calls the compiled function through a pointer,
    cpp/assemble_matrix.cpp:438-439).  ``spaces`` = [V] (linear form) or [V0, V1] (bilinear form: rows V0,
    columns V1); the function must write the row-major [nd0*bs0][nd1*bs1] tensor of ONE entity with the
    blocked dof index i*bs + k, accumulate into A (handed over zeroed) and read the local facet of an
    exterior-facet integral from ``entity_local_index[0]``.  ``entities``: cells (default all owned cells) or
    (cell, local_facet) pairs for ``itype="exterior_facet"``.  ``coefficient``: a ``Function`` or a list of them
    in the order the form declares its coefficients; ``w`` then holds, per entity, the cell dofs of each in turn,
    unrolled ``dof * bs + k`` for blocked spaces (dolfinx ``pack_coefficients``).  The kernel is compiled for
    gfx950 with hipRTC and runs inside the LDS row-block kernels (or the per-entity kernels with device atomics
    when ``algorithm="atomic"``).

    ``source`` may be a WHOLE FFCx output file (``#include <ufcx.h>``, the functions, then the ``ufcx_integral`` /
    ``ufcx_form`` objects and the alias ``form_<file>_<name>``): ``function_name`` then names the function, a
    ``ufcx_integral`` object, a ``ufcx_form`` object or its alias -- the kernel is found through the objects, the way DOLFINx
    (and through it the reference) finds it -- or is None for a file with one integral (include/mpcx.h mpcx_ufcx_resolve).

    ``builtin``: the ``KernelSpec`` of a built-in operator the caller (a form generator) states this text implements.
    On simplices the library CHECKS the statement at first use -- both kernels are evaluated on a sample of the form's
    entities on the device and must agree to 1e-12 of the largest entry -- and, if it holds, runs the built-in operator
    in place of the text (cell-cluster kernels, closed forms); otherwise, and with MPCX_UFCX_BUILTIN=0, the text runs
    everywhere.  (Hexahedra: the built-in Q1 kernels take the bulk of the cells, the text keeps the constrained ones.)"""
    spaces = list(spaces)
    V0 = spaces[0]
    V1 = spaces[1] if len(spaces) > 1 else None
    if itype == "cell":
        ents = _cells_or_all(V0.mesh, entities)
    else:
        ents = np.ascontiguousarray(entities, dtype=np.int32).reshape(-1, 2)
    k = KernelSpec(FORM_UFCX, _CELL_ID[V0.mesh.cell_name], V0.degree, V0.dofmap.bs, ufcx_source=source, ufcx_name=function_name or "")
    if V1 is not None:
        k.degree1, k.bs1 = V1.degree, V1.dofmap.bs
    k.builtin = builtin
    if dof_transformations is not None:
        # (name of the test space's transformation, name of the trial space's transposed one) -- functions of ``source`` with
        # the shape ``void T(double* A, const uint32_t* cell_info, int32_t cell, int32_t n)`` that the reference applies right
        # after the kernel call (cpp/assemble_matrix.cpp:507-508, cpp/assemble_vector.cpp:184); the cell permutation words
        # are the meshes' ``cell_permutation_info`` (uint32 per cell)
        t = tuple(dof_transformations) if not isinstance(dof_transformations, str) else (dof_transformations,)
        k.ufcx_transforms = (t[0], t[1] if len(t) > 1 else None)
        for V in spaces:
            if getattr(V.mesh, "cell_permutation_info", None) is None:
                raise ValueError("form_ufcx(dof_transformations=...): mesh.cell_permutation_info (uint32 per cell) is not set")
    return Form(spaces, [Integral(itype, ents, k, coefficient, _constants(constant))])


def form_generated(kind: str, V: FunctionSpace, fn_id: int = FN_ONE, constant=None, coefficient: Optional[Function] = None,
                   cells=None, mu: float = 1.0, lmbda: float = 0.0) -> Form:
    """The library's stand-in for running FFCx on a form: ``kind`` = "stiffness" | "mass" | "source" | "elasticity" on
    Lagrange P1 / P2 simplices, written as UFCx C text in the shape FFCx gives its output (``codegen.generate``: baked
    rule and tables, a loop over the points, libm calls in the source function) and imported through ``form_ufcx`` --
    with the built-in operator of the same integral attached as its stated twin (``form_ufcx(builtin=...)``)."""
    from .codegen import generate

    cell = V.mesh.cell_name
    if cell not in ("triangle", "tetrahedron"):
        raise NotImplementedError("form_generated: simplices (hexahedra: the form_* functions already generate their kernels)")
    ref = {"stiffness": lambda: form_stiffness(V, constant, coefficient, cells), "mass": lambda: form_mass(V, constant, coefficient, cells),
           "source": lambda: form_source(V, fn_id, constant, coefficient, cells),
           "elasticity": lambda: form_elasticity(V, mu, lmbda, cells)}[kind]()
    integ = ref.integrals[0]
    k = integ.kernel
    has_c = integ.constant is not None or (kind == "source" and fn_id == FN_CONSTANT_VEC)
    src, name = generate(kind, cell, V.degree, V.dofmap.bs, (k.qpts, k.qwts), coefficient_degree=k.coeff_degree,
                         use_constant=has_c and kind != "elasticity", fexpr=fn_c_expression(fn_id) if kind == "source" else "1.0")
    return form_ufcx([V] if kind == "source" else [V, V], src, name, "cell", integ.entities, integ.coefficient, integ.constant, builtin=k)


def forms_nonlinear_poisson(V, u: Function, f_expr: str, quadrature_degree: Optional[int] = None):
    """(F, J) of the quasi-linear Poisson problem of python/tests/test_nonlinear_assembly.py:76-81,
    F = inner((1 + u^2) grad(u), grad(v)) dx - inner(f, v) dx and its Gateaux derivative J, as imported (generated)
    kernels with ``u`` (a Function on ``V``) as coefficient -- what ``fem.form(F)`` / ``fem.form(ufl.derivative(F, u))``
    give the reference through FFCx.  ``f_expr``: C expression of f in ``x[0..2]``.  Rule: degree 4 p (the integrand
    u^2 grad(u).grad(v) has degree 4 p - 2 on affine cells)."""
    from . import elements
    from .codegen import gauss_tensor, generate_nonlinear_poisson

    if u.function_space is not V or V.dofmap.bs != 1:
        raise ValueError("forms_nonlinear_poisson: a scalar space and the unknown on the same space")
    cell, p = V.mesh.cell_name, V.degree
    qdeg = 4 * p if quadrature_degree is None else quadrature_degree
    rule = make_quadrature(cell, qdeg) if elements.is_simplex(cell) else gauss_tensor(elements.tdim(cell), qdeg)
    out = []
    for which, spaces in (("F", [V]), ("J", [V, V])):
        src, name = generate_nonlinear_poisson(which, cell, p, rule, f_expr)
        out.append(form_ufcx(spaces, src, name, "cell", None, u, None))
    return tuple(out)


def form_facet_mass(V, facets: np.ndarray, constant=None) -> Form:
    """a(u, v) = c * inner(u, v) ds over the given exterior facets."""
    k = _facet_kernel(V, FORM_FACET_MASS, 2 * V.degree)
    return Form([V, V], [Integral("exterior_facet", np.ascontiguousarray(facets, dtype=np.int32), k, None, _constants(constant))])


def form_facet_source(V, facets: np.ndarray, fn_id: int = FN_ONE, constant=None) -> Form:
    """L(v) = c * inner(f, v) ds (python/tests/test_surface_integral.py:64-66 traction term)."""
    fdeg = _FN_DEGREE[fn_id]
    k = _facet_kernel(V, FORM_FACET_SOURCE, V.degree + fdeg, fn_id)
    return Form([V], [Integral("exterior_facet", np.ascontiguousarray(facets, dtype=np.int32), k, None, _constants(constant))])
