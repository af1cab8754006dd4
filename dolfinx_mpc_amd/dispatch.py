"""Which kernel runs for which integral: ONE table per call (matrix, vector) instead of if-chains spread over the
assemblers.  Every entry says when the kernel CAN run (``applies``: a property of the operator and
the element shapes -- the C library rejects anything else), when it is the DEFAULT choice (``default``: the measured
crossover, recorded next to it) and what it was measured against.  The first entry, in table order, that applies and
is a default wins; the remaining applicable ones follow as fall-backs (a plan that cannot be represented -- no clean
clusters, a block beyond the LDS budget -- moves on to the next).

Overrides
  ``MPCX_FORCE_KERNEL=matrix=<name>,vector=<name>``  take that kernel wherever it applies (ignored where it does not,
                                                    so that a whole test-suite can run under one setting)
  ``algorithm="atomic"`` / ``MPCX_MATRIX_ALG`` / ``MPCX_VECTOR_ALG``   the thread-per-entity kernels, no plan (API)
The older single-purpose switches keep working and map onto the table: MPCX_NO_CUBE, MPCX_NO_LEAN, MPCX_ROWPAIR=all|none,
MPCX_NO_ROWPAIR, MPCX_NO_NODEBLOCK, MPCX_OFFSET_DICT, MPCX_VECTOR_OWNER=0|1, MPCX_VCUBE_OWNER=0.

All timings: MI355X, fp64, the BASELINE configs (DESIGN.md section 4 / 5 has the full history)."""

from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Callable, List, Optional

FORM_STIFFNESS, FORM_MASS, FORM_SOURCE, FORM_ELASTICITY, FORM_FACET_MASS, FORM_FACET_SOURCE, FORM_DIV_TEST, FORM_DIV_TRIAL = range(8)
FORM_UFCX = 100


@dataclass(frozen=True)
class Ctx:
    """the facts the tables look at (one integral of one form)"""
    form: int
    tet: bool
    d0: int
    bs0: int
    d1: int
    bs1: int
    nd0: int
    nd1: int
    nq: int
    cell_integral: bool
    has_coefficient: bool
    coeff_degree: int
    all_cells: bool  # the integral runs over cells 0..n-1 in order (no entity list)
    p1_geometry: bool  # the space's dofmap IS the geometry dofmap (P1 on an affine mesh: one device array)
    same: bool  # test space is trial space, one constraint, one Dirichlet set
    tiled: bool  # the numbering carries tile hints (generators, mesh.reorder_spatial)
    builtin_form: int = -1  # an imported kernel the library also knows as a built-in operator (fem.KernelSpec.builtin): its form id
    has_transforms: bool = False  # imported kernel with dof transformations (the cluster instances do not carry the hook)


@dataclass
class Kernel:
    name: str
    applies: Callable[[Ctx], bool]
    default: Callable[[Ctx], bool]
    note: str


def _lean_ok(c: Ctx) -> bool:
    return (c.same and c.p1_geometry and c.all_cells and c.cell_integral and not c.has_coefficient and c.form != FORM_UFCX)


def _compact_context(c: Ctx) -> bool:
    """operators with a closed form per entry (csrc/mpcx_elements.hpp ElementOp::LAZY)"""
    if not c.cell_integral or c.has_coefficient:
        return False
    if c.form == FORM_STIFFNESS:
        return c.coeff_degree == 0 and c.d0 == c.d1 and c.d0 in (1, 2)
    if c.form == FORM_ELASTICITY:
        return c.d0 == c.d1 and c.d0 in (1, 2)
    if c.form == FORM_DIV_TEST:
        return c.coeff_degree == 0 and c.d0 == 2 and c.d1 == 1
    if c.form == FORM_DIV_TRIAL:
        return c.coeff_degree == 0 and c.d0 == 1 and c.d1 == 2
    return False


MATRIX: List[Kernel] = [
    Kernel("cube",
           lambda c: (_lean_ok(c) and c.form == FORM_STIFFNESS and c.tet and c.d0 == 1 and c.bs0 == 1 and c.coeff_degree == 0),
           lambda c: True,
           "matrix_cube_kernel: six-tet clusters, 46 scatter-adds per 6 cells; config 2: 1.27 ms vs 1.75 ms (rowblock_lean)"),
    Kernel("cube_el",
           lambda c: (_lean_ok(c) and c.form == FORM_ELASTICITY and c.tet and c.d0 == 1 and c.bs0 == 3),
           lambda c: True,
           "matrix_cube_elasticity_rowpair_kernel: parallelepiped clusters in closed form, one thread per (cluster, local row "
           "vertex) pair whose rows lie in the block, 414 scatter-adds per cluster instead of 864 from six element tensors; "
           "cells of other clusters through rowpair; contact elasticity (config 4): 0.86 ms vs 1.02 ms (rowpair alone); a "
           "thread per (block, cluster) slot -- closed form 1.12 ms, summing the six tensors tet by tet 1.81 ms -- loses: half "
           "the lanes of every LDS instruction are masked"),
    Kernel("p2_cube",
           lambda c: (c.form == FORM_STIFFNESS and c.tet and c.d0 == 2 and c.d1 == 2 and c.bs0 == 1 and c.bs1 == 1 and c.same
                      and c.all_cells and c.cell_integral and not c.has_coefficient and c.coeff_degree == 0),
           lambda c: False,
           "matrix_p2_cube_kernel: scalar P2 stiffness on parallelepiped clusters in closed form, one thread per (cluster, local "
           "dof) pair whose row lies in the block, 393 scatter-adds per cluster instead of 600, every lane keeps what it computes; "
           "cells of other clusters through rowblock -- measured and NOT the default: P2 Poisson 246^3 (config 5) 22.2 ms (row "
           "coefficients through scalar loads, geometry in the record, 1024 threads; 512: 30.2) and 20.8 ms (27 unrolled row "
           "bodies with folded constants, geometry per unit: 120 KB of code) against 14.1 ms (rowblock): 402 M units of ~15 "
           "entries each pay their record / coefficient fetches per unit"),
    Kernel("hex_cube",
           lambda c: (c.form == FORM_UFCX and c.builtin_form == FORM_STIFFNESS and c.same and c.p1_geometry and c.all_cells
                      and c.cell_integral and not c.has_coefficient),
           lambda c: True,
           "matrix_hex_kernel: Q1 stiffness on hexahedra, thread per (row block, cell) slot, 96-byte records, closed form on "
           "parallelepipeds; 256^3 cells: see DESIGN (generated UFCx kernel in the row blocks: 5.2 ms)"),
    Kernel("ufcx_cube",
           lambda c: (c.form == FORM_UFCX and c.tet and c.d0 == 1 and c.bs0 == 1 and c.d1 == 1 and c.bs1 == 1 and c.same
                      and c.p1_geometry and c.all_cells and c.cell_integral and not c.has_transforms),
           lambda c: True,
           "ufcx_matrix_cube_kernel (hipRTC): the imported tabulate_tensor called six times per cluster, the tensors summed per "
           "vertex pair in registers (no symmetry assumed), 46 scatter-adds per 6 cells; clusters whose cells the mesh lists in "
           "another vertex order, and cells in no cluster, through ufcx_rowblock"),
    Kernel("ufcx_pairs",
           lambda c: (c.form == FORM_UFCX and c.cell_integral and not c.has_transforms and c.nd0 <= 10 and c.nd1 <= 10
                      and 36 < c.nd0 * c.bs0 * c.nd1 * c.bs1 <= 900 and c.nd0 * c.bs0 <= 30 and c.bs0 <= 3),
           lambda c: c.bs0 == 1 and c.bs1 == 1,
           "ufcx_matrix_pairs_kernel (hipRTC, round 6): pair records for imported text -- one (entity, node row) pair per lane, a "
           "wave runs ONE row-wise copy of the text (the unused rows of the tensor are dead code in a copy); needs a kernel compiled "
           "with row-wise copies (mpcx_ufcx_rowwise).  Default for scalar spaces: P2 stiffness text 246^3 21.2 ms vs 26.7 ms "
           "(row-wise copies inside ufcx_rowblock) vs 42 ms (whole tensor per visit); NOT for blocked spaces: P2^3 stiffness text "
           "128^3 89 ms vs 24 ms, P1^3 elasticity text 1.90 vs 1.46 ms (the geometry is gathered per pair, a pair carries bs rows "
           "of bs x the columns), p div(v) 2.5 vs 2.8 ms, div(u) q 2.9 vs 2.0 ms"),
    Kernel("ufcx_rowblock", lambda c: c.form == FORM_UFCX, lambda c: True,
           "imported tabulate_tensor inside the LDS row-block kernel (hipRTC); config 2 with tests/ufcx/laplace_p1_tet.c: 2.32 ms "
           "vs 1.75 ms built-in, vs ~50 ms thread-per-entity atomics"),
    Kernel("pairs",
           lambda c: (c.form != FORM_UFCX and _compact_context(c) and not (c.form == FORM_STIFFNESS and c.bs0 > 1)
                      and c.nd0 <= 16 and c.nd1 * c.bs1 <= 32 and c.bs0 <= 3),
           lambda c: c.d0 == 2 or c.d1 == 2,
           "matrix_pairs_kernel: thread per (entity, local row) pair, one coalesced record per pair (entity, LDS slot of the row, "
           "scatter offsets) and the entity's context cached in HBM per geometry version -- no masked lanes, no per-pair "
           "geometry; round 4: P2 stiffness 246^3 and the Taylor-Hood coupling blocks, see DESIGN section 4"),
    Kernel("rowpair",
           lambda c: c.form != FORM_UFCX and _compact_context(c) and not (_lean_ok(c) and c.bs0 == 1 and c.nd0 <= 4),
           lambda c: c.d0 == 1 and c.d1 == 1 and c.bs0 > 1,
           "matrix_rowpair_kernel: thread per (entity, local row); contact elasticity (vector P1) 0.96 ms vs 1.49 ms; loses for "
           "P2 (context recomputed ten times per cell: stiffness 246^3 14.3 -> 22.6 ms, a01 1.9 -> 2.7 ms)"),
    Kernel("nodeblock",
           lambda c: c.form in (FORM_STIFFNESS, FORM_MASS, FORM_FACET_MASS) and c.bs0 > 1 and c.bs1 == c.bs0,
           lambda c: True,
           "matrix_nodeblock_kernel: one LDS value per bs x bs block of component-diagonal forms; Stokes a00 128^3 10.9 ms vs "
           "13.9 ms (per-row compact layout) vs 18.8 ms (scalar layout)"),
    Kernel("rowblock_lean", _lean_ok, lambda c: True,
           "matrix_rowblock_kernel<LEAN>: one index array, rotated local numbering, three entities in flight; config 2 without "
           "clusters 1.75 ms vs 2.0 ms (general path)"),
    Kernel("rowblock", lambda c: c.form != FORM_UFCX, lambda c: True,
           "matrix_rowblock_kernel: every built-in operator; P2 stiffness 246^3 14.3 ms, Taylor-Hood a01 / a10 1.9 ms"),
]

VECTOR: List[Kernel] = [
    Kernel("cube_own",
           lambda c: (c.form == FORM_SOURCE and c.tet and c.d0 == 1 and c.bs0 == 1 and c.coeff_degree == 0 and not c.has_coefficient
                      and c.cell_integral and c.all_cells and c.p1_geometry),
           lambda c: True,
           "vector_cube_own_kernel: thread per cluster, owner-computes row blocks, no device atomics; config 2: 2.75 ms "
           "(deterministic) vs 2.82 ms (cube_hash) vs 3.24 ms (ownblock).  Round 6, the benchmark's right-hand side with the "
           "14-point rule: clusters that are axis-aligned boxes evaluate its univariate factors on the 19 coordinates per axis "
           "the rule puts into a box (1.5 ms); on a tensor-grid mesh vector_cube_grid_kernel reads them from a table filled per "
           "launch and interval (mpcx_vector_args_t::grid_*: 0.55 ms)"),
    Kernel("cube_hash",
           lambda c: (c.form == FORM_SOURCE and c.tet and c.d0 == 1 and c.bs0 == 1 and c.coeff_degree == 0 and not c.has_coefficient
                      and c.cell_integral and c.all_cells and c.p1_geometry),
           lambda c: False,
           "vector_cube_kernel: thread per cluster, LDS hash + one device atomic per distinct dof of the workgroup (2.82 ms)"),
    Kernel("hex_own",
           lambda c: (c.form == FORM_UFCX and c.builtin_form == FORM_SOURCE and c.p1_geometry and c.all_cells and c.cell_integral
                      and not c.has_coefficient),
           lambda c: True,
           "vector_hex_own_kernel: Q1 source on hexahedra, thread per cell, owner-computes row blocks, sum-factorised basis, "
           "fast sin / exp; 256^3 cells, 27 points: see DESIGN (generated UFCx kernel with libm: 4.1 ms)"),
    Kernel("ufcx_cube_own",
           lambda c: (c.form == FORM_UFCX and c.tet and c.d0 == 1 and c.bs0 == 1 and c.p1_geometry and c.all_cells
                      and c.cell_integral and not c.has_transforms),
           lambda c: True,
           "ufcx_vector_cube_own_kernel (hipRTC): thread per cluster, six calls of the imported tabulate_tensor, owner-computes "
           "row blocks (8 LDS adds per 6 cells), no device atomics"),
    Kernel("ufcx_ownblock", lambda c: c.form == FORM_UFCX, lambda c: True,
           "imported tabulate_tensor, every entity evaluated once (its cost is unknown); config 2 with source_p1_tet.c 2.35 ms"),
    Kernel("ufcx_rowblock", lambda c: c.form == FORM_UFCX, lambda c: False, "imported tabulate_tensor, halo entities re-evaluated"),
    Kernel("ownblock",
           lambda c: c.form != FORM_UFCX,
           lambda c: c.nq > 4 and c.tiled and c.cell_integral,
           "vector_ownblock_kernel + vector_spill_reduce_kernel; P2 source 246^3 (24 points) 5.7 ms vs 6.6 ms (rowblock) vs 11.9 ms "
           "(hash); Stokes b0 1.46 vs 1.68 ms; a one-point rule loses (contact b 0.31 vs 0.28 ms)"),
    Kernel("rowblock",
           lambda c: c.form != FORM_UFCX,
           lambda c: c.nq <= (8 if (c.d0 == 2 and c.tiled) else 4)
           or (c.d0 == 2 and c.tiled and c.form == FORM_SOURCE and c.coeff_degree == 0 and c.cell_integral),
           "vector_rowblock_kernel: halo entities evaluated by every block they touch; 2.0 ms + 0.22 ms per quadrature point at "
           "256^3 P1: wins for rules of <= 4 points (contact b 0.28 ms)"),
    Kernel("hash", lambda c: c.form != FORM_UFCX, lambda c: True,
           "vector_kernel: LDS hash per workgroup + device atomics; any numbering, no plan (P1 14 points 256^3: 3.77 ms)"),
]


# table entry -> the __global__ function it launches (profiles, bench.py's per-kernel roofline lines)
FUNCTION = {
    ("matrix", "p2_cube"): "matrix_p2_cube_kernel", ("matrix", "hex_cube"): "matrix_hex_kernel", ("vector", "hex_own"): "vector_hex_own_kernel",
    ("matrix", "ufcx_cube"): "ufcx_matrix_cube_narrow_kernel", ("vector", "ufcx_cube_own"): "ufcx_vector_cube_own_kernel",
    ("matrix", "cube"): "matrix_cube_kernel", ("matrix", "cube_el"): "matrix_cube_elasticity_rowpair_kernel", ("matrix", "ufcx_rowblock"): "ufcx_matrix_rowblock_kernel",
    ("matrix", "pairs"): "matrix_pairs_kernel", ("matrix", "ufcx_pairs"): "ufcx_matrix_pairs_kernel", ("matrix", "rowpair"): "matrix_rowpair_kernel", ("matrix", "nodeblock"): "matrix_nodeblock_kernel",
    ("matrix", "rowblock_lean"): "matrix_rowblock_kernel", ("matrix", "rowblock"): "matrix_rowblock_kernel",
    ("matrix", "atomic"): "matrix_atomic_kernel", ("matrix", "ufcx_atomic"): "ufcx_matrix_kernel",
    ("vector", "cube_own"): "vector_cube_own_kernel", ("vector", "cube_hash"): "vector_cube_kernel",
    ("vector", "ufcx_ownblock"): "ufcx_vector_rowblock_kernel", ("vector", "ufcx_rowblock"): "ufcx_vector_rowblock_kernel",
    ("vector", "ownblock"): "vector_ownblock_kernel", ("vector", "rowblock"): "vector_rowblock_kernel",
    ("vector", "hash"): "vector_kernel", ("vector", "atomic"): "vector_kernel", ("vector", "ufcx_atomic"): "ufcx_vector_kernel",
}


def forced(which: str) -> Optional[str]:
    """MPCX_FORCE_KERNEL=matrix=<name>,vector=<name> (either part optional)"""
    spec = os.environ.get("MPCX_FORCE_KERNEL", "")
    for part in spec.split(","):
        k, _, v = part.strip().partition("=")
        if k == which and v:
            return v
    return None


def _legacy_matrix(c: Ctx):
    """(excluded names, preferred name) from the older switches"""
    ex, prefer = set(), None
    env = os.environ
    if env.get("MPCX_NO_CUBE"):
        ex |= {"cube", "cube_el", "hex_cube", "p2_cube", "ufcx_cube"}
    if env.get("MPCX_NO_LEAN"):
        ex |= {"cube", "cube_el", "rowblock_lean"}
    mode = env.get("MPCX_ROWPAIR", "auto")
    if env.get("MPCX_NO_ROWPAIR") or mode == "none":
        ex.add("rowpair")
    elif mode == "all":
        prefer = "rowpair"
    if env.get("MPCX_NO_NODEBLOCK"):
        ex.add("nodeblock")
    if env.get("MPCX_OFFSET_DICT"):
        ex |= {"nodeblock", "rowpair"}  # both read the direct offset table
    return ex, prefer


def _legacy_vector(c: Ctx):
    ex, prefer = set(), None
    env = os.environ
    if env.get("MPCX_NO_CUBE"):
        ex |= {"cube_own", "cube_hash", "hex_own", "ufcx_cube_own"}
    if env.get("MPCX_VCUBE_OWNER", "1") == "0":
        ex.add("cube_own")
        prefer = "cube_hash"
    owner = env.get("MPCX_VECTOR_OWNER", "auto")
    if owner == "0":
        ex |= {"ownblock", "ufcx_ownblock"}
    elif owner == "1":
        prefer = prefer or "ownblock"
    return ex, prefer


_ENV_KEYS = ("MPCX_FORCE_KERNEL", "MPCX_NO_CUBE", "MPCX_NO_LEAN", "MPCX_ROWPAIR", "MPCX_NO_ROWPAIR", "MPCX_NO_NODEBLOCK",
             "MPCX_OFFSET_DICT", "MPCX_VCUBE_OWNER", "MPCX_VECTOR_OWNER")
_memo: dict = {}


def candidates(table: List[Kernel], c: Ctx, which: str, plan_only: bool = False) -> List[str]:
    """kernel names to try, in order: the forced / preferred one (if it applies), the defaults in table order, then
    every other applicable entry.  ``plan_only``: the caller asked for the row-block family explicitly
    (algorithm="rowblock"): plan-free entries ("hash") are dropped unless nothing else applies.  Memoised per (facts,
    switches): the answer is a pure function of both and is asked for on every assembly call."""
    key = (c, which, plan_only, tuple(os.environ.get(k) for k in _ENV_KEYS))
    hit = _memo.get(key)
    if hit is None:
        if len(_memo) > 4096:
            _memo.clear()
        hit = _memo[key] = tuple(_candidates(table, c, which, plan_only))
    return list(hit)


def _candidates(table: List[Kernel], c: Ctx, which: str, plan_only: bool) -> List[str]:
    ex, prefer = (_legacy_matrix if which == "matrix" else _legacy_vector)(c)
    usable = [k for k in table if k.name not in ex and k.applies(c)]
    names = [k.name for k in usable]
    out: List[str] = []
    f = forced(which)
    for want in (f, prefer):
        if want in names and want not in out:
            out.append(want)
    for k in usable:  # every default, in table order: the first one runs, the next ones are its fall-backs
        if k.default(c) and k.name not in out:
            out.append(k.name)
    for k in usable:
        if k.name not in out:
            out.append(k.name)
    if plan_only:
        planned = [n for n in out if n != "hash"]
        out = planned or out
    if which == "vector" and prefer is None and f is None and "cube_hash" in out and "cube_own" in out:
        # the hash variant is only a fall-back of cube_own (owner plan beyond the LDS budget), keep it right behind
        out.remove("cube_hash")
        out.insert(out.index("cube_own") + 1, "cube_hash")
    return out


def describe() -> str:
    """the two tables as text (README / DESIGN)"""
    lines = []
    for title, table in (("assemble_matrix", MATRIX), ("assemble_vector", VECTOR)):
        lines.append(title)
        for k in table:
            lines.append(f"  {k.name:14s} {k.note}")
    return "\n".join(lines)
