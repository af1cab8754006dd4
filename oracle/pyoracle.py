"""oracle/pyoracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes front-end of ``libmpc_oracle.so`` (oracle/mpc_oracle.c) plus numpy
restatements of the reference's one-off builders and of its verification
toolkit.  Only tests/, ``__graft_entry__.smoke()`` and bench.py's cpu_baseline
leg may import this module; nothing under dolfinx_mpc_amd/ does.

Restated here (integer work, numpy / small Python loops):
  cpp/MultiPointConstraint.h:36-126      -> finalize_np
  cpp/mpc_helpers.h:19-94                -> cell_to_slaves_np
  cpp/utils.h:381-496                    -> sparsity_pattern_np
  python/src/dolfinx_mpc/utils/test.py:67-149, 196-265
                                         -> gather_transformation_matrix,
                                            compare_mpc_lhs, compare_mpc_rhs

PARITY: the MPC algebra is pinned by the reference's own test identities
(K^T A K, K^T b); absolute element-tensor values are PARITY UNPINNED (FFCx is
absent; see mpc_oracle.h).
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libmpc_oracle.so")


class _Desc(C.Structure):
    _fields_ = [
        ("form", C.c_int32),
        ("celltype", C.c_int32),
        ("degree", C.c_int32),
        ("bs", C.c_int32),
        ("degree1", C.c_int32),
        ("bs1", C.c_int32),
        ("fn_id", C.c_int32),
        ("coeff_degree", C.c_int32),
        ("nq", C.c_int32),
        ("nqf", C.c_int32),
        ("qpts", C.c_void_p),
        ("qwts", C.c_void_p),
        ("fqpts", C.c_void_p),
        ("fqwts", C.c_void_p),
    ]


class _Mpc(C.Structure):
    _fields_ = [
        ("num_dofs", C.c_int32),
        ("num_slaves", C.c_int32),
        ("num_local_slaves", C.c_int32),
        ("is_slave", C.c_void_p),
        ("slaves", C.c_void_p),
        ("masters_offsets", C.c_void_p),
        ("masters", C.c_void_p),
        ("coeffs", C.c_void_p),
        ("c2s_offsets", C.c_void_p),
        ("c2s", C.c_void_p),
    ]


class _Csr(C.Structure):
    _fields_ = [
        ("nrows", C.c_int32),
        ("rowptr", C.c_void_p),
        ("cols", C.c_void_p),
        ("vals", C.c_void_p),
        ("missing", C.c_int64),
    ]


_lib = None


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
        L.oracle_assemble_matrix.argtypes = [C.POINTER(_Csr), C.c_int, C.POINTER(_Desc), C.c_int, vp, vp, vp, i64,
                                             vp, vp, C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp,
                                             vp, C.c_int, vp, C.POINTER(_Mpc), C.POINTER(_Mpc)]
        L.oracle_assemble_matrix.restype = C.c_int
        L.oracle_add_slave_diagonal.argtypes = [C.POINTER(_Csr), C.POINTER(_Mpc), dbl]
        L.oracle_add_slave_diagonal.restype = C.c_int
        L.oracle_insert_diagonal.argtypes = [C.POINTER(_Csr), vp, i64, dbl]
        L.oracle_insert_diagonal.restype = C.c_int
        L.oracle_assemble_vector.argtypes = [vp, C.c_int, C.POINTER(_Desc), C.c_int, vp, vp, i64, vp, vp, C.c_int,
                                             vp, C.c_int, C.c_int, vp, C.c_int, vp, C.POINTER(_Mpc)]
        L.oracle_assemble_vector.restype = C.c_int
        L.oracle_apply_lifting.argtypes = [vp, C.c_int, C.POINTER(_Desc), C.c_int, vp, vp, vp, i64, vp, vp, C.c_int,
                                           vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, dbl, vp, C.c_int,
                                           vp, C.POINTER(_Mpc)]
        L.oracle_apply_lifting.restype = C.c_int
        L.oracle_backsubstitution.argtypes = [C.POINTER(_Mpc), vp]
        L.oracle_homogenize.argtypes = [C.POINTER(_Mpc), vp]
        L.oracle_tabulate_one.argtypes = [C.c_int, vp, vp, vp, vp, C.c_int, C.POINTER(_Desc)]
        L.oracle_eval_fn.argtypes = [C.c_int, vp, C.c_int, vp]
        L.oracle_eval_fn.restype = dbl
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------------------
# set-up restatements
# ---------------------------------------------------------------------------
def finalize_np(num_dofs, num_owned, slaves, masters, coeffs, owners, offsets):
    """cpp/MultiPointConstraint.h:36-126 on flat arrays (single process)."""
    slaves = np.asarray(slaves, dtype=np.int32)
    is_slave = np.zeros(num_dofs, dtype=np.int8)
    is_slave[slaves] = 1
    num_masters = np.zeros(num_dofs, dtype=np.int32)
    for i, s in enumerate(slaves):
        num_masters[s] = offsets[i + 1] - offsets[i]
    moff = np.zeros(num_dofs + 1, dtype=np.int32)
    moff[1:] = np.cumsum(num_masters)
    m = np.zeros(moff[-1], dtype=np.int32)
    c = np.zeros(moff[-1], dtype=np.float64)
    o = np.zeros(moff[-1], dtype=np.int32)
    fill = np.zeros(num_dofs, dtype=np.int32)
    for i, s in enumerate(slaves):
        for j in range(offsets[i], offsets[i + 1]):
            pos = moff[s] + fill[s]
            m[pos], c[pos], o[pos] = masters[j], coeffs[j], owners[j]
            fill[s] += 1
    sorted_slaves = np.flatnonzero(is_slave).astype(np.int32)
    nloc = int(np.searchsorted(sorted_slaves, num_owned))
    return dict(is_slave=is_slave, slaves=sorted_slaves, num_local_slaves=nloc, masters_offsets=moff, masters=m,
                coeffs=c, owners=o)


def cell_to_slaves_np(dofmap, bs, num_dofs, slaves):
    """cpp/mpc_helpers.h:19-94: dof->cells for slave dofs, inverted."""
    nc, nd = dofmap.shape
    unrolled = (dofmap[:, :, None].astype(np.int64) * bs + np.arange(bs)[None, None, :]).reshape(nc, nd * bs)
    in_num_cells = np.bincount(unrolled.ravel(), minlength=num_dofs)
    num_slave_cells = np.zeros(num_dofs, dtype=np.int64)
    num_slave_cells[slaves] = in_num_cells[slaves]
    # dof -> cells (ascending cell order), only for slave dofs
    mask = num_slave_cells[unrolled] > 0
    cells_idx, _ = np.nonzero(mask)
    dofs_hit = unrolled[mask]
    order = np.lexsort((cells_idx, dofs_hit))  # by dof, then cell
    d_sorted, c_sorted = dofs_hit[order], cells_idx[order]
    # invert: for dof ascending, for its cells: append dof to cell
    order2 = np.argsort(c_sorted, kind="stable")
    c2s = d_sorted[order2].astype(np.int32)
    counts = np.bincount(c_sorted, minlength=nc)
    off = np.zeros(nc + 1, dtype=np.int32)
    off[1:] = np.cumsum(counts)
    return off, c2s


def sparsity_pattern_np(dofmap0, bs0, nblocks0, dofmap1, bs1, nblocks1, mpc0, mpc1):
    """cpp/utils.h:381-496 as a set of (row block, col block) pairs, expanded
    to a scalar CSR with sorted columns.  mpc*: dicts with c2s_offsets, c2s,
    masters_offsets, masters."""
    nc = dofmap0.shape[0]
    rows = np.repeat(dofmap0.astype(np.int64), dofmap1.shape[1], axis=1).ravel()
    cols = np.tile(dofmap1.astype(np.int64), (1, dofmap0.shape[1])).ravel()
    pr, pc = [rows], [cols]
    n0 = np.diff(mpc0["c2s_offsets"])
    n1 = np.diff(mpc1["c2s_offsets"])
    for c in np.flatnonzero((n0 > 0) | (n1 > 0)):
        col_set = list(dofmap1[c])
        for s in mpc1["c2s"][mpc1["c2s_offsets"][c] : mpc1["c2s_offsets"][c + 1]]:
            col_set += [m // bs1 for m in mpc1["masters"][mpc1["masters_offsets"][s] : mpc1["masters_offsets"][s + 1]]]
        row_set = list(dofmap0[c])
        for s in mpc0["c2s"][mpc0["c2s_offsets"][c] : mpc0["c2s_offsets"][c + 1]]:
            row_set += [m // bs0 for m in mpc0["masters"][mpc0["masters_offsets"][s] : mpc0["masters_offsets"][s + 1]]]
        R, Cc = np.meshgrid(np.array(row_set, dtype=np.int64), np.array(col_set, dtype=np.int64), indexing="ij")
        pr.append(R.ravel())
        pc.append(Cc.ravel())
    r = np.concatenate(pr)
    c = np.concatenate(pc)
    P = scipy.sparse.coo_matrix((np.ones(r.size, dtype=np.int32), (r, c)), shape=(nblocks0, nblocks1)).tocsr()
    P.sum_duplicates()
    P.sort_indices()
    # expand blocks
    if bs0 == 1 and bs1 == 1:
        return P.indptr.astype(np.int32), P.indices.astype(np.int32)
    K = scipy.sparse.kron(P, np.ones((bs0, bs1), dtype=np.int8), format="csr")
    K.sort_indices()
    return K.indptr.astype(np.int32), K.indices.astype(np.int32)


class OracleMPC:
    """Finalized constraint in the reference's dense-offset layout."""

    def __init__(self, V, fin: dict, c2s_offsets, c2s):
        self.V = V
        self.is_slave = np.ascontiguousarray(fin["is_slave"], dtype=np.int8)
        self.slaves = np.ascontiguousarray(fin["slaves"], dtype=np.int32)
        self.num_local_slaves = int(fin["num_local_slaves"])
        self.masters_offsets = np.ascontiguousarray(fin["masters_offsets"], dtype=np.int32)
        self.masters = np.ascontiguousarray(fin["masters"], dtype=np.int32)
        self.coeffs = np.ascontiguousarray(fin["coeffs"], dtype=np.float64)
        self.c2s_offsets = np.ascontiguousarray(c2s_offsets, dtype=np.int32)
        self.c2s = np.ascontiguousarray(c2s, dtype=np.int32)
        self._s = _Mpc(V.num_dofs, self.slaves.size, self.num_local_slaves, _p(self.is_slave), _p(self.slaves),
                       _p(self.masters_offsets), _p(self.masters), _p(self.coeffs), _p(self.c2s_offsets), _p(self.c2s))

    @classmethod
    def from_raw(cls, V, slaves, masters, coeffs, owners, offsets):
        nd = V.num_dofs
        nowned = V.dofmap.index_map.size_local * V.dofmap.index_map_bs
        fin = finalize_np(nd, nowned, slaves, masters, coeffs, owners, offsets)
        off, c2s = cell_to_slaves_np(V.dofmap.list, V.dofmap.bs, nd, fin["slaves"])
        return cls(V, fin, off, c2s)

    @classmethod
    def empty(cls, V):
        z = np.zeros(0, dtype=np.int32)
        return cls.from_raw(V, z, np.zeros(0, dtype=np.int64), np.zeros(0), z, np.zeros(1, dtype=np.int32))

    @classmethod
    def from_arrays(cls, V, is_slave, slaves, num_local_slaves, masters_offsets, masters, coeffs, c2s_offsets, c2s):
        fin = dict(is_slave=is_slave, slaves=slaves, num_local_slaves=num_local_slaves,
                   masters_offsets=masters_offsets, masters=masters, coeffs=coeffs)
        return cls(V, fin, c2s_offsets, c2s)

    def as_dict(self):
        return dict(c2s_offsets=self.c2s_offsets, c2s=self.c2s, masters_offsets=self.masters_offsets,
                    masters=self.masters)


def _desc(k):
    keep = [np.ascontiguousarray(a, dtype=np.float64) for a in (k.qpts, k.qwts, k.fqpts, k.fqwts)]
    d = _Desc(k.form, k.celltype, k.degree, k.bs, getattr(k, "degree1", 0) or k.degree, getattr(k, "bs1", 0) or k.bs,
              k.fn_id, k.coeff_degree, keep[1].size, keep[3].size,
              _p(keep[0]), _p(keep[1]), _p(keep[2]), _p(keep[3]))
    return d, keep


_user_libs = {}


# The descriptor objects at the end of an FFCx output file, as DOLFINx reads them (declarations: oracle/include/ufcx.h).
class _UfcxIntegral(C.Structure):
    _fields_ = [("enabled_coefficients", C.c_void_p), ("tabulate_tensor_float32", C.c_void_p),
                ("tabulate_tensor_float64", C.c_void_p), ("tabulate_tensor_complex64", C.c_void_p),
                ("tabulate_tensor_complex128", C.c_void_p), ("needs_facet_permutations", C.c_bool),
                ("coordinate_element_hash", C.c_uint64), ("domain", C.c_uint8)]


class _UfcxForm(C.Structure):
    _fields_ = [("signature", C.c_char_p), ("rank", C.c_int), ("num_coefficients", C.c_int), ("num_constants", C.c_int),
                ("original_coefficient_positions", C.POINTER(C.c_int)), ("coefficient_name_map", C.POINTER(C.c_char_p)),
                ("constant_name_map", C.POINTER(C.c_char_p)), ("finite_element_hashes", C.POINTER(C.c_uint64)),
                ("form_integrals", C.POINTER(C.POINTER(_UfcxIntegral))), ("form_integral_ids", C.POINTER(C.c_int)),
                ("form_integral_offsets", C.POINTER(C.c_int))]


def _ufcx_kernel_pointer(so, source: str, name: str):
    """the float64 tabulate_tensor ``name`` stands for, reached the way the reference reaches it: a function of the text, or
    THROUGH the objects of an FFCx output file -- ``ufcx_integral`` -> ``.tabulate_tensor_float64``; ``ufcx_form`` (or the alias
    pointer ``form_<file>_<name>``) -> ``form_integrals[k]`` -> the first integral with a float64 kernel (DOLFINx fills
    ``Form::kernel`` from exactly these members; cpp/assemble_matrix.cpp:438-439 reads it back).  Which kind of symbol a name is
    comes from its declaration in the text; the VALUES come from the compiled objects."""
    import re

    def integral_fn(obj: _UfcxIntegral):
        return obj.tabulate_tensor_float64

    def form_fn(form: _UfcxForm):
        n = form.form_integral_offsets[4]  # cell | exterior facet | interior facet | vertex: five offsets
        for i in range(n):
            fn = integral_fn(form.form_integrals[i].contents)
            if fn:
                return fn
        raise RuntimeError("oracle: the ufcx_form holds no integral with a float64 kernel")

    if not name:
        objs = re.findall(r"\bufcx_integral\s+(\w+)\s*=", source)
        if len(objs) != 1:
            raise RuntimeError(f"oracle: no kernel name given and the text holds {len(objs)} ufcx_integral objects")
        name = objs[0]
    if re.search(r"\bufcx_integral\s+%s\s*=" % re.escape(name), source):
        fn = integral_fn(_UfcxIntegral.in_dll(so, name))
    elif re.search(r"\bufcx_form\s+%s\s*=" % re.escape(name), source):
        fn = form_fn(_UfcxForm.in_dll(so, name))
    elif re.search(r"\bufcx_form\s*\*\s*%s\s*=" % re.escape(name), source):
        fn = form_fn(C.POINTER(_UfcxForm).in_dll(so, name).contents)
    else:
        return getattr(so, name)
    if not fn:
        raise RuntimeError(f"oracle: '{name}' has no float64 kernel")
    return C.c_void_p(fn)


def _load_user_kernel(k):
    """UFCx import (fem.FORM_UFCX): the kernel's C source compiled with gcc -- the same text the product
    compiles with hipRTC -- and registered as the oracle's kernel 100."""
    import hashlib
    import tempfile

    name = k.ufcx_name or ""
    key = hashlib.sha256((k.ufcx_source + name).encode()).hexdigest()[:20]
    if key not in _user_libs:
        d = os.path.join(tempfile.gettempdir(), "mpcx_oracle_ufcx")
        os.makedirs(d, exist_ok=True)
        src, so = os.path.join(d, key + ".c"), os.path.join(d, key + ".so")
        if not os.path.exists(so):
            # private names, then an atomic rename: several test workers may want the same kernel at once
            tag = "%s.%d" % (key, os.getpid())
            src, tmp = os.path.join(d, tag + ".c"), os.path.join(d, tag + ".so")
            with open(src, "w") as fh:
                fh.write("#include <stdint.h>\n#include <math.h>\n" + k.ufcx_source)
            # (-I oracle/include: the minimal ufcx.h a whole FFCx-layout file includes, oracle/include/ufcx.h)
            subprocess.run(["gcc", "-O2", "-std=c99", "-fPIC", "-shared", "-I", os.path.join(os.path.dirname(os.path.abspath(__file__)), "include"),
                            "-o", tmp, src, "-lm"], check=True)
            os.replace(tmp, so)
            os.remove(src)
        _user_libs[key] = C.CDLL(so)
    fn = _ufcx_kernel_pointer(_user_libs[key], k.ufcx_source, name)
    lib().oracle_set_user_kernel(C.cast(fn, C.c_void_p))
    # dof transformations of the element (cpp/assemble_matrix.cpp:507-508): functions of the same text + the cell permutation
    # words of the meshes; reset for kernels without any
    tr = getattr(k, "ufcx_transforms", None)
    L = lib()
    L.oracle_set_dof_transformations.argtypes = [C.c_void_p] * 4
    if tr is None:
        L.oracle_set_dof_transformations(None, None, None, None)
    else:
        t0 = None if tr[0] is None else C.cast(getattr(_user_libs[key], tr[0]), C.c_void_p)
        t1 = None if tr[1] is None else C.cast(getattr(_user_libs[key], tr[1]), C.c_void_p)
        info = _cell_info_of.get("current")
        L.oracle_set_dof_transformations(t0, t1, None if info is None else info[0].ctypes.data, None if info is None else info[1].ctypes.data)
    return 100


# the cell permutation words (uint32 per cell) of the test / trial space's mesh for the NEXT imported kernel with dof
# transformations: set by assemble_matrix / assemble_vector / apply_lifting from the form's spaces
_cell_info_of = {}


def _set_cell_info(V0, V1=None):
    i0 = getattr(V0.mesh, "cell_permutation_info", None)
    i1 = getattr((V1 or V0).mesh, "cell_permutation_info", None)
    if i0 is None:
        _cell_info_of.pop("current", None)
    else:
        _cell_info_of["current"] = (np.ascontiguousarray(i0, dtype=np.uint32), np.ascontiguousarray(i1, dtype=np.uint32))


def _which(k, fast: bool):
    """0 generic; FFCx-like fast paths only for the benchmark kernels; 100 an imported UFCx kernel."""
    if k.form == 100:
        return _load_user_kernel(k)
    if not fast or k.coeff_degree != 0 or k.bs != 1 or k.degree != 1 or k.celltype != 2:
        return 0
    if k.form == 0:
        return 1
    if k.form == 2:
        return 2
    return 0


def _ents(integ):
    e = np.ascontiguousarray(integ.entities.astype(np.int32).reshape(-1))
    return e


def create_pattern(form, mpc0: OracleMPC, mpc1: OracleMPC):
    V0, V1 = mpc0.V, mpc1.V
    return sparsity_pattern_np(V0.dofmap.list, V0.dofmap.bs, V0.dofmap.index_map.size_local + V0.dofmap.index_map.num_ghosts, V1.dofmap.list,
                               V1.dofmap.bs, V1.dofmap.index_map.size_local + V1.dofmap.index_map.num_ghosts, mpc0.as_dict(), mpc1.as_dict())


def assemble_matrix(form, mpc0: OracleMPC, mpc1: OracleMPC = None, bcs=(), diagval=1.0, pattern=None, fast=False,
                    same_space=None, out_vals=None, raw_vals=None):
    """Restates python/src/dolfinx_mpc/assemble_matrix.py:43-65 +
    cpp/assemble_matrix.cpp:662-726 on the oracle; returns scipy CSR."""
    L = lib()
    mpc1 = mpc0 if mpc1 is None else mpc1
    V0, V1 = form.function_spaces
    rowptr, cols = create_pattern(form, mpc0, mpc1) if pattern is None else pattern
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    cols = np.ascontiguousarray(cols, dtype=np.int32)
    # out_vals: caller-owned value array (zeroed here); the values are then returned as that array
    # instead of a scipy matrix (oracle/cpu_parallel.py: workers write into shared memory)
    # raw_vals: (address of the value with position 0, already zeroed) -- oracle/cpu_parallel.py hands a
    # worker a compact private array covering only the rows its cells touch; no diagonals are added then
    if raw_vals is not None:
        csr = _Csr(rowptr.size - 1, _p(rowptr), _p(cols), C.c_void_p(raw_vals), 0)
    else:
        vals = np.zeros(cols.size, dtype=np.float64) if out_vals is None else out_vals
        vals[:] = 0.0
        csr = _Csr(rowptr.size - 1, _p(rowptr), _p(cols), _p(vals), 0)
    bc0 = bc1 = None
    for bc in bcs:
        if V0.contains(bc.function_space):
            bc0 = np.zeros(V0.num_dofs, dtype=np.int8) if bc0 is None else bc0
            bc.mark_dofs(bc0)
        if V1.contains(bc.function_space):
            bc1 = np.zeros(V1.num_dofs, dtype=np.int8) if bc1 is None else bc1
            bc.mark_dofs(bc1)
    x = form.mesh.geometry.x
    xd = form.mesh.geometry.dofmap
    _set_cell_info(V0, V1)
    for integ in form.integrals:
        d, keep = _desc(integ.kernel)
        e = _ents(integ)
        w = None if integ.coeffs is None else np.ascontiguousarray(integ.coeffs, dtype=np.float64)
        cst = None if integ.constants is None else np.ascontiguousarray(integ.constants, dtype=np.float64)
        rc = L.oracle_assemble_matrix(
            C.byref(csr), _which(integ.kernel, fast), C.byref(d), integ.estride, _p(e), _p(e), _p(e),
            integ.num_entities, _p(x), _p(xd), xd.shape[1], _p(V0.dofmap.list), V0.element_ndofs, V0.dofmap.bs,
            _p(V1.dofmap.list), V1.element_ndofs, V1.dofmap.bs, _p(bc0), _p(bc1), _p(w),
            0 if w is None else w.shape[1], _p(cst), C.byref(mpc0._s), C.byref(mpc1._s))
        if rc != 0:
            raise RuntimeError(f"oracle_assemble_matrix rc={rc} missing={csr.missing}")
    if raw_vals is not None:
        if csr.missing:
            raise RuntimeError(f"oracle: {csr.missing} insertions outside the pattern")
        return None
    if same_space is None:
        same_space = V0 is V1
    if mpc0.V is mpc1.V:
        L.oracle_add_slave_diagonal(C.byref(csr), C.byref(mpc0._s), float(diagval))
    if same_space:
        for bc in bcs:
            if V0.contains(bc.function_space):
                d_all, nowned = bc.dof_indices()
                dofs = np.ascontiguousarray(d_all[:nowned], dtype=np.int32)
                L.oracle_insert_diagonal(C.byref(csr), _p(dofs), dofs.size, float(diagval))
    if csr.missing:
        raise RuntimeError(f"oracle: {csr.missing} insertions outside the pattern")
    if out_vals is not None:
        return vals
    return scipy.sparse.csr_matrix((vals, cols, rowptr), shape=(rowptr.size - 1, V1.num_dofs))


def assemble_vector(form, mpc: OracleMPC, b=None, fast=False, raw_b=None):
    """python/src/dolfinx_mpc/assemble_vector.py:79-104: zero then accumulate.
    raw_b: address of entry 0 of an already zeroed array (compact private arrays of oracle/cpu_parallel.py)."""
    L = lib()
    V = form.function_spaces[0]
    if raw_b is not None:
        b = C.c_void_p(raw_b)
    else:
        if b is None:
            b = np.zeros(V.num_dofs, dtype=np.float64)
        b[:] = 0.0
    x = form.mesh.geometry.x
    xd = form.mesh.geometry.dofmap
    _set_cell_info(V)
    for integ in form.integrals:
        d, keep = _desc(integ.kernel)
        e = _ents(integ)
        w = None if integ.coeffs is None else np.ascontiguousarray(integ.coeffs, dtype=np.float64)
        cst = None if integ.constants is None else np.ascontiguousarray(integ.constants, dtype=np.float64)
        rc = L.oracle_assemble_vector(b if raw_b is not None else _p(b), _which(integ.kernel, fast), C.byref(d), integ.estride, _p(e), _p(e),
                                      integ.num_entities, _p(x), _p(xd), xd.shape[1], _p(V.dofmap.list),
                                      V.element_ndofs, V.dofmap.bs, _p(w), 0 if w is None else w.shape[1], _p(cst),
                                      C.byref(mpc._s))
        if rc != 0:
            raise RuntimeError(f"oracle_assemble_vector rc={rc}")
    return b


def apply_lifting(b, forms, bcs, mpc: OracleMPC, x0=None, scale=1.0, fast=False):
    """python/src/dolfinx_mpc/assemble_vector.py:25-76 + cpp/lifting.h:441-483."""
    L = lib()
    x0 = [] if x0 is None else x0
    if len(x0) and len(x0) != len(forms):
        raise RuntimeError("Mismatch in size between x0 and bilinear form in assembler.")
    if len(forms) != len(bcs):
        raise RuntimeError("Mismatch in size between a and bcs in assembler.")
    for j, aj in enumerate(forms):
        if aj is None or len(bcs[j]) == 0:
            continue
        V0, V1 = aj.function_spaces
        markers = np.zeros(V1.num_dofs, dtype=np.int8)
        values = np.zeros(V1.num_dofs, dtype=np.float64)
        for bc in bcs[j]:
            bc.mark_dofs(markers)
            bc.set(values, None, 1.0)
        x0j = np.ascontiguousarray(x0[j], dtype=np.float64) if len(x0) else None
        x = aj.mesh.geometry.x
        xd = aj.mesh.geometry.dofmap
        _set_cell_info(V0, V1)
        for integ in aj.integrals:
            d, keep = _desc(integ.kernel)
            e = _ents(integ)
            w = None if integ.coeffs is None else np.ascontiguousarray(integ.coeffs, dtype=np.float64)
            cst = None if integ.constants is None else np.ascontiguousarray(integ.constants, dtype=np.float64)
            rc = L.oracle_apply_lifting(_p(b), _which(integ.kernel, fast), C.byref(d), integ.estride, _p(e), _p(e),
                                        _p(e), integ.num_entities, _p(x), _p(xd), xd.shape[1], _p(V0.dofmap.list),
                                        V0.element_ndofs, V0.dofmap.bs, _p(V1.dofmap.list), V1.element_ndofs,
                                        V1.dofmap.bs, _p(markers), _p(values), _p(x0j), float(scale), _p(w),
                                        0 if w is None else w.shape[1], _p(cst), C.byref(mpc._s))
            if rc != 0:
                raise RuntimeError(f"oracle_apply_lifting rc={rc}")
    return b


def backsubstitution(mpc: OracleMPC, u):
    lib().oracle_backsubstitution(C.byref(mpc._s), _p(u))
    return u


def homogenize(mpc: OracleMPC, u):
    lib().oracle_homogenize(C.byref(mpc._s), _p(u))
    return u


def tabulate_one(kernel, coordinate_dofs, w=None, c=None, local_facet=0, which=0):
    """element tensor of one cell (known-answer tests)."""
    ndofs = {(1, 1): 3, (1, 2): 6, (2, 1): 4, (2, 2): 10}
    n = ndofs[(kernel.celltype, kernel.degree)] * kernel.bs
    d1, b1 = getattr(kernel, "degree1", 0) or kernel.degree, getattr(kernel, "bs1", 0) or kernel.bs
    n1 = ndofs[(kernel.celltype, d1)] * b1
    rank1 = kernel.form in (2, 5)
    A = np.zeros(n if rank1 else n * n1, dtype=np.float64)
    d, keep = _desc(kernel)
    cd = np.ascontiguousarray(coordinate_dofs, dtype=np.float64)
    w = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
    c = None if c is None else np.ascontiguousarray(c, dtype=np.float64)
    lib().oracle_tabulate_one(which, _p(A), _p(w), _p(c), _p(cd), local_facet, C.byref(d))
    return A if rank1 else A.reshape(n, n1)


# ---------------------------------------------------------------------------
# verification toolkit (python/src/dolfinx_mpc/utils/test.py)
# ---------------------------------------------------------------------------
def gather_transformation_matrix(mpc: OracleMPC):
    """K (n x (n - n_slaves)), utils/test.py:67-149, single process."""
    n = mpc.V.num_dofs
    all_slaves = mpc.slaves[: mpc.num_local_slaves]
    is_slave = np.zeros(n, dtype=bool)
    is_slave[all_slaves] = True
    shift = np.cumsum(is_slave)  # number of slaves <= dof
    K_val, rows, cols = [], [], []
    for s in all_slaves:
        m = mpc.masters[mpc.masters_offsets[s] : mpc.masters_offsets[s + 1]]
        cf = mpc.coeffs[mpc.masters_offsets[s] : mpc.masters_offsets[s + 1]]
        if len(m) > 0:
            for master, coeff in zip(m, cf):
                K_val.append(coeff)
                rows.append(s)
                cols.append(master - np.sum(master > all_slaves))
        else:
            K_val.append(1)
            rows.append(s)
            cols.append(s - np.sum(s > all_slaves))
    free = np.flatnonzero(~is_slave)
    rows = np.concatenate([np.array(rows, dtype=np.int64), free])
    cols = np.concatenate([np.array(cols, dtype=np.int64), free - shift[free]])
    vals = np.concatenate([np.array(K_val, dtype=np.float64), np.ones(free.size)])
    return scipy.sparse.coo_matrix((vals, (rows, cols)), shape=(n, n - all_slaves.size)).tocsr()


def compare_csr(A, B, atol=1e-10):
    """utils/test.py:196-199"""
    diff = np.abs(A - B)
    assert diff.max() < atol, f"max diff {diff.max()}"


def compare_mpc_lhs(A_org, A_mpc, mpc: OracleMPC, atol=5e3 * np.finfo(np.float64).resolution):
    """utils/test.py:202-242: K^T A_org K == A_mpc without slave rows/cols."""
    K = gather_transformation_matrix(mpc)
    KTAK = K.T @ A_org @ K
    n = mpc.V.num_dofs
    free = np.flatnonzero(~np.isin(np.arange(n), mpc.slaves[: mpc.num_local_slaves]))
    red = A_mpc.tocsr()[free, :][:, free]
    compare_csr(KTAK, red, atol=atol)


def compare_mpc_rhs(b_org, b, mpc: OracleMPC):
    """utils/test.py:245-265"""
    K = gather_transformation_matrix(mpc)
    reduced_b = K.T @ b_org
    n = mpc.V.num_dofs
    slaves = mpc.slaves[: mpc.num_local_slaves]
    free = np.flatnonzero(~np.isin(np.arange(n), slaves))
    assert np.allclose(b[slaves], 0)
    assert np.allclose(b[free], reduced_b)
