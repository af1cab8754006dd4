"""Device mirrors of the flat host arrays (torch is the allocator / stream
provider; the kernels only ever see raw pointers through the C ABI)."""

from __future__ import annotations

import os

from collections import OrderedDict

import numpy as np

from . import _native
from .fem import Form, FunctionSpace, Integral


def _update_dev(old, a: np.ndarray, dev):
    """``a`` on the device, written INTO ``old`` when that is a device tensor of the same shape and dtype (argument blocks,
    cached plans and captured HIP graphs hold its address), a new tensor otherwise"""
    import torch

    if old is not None and isinstance(old, torch.Tensor) and tuple(old.shape) == tuple(a.shape) and old.is_contiguous():
        src = torch.from_numpy(np.ascontiguousarray(a))
        if src.dtype == old.dtype:
            # The buffer is shared by everything built from this form (the matrix call on the library's matrix stream, lifting
            # on the vector stream, captured graphs): order the overwrite against ALL library streams -- kernels still queued
            # there read the old values first, kernels launched there afterwards see the new ones (nothing did)
            from .la import _side

            cur = torch.cuda.current_stream(old.device) if old.device.type == "cuda" else None
            others = [st for st in _side.values() if cur is not None and st.cuda_stream != cur.cuda_stream and st.device == old.device]
            for st in others:
                cur.wait_stream(st)
            old.copy_(src)
            for st in others:
                st.wait_stream(cur)
            return old
    return _to_dev(a, dev)


def _to_dev(a: np.ndarray, dev):
    import warnings

    import torch

    a = np.ascontiguousarray(a)
    if a.flags.writeable:
        return torch.from_numpy(a).to(dev)
    with warnings.catch_warnings():  # read-only host arrays (mesh.geometry.x): the host tensor is only the source of the copy
        warnings.simplefilter("ignore", UserWarning)
        t = torch.from_numpy(a)
    return t.to(dev) if torch.device(dev).type != "cpu" else t.clone()


def _timed_build(kind, build):
    """MPCX_TIMING=1: print what every one-off builder (plans, masks, device mirrors) costs"""
    import os
    import sys
    import time

    if not os.environ.get("MPCX_TIMING"):
        return build()
    import torch

    if torch.cuda.is_available():
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    val = build()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    print(f"[mpcx timing] {kind}: {time.perf_counter() - t0:.3f}s", file=sys.stderr, flush=True)
    return val


def _settle():
    """a freshly built cached object may be shared with calls running on another stream (la.side_stream): finish its
    construction before anybody can see it (cache misses are set-up, not the timed path)"""
    import torch

    if torch.cuda.is_available():
        torch.cuda.current_stream().synchronize()


def cached(store: dict, kind: str, objs, extra, build, maxsize: int = 8):
    """Small LRU cache inside ``store`` (the ``_device`` / ``_cache`` / ``_plans`` dict of the object
    that owns the derived data), keyed by the IDENTITY of ``objs`` plus the hashable ``extra``.
    Every entry keeps strong references to its key objects, so their ids cannot be handed to new
    objects while the entry lives, and a hit is confirmed with ``is``.  (Keys made of bare ``id()``s
    can silently match a new form / bc / constraint that inherited the id of a collected one.)"""
    od = store.setdefault(("objcache", kind), OrderedDict())
    objs = tuple(objs)
    key = (tuple(id(o) for o in objs), extra)
    hit = od.get(key)
    if hit is not None and all(a is b for a, b in zip(hit[0], objs)):
        od.move_to_end(key)
        return hit[1]
    val = _timed_build(kind, build)
    _settle()
    od[key] = (objs, val)
    while len(od) > maxsize:
        od.popitem(last=False)
    return val


def mesh_device(mesh):
    dev = _native.require_gpu()
    key = str(dev)
    if key not in mesh._device:
        mesh._device[key] = {
            "x": _to_dev(mesh.geometry.x, dev),
            "x_dofmap": _to_dev(mesh.geometry.dofmap, dev),
            "x_version": mesh.geometry.version,
        }
        _settle()
    d = mesh._device[key]
    if d["x_version"] != mesh.geometry.version:
        # the mesh was moved (mesh.geometry.x = ...): same tensor, new values, so that argument blocks and plans
        # that hold its address stay valid (the reference re-reads x on every call, cpp/assemble_matrix.cpp:495-501);
        # assemblies still running on the side streams read the old coordinates first
        from .la import wait_assembly

        wait_assembly()
        d["x"].copy_(_to_dev(mesh.geometry.x, dev))
        d["x_version"] = mesh.geometry.version
    return d


def space_device(V: FunctionSpace):
    dev = _native.require_gpu()
    key = str(dev)
    if key not in V._device:
        xd = V.mesh.geometry.dofmap
        dm = V.dofmap.list
        if dm.shape == xd.shape and np.array_equal(dm, xd):
            # P1: dofs are numbered like the nodes -> ONE device array for both maps; the row-block
            # kernel sees dofmap0 == x_dofmap and skips the second 16 B/entity read
            V._device[key] = {"dofmap": mesh_device(V.mesh)["x_dofmap"]}
        else:
            V._device[key] = {"dofmap": _to_dev(dm, dev)}
        _settle()
    return V._device[key]


_ufcx_handles = {}
# switches read by mpcx_ufcx_compile (csrc/mpcx_ufcx.cpp): part of the handle cache key
_UFCX_COMPILE_ENV = ("MPCX_UFCX_FP", "MPCX_UFCX_LIBM", "MPCX_UFCX_CUBE_THREADS", "MPCX_UFCX_CUBE_PIPE", "MPCX_UFCX_CUBE_WAVES",
                     "MPCX_UFCX_VCUBE_THREADS", "MPCX_UFCX_VCUBE_WAVES", "MPCX_UFCX_ROWWISE", "MPCX_UFCX_RB_THREADS")


def ufcx_compile(k, form: Form):
    """handle of the hipRTC-compiled kernels for an imported UFCx function (include/mpcx.h mpcx_ufcx_compile),
    one per (source, name, element shapes); compilation needs no device"""
    V0 = form.function_spaces[0]
    V1 = form.function_spaces[1] if form.rank == 2 else None
    nv = form.mesh.geometry.dofmap.shape[1]
    tr = getattr(k, "ufcx_transforms", None) or (None, None)
    key = (k.ufcx_source, k.ufcx_name, form.rank, V0.element_ndofs, V0.dofmap.bs,
           0 if V1 is None else V1.element_ndofs, 0 if V1 is None else V1.dofmap.bs, nv, tr,
           tuple(os.environ.get(e) for e in _UFCX_COMPILE_ENV))
    if key not in _ufcx_handles:
        d = _native.UfcxDescT(k.ufcx_source.encode(), k.ufcx_name.encode() if k.ufcx_name else None, form.rank, V0.element_ndofs, V0.dofmap.bs,
                              0 if V1 is None else V1.element_ndofs, 0 if V1 is None else V1.dofmap.bs, nv,
                              None if tr[0] is None else tr[0].encode(), None if tr[1] is None else tr[1].encode())
        L = _native.lib()
        h = L.mpcx_ufcx_compile(d)
        if not h:
            raise RuntimeError("mpcx_ufcx_compile failed: " + L.mpcx_last_error().decode())
        _ufcx_handles[key] = h
    return _ufcx_handles[key]


def resolve_builtin_twins(form: Form) -> None:
    """Imported kernels with a stated built-in twin on simplices (fem.form_ufcx(builtin=...)): evaluate both on a sample of
    the entities with the plan-free kernels, and if they agree to 1e-12 of the largest entry let the built-in operator
    stand in for the text from now on (the integral's ``kernel`` becomes the twin; the text stays in ``kernel_imported``).
    A twin that does not check out is dropped.  MPCX_UFCX_BUILTIN=0: no substitution.  Once per form."""
    if getattr(form, "_twins_resolved", False):
        return
    form._twins_resolved = True
    swapped = False

    from .fem import Form as _Form
    from .fem import Integral as _Integral

    for i, integ in enumerate(form.integrals):
        k = integ.kernel
        kb = getattr(k, "builtin", None)
        if k.form != 100 or kb is None or kb.celltype not in (1, 2) or os.environ.get("MPCX_UFCX_BUILTIN", "1") == "0":
            continue
        import warnings

        ok = False
        try:
            ok = _twin_agrees(form, integ, kb, _Form, _Integral)
        except Exception as e:  # noqa: BLE001  (a twin the library cannot evaluate is no twin -- but say so: ADVICE r4)
            warnings.warn(f"dolfinx_mpc_amd: the stated built-in twin of imported kernel '{k.ufcx_name}' could not be checked ({e}); "
                          "the imported text runs", RuntimeWarning, stacklevel=3)
            ok = False
        if ok is None:
            # both kernels gave zero on the sample (a constant that is still 0, ...): undecided.  Looked at again on the next
            # calls, a bounded number of times -- every retry costs two sample assemblies
            form._twin_retries = getattr(form, "_twin_retries", 0) + 1
            if form._twin_retries < 4:
                form._twins_resolved = False
            continue
        if ok:
            integ.kernel_imported = k
            integ.kernel = kb
            swapped = True
        else:
            warnings.warn(f"dolfinx_mpc_amd: imported kernel '{k.ufcx_name}' does not agree with its stated built-in twin on the "
                          "sample entities; the imported text runs (the fast built-in kernels are not used)", RuntimeWarning, stacklevel=3)
            k.builtin = None
    if swapped:
        form._device.clear()  # argument blocks built for the text are stale


def _twin_agrees(form, integ, kb, _Form, _Integral) -> bool:
    import importlib

    import torch

    from .multipointconstraint import MultiPointConstraint

    n = integ.num_entities
    if n == 0:
        return True
    pick = np.unique(np.linspace(0, n - 1, min(n, 257)).astype(np.int64))
    ents = np.ascontiguousarray(integ.entities[pick])
    # coefficients: NOT the live values (zero at the usual Newton initial guess; a text that agrees with its twin only for
    # special data would slip through) -- reproducible pseudo-random packed values in [0.5, 1.5]
    coeff = integ.coefficient
    if coeff is not None:
        coeff = 0.5 + np.random.default_rng(12345).random((pick.size, integ.cstride))
    spaces = form.function_spaces
    f_txt = _Form(spaces, [_Integral(integ.itype, ents, integ.kernel, coeff, integ.constant)])
    f_txt._twins_resolved = True
    f_blt = _Form(spaces, [_Integral(integ.itype, ents, kb, coeff, integ.constant)])
    f_blt._twins_resolved = True
    empties = []
    for V in spaces:
        m = getattr(V, "_empty_mpc", None)
        if m is None:
            m = V._empty_mpc = MultiPointConstraint(V)
            m.finalize()
        empties.append(m)
    if form.rank == 1:
        av = importlib.import_module(__package__ + ".assemble_vector")
        a = av.assemble_vector(f_txt, empties[0], algorithm="atomic").array
        b = av.assemble_vector(f_blt, empties[0], algorithm="atomic").array
    else:
        am = importlib.import_module(__package__ + ".assemble_matrix")
        A = am.create_matrix(f_txt, empties[0], empties[1])
        a = am.assemble_matrix(f_txt, (empties[0], empties[1]), A=A, algorithm="atomic").vals.clone()
        b = am.assemble_matrix(f_blt, (empties[0], empties[1]), A=A, algorithm="atomic").vals
    scale = float(torch.maximum(a.abs().max(), b.abs().max()).item())
    if scale == 0.0:
        return None  # nothing to compare (e.g. constants that are zero right now): undecided
    return bool(float((a - b).abs().max().item()) <= 1e-12 * scale)


def integral_device(form: Form, i: int):
    """entities / quadrature tables of integral i (structural: uploaded once) and its packed
    coefficients / constants, which are VALUES: refreshed whenever the coefficient's dof array was
    handed out again or the constants differ from the uploaded copy (the reference packs both on every
    call, cpp/assemble_matrix.cpp:583-589)."""
    dev = _native.require_gpu()
    key = (str(dev), i)
    integ: Integral = form.integrals[i]
    if key not in form._device:
        k = integ.kernel
        # basis of a P2 source form at the rule's points (mpcx_kernel_t::qphi)
        qphi = None
        if k.form == 2 and k.degree == 2 and k.qwts.size > 0:
            from .quadrature import lagrange_basis

            qphi = _to_dev(lagrange_basis(form.mesh.cell_name, 2, k.qpts).reshape(-1), dev)
        # source forms with an integrand function that is affine in x (constant / linear) on affine simplices: the rule's
        # moments against the barycentric coordinates (mpcx_kernel_t::vphi) -- the kernels then evaluate f at the vertices
        # instead of walking the rule (the same sum up to rounding); MPCX_VERTEX_SOURCE=0 keeps the rule
        vphi = None
        if (k.form == 2 and integ.itype == "cell" and k.celltype in (1, 2) and k.degree in (1, 2) and k.coeff_degree == 0
                and k.qwts.size > (3 if k.celltype == 1 else 4) and os.environ.get("MPCX_VERTEX_SOURCE", "1") != "0"):
            # (only rules of more points than the cell has vertices: a one-point rule is cheaper walked -- contact b of config 4
            # 0.28 ms walked, 0.38 ms from the moments)
            from .fem import _FN_DEGREE
            from .quadrature import lagrange_basis

            if _FN_DEGREE.get(k.fn_id, 99) <= 1:
                cell = form.mesh.cell_name
                phi = lagrange_basis(cell, k.degree, k.qpts)  # (nq, nd)
                lam = lagrange_basis(cell, 1, k.qpts)  # (nq, nv)
                vphi = _to_dev(np.ascontiguousarray(np.einsum("q,qv,qi->vi", k.qwts.astype(np.float64), lam, phi)).reshape(-1), dev)
        # cell integral over cells 0..n-1 in order: the kernels skip the indirection (and nothing is uploaded)
        n = integ.entities.shape[0]
        ident = integ.itype == "cell" and n > 0 and int(integ.entities[0]) == 0 and int(integ.entities[-1]) == n - 1 and \
            (integ.entities is getattr(form.mesh, "_all_cells", None)
             or bool(np.array_equal(integ.entities, np.arange(n, dtype=integ.entities.dtype))))
        d = {
            "entities": None if ident else _to_dev(integ.entities.astype(np.int32).reshape(-1), dev),
            "coeffs": None,
            "coeff_host": None,  # per coefficient: (copy of the dof values the device pack was made from, device dofs)
            "constants": None,
            "constants_host": None,
            "qpts": _to_dev(k.qpts.astype(np.float64).reshape(-1), dev),
            "qwts": _to_dev(k.qwts.astype(np.float64), dev),
            "fqpts": _to_dev(k.fqpts.astype(np.float64).reshape(-1), dev),
            "fqwts": _to_dev(k.fqwts.astype(np.float64), dev),
            "qphi": qphi,
            "vphi": vphi,
        }
        d["entities_ptr"] = None if ident else d["entities"].data_ptr()
        d["kernel"] = _native.KernelT(
            k.form, k.celltype, k.degree, k.bs, k.degree1 or k.degree, k.bs1 or k.bs, k.fn_id, k.coeff_degree,
            int(k.qwts.size), int(k.fqwts.size),
            d["qpts"].data_ptr(), d["qwts"].data_ptr(), d["fqpts"].data_ptr(), d["fqwts"].data_ptr(),
            ufcx_compile(k, form) if k.form == 100 else None, None if qphi is None else qphi.data_ptr(),
            _native.scalar_id(getattr(form, "dtype", np.float64)), None if vphi is None else vphi.data_ptr(),
        )
        kb = getattr(k, "builtin", None)
        d["kernel_builtin"] = None
        if kb is not None:
            d["qpts_b"], d["qwts_b"] = _to_dev(kb.qpts.astype(np.float64).reshape(-1), dev), _to_dev(kb.qwts.astype(np.float64), dev)
            d["kernel_builtin"] = _native.KernelT(
                kb.form, kb.celltype, kb.degree, kb.bs, kb.degree1 or kb.degree, kb.bs1 or kb.bs, kb.fn_id, kb.coeff_degree,
                int(kb.qwts.size), 0, d["qpts_b"].data_ptr(), d["qwts_b"].data_ptr(), d["fqpts"].data_ptr(), d["fqwts"].data_ptr(),
                None, None, 0, None)
        form._device[key] = d
    d = form._device[key]
    if integ.coefficient is not None:
        _refresh_coefficients(form, integ, d, dev)
    c = integ.constants
    if c is not None:
        c = c.astype(getattr(form, "dtype", np.float64), copy=False)  # (the form's scalar type: the kernels read T)
    if c is not None and (d["constants_host"] is None or not np.array_equal(c, d["constants_host"])):
        d["constants_host"] = c.copy()
        d["constants"] = _update_dev(d.get("constants"), d["constants_host"], dev)
    return d


_cmp_pool = None


def same_values(a: np.ndarray, b: np.ndarray) -> bool:
    """bitwise equality of two C-contiguous arrays of one dtype and shape; large arrays are compared with memcmp in
    a few threads (ctypes releases the GIL): the check that stands in for the reference's re-packing of every
    coefficient on every call has to cost less than the assembly it guards (17 M dofs: ~1 ms instead of 4-30)"""
    global _cmp_pool
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    if a.nbytes < (4 << 20) or not (a.flags.c_contiguous and b.flags.c_contiguous):
        return bool(np.array_equal(a, b, equal_nan=True))
    import ctypes
    from concurrent.futures import ThreadPoolExecutor

    if _cmp_pool is None:
        import os

        _cmp_pool = (ThreadPoolExecutor(min(16, os.cpu_count() or 1)), ctypes.CDLL(None).memcmp)
        _cmp_pool[1].argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        _cmp_pool[1].restype = ctypes.c_int
    pool, memcmp = _cmp_pool
    n, pa, pb = a.nbytes, a.ctypes.data, b.ctypes.data
    step = ((n // 32) + 4095) // 4096 * 4096
    return all(r == 0 for r in pool.map(lambda o: memcmp(pa + o, pb + o, min(step, n - o)), range(0, n, step)))


def _refresh_coefficients(form: Form, integ: Integral, d: dict, dev):
    """Packed coefficients of one integral on the device, float64 [n_entities][cstride] in the layout of dolfinx
    ``pack_coefficients``.  The reference packs on every call (cpp/assemble_matrix.cpp:587-589); here the CURRENT
    dof values of every coefficient are compared with the copy the device pack was made from (a caller may keep
    ``f.x.array`` and write through it at any time, so nothing short of looking at the values is reliable) and
    only a changed coefficient is uploaded again; the gather through the dofmap runs on the device (torch:
    plumbing)."""
    import torch

    T = np.dtype(getattr(form, "dtype", np.float64))  # the form's scalar type: the kernels read packed values of T
    if isinstance(integ.coefficient, np.ndarray):  # packed by the caller
        if d["coeff_host"] is None or not same_values(integ.coefficient, d["coeff_host"]):
            d["coeff_host"] = integ.coefficient.copy()
            d["coeffs"] = _update_dev(d["coeffs"], integ.coefficient.astype(T, copy=False), dev)
        return
    fs = integ.coefficient_functions
    if d["coeff_host"] is None:
        d["coeff_host"] = [[None, None] for _ in fs]
    dirty = False
    for slot, f in zip(d["coeff_host"], fs):
        cur = f.x._data
        if slot[0] is None or not same_values(cur, slot[0]):
            slot[0] = cur.copy()
            slot[1] = _update_dev(slot[1], slot[0].astype(T, copy=False), dev)
            dirty = True
    if not dirty and d["coeffs"] is not None:
        return
    n = integ.num_entities
    cells = None if d["entities"] is None else d["entities"].view(n, integ.estride)[:, 0].long()
    parts = []
    for slot, f in zip(d["coeff_host"], fs):
        Vc = f.function_space
        dm = space_device(Vc)["dofmap"].view(-1, Vc.element_ndofs)
        dofs = (dm[:n] if cells is None else dm[cells]).long()
        bs = Vc.dofmap.bs
        if bs > 1:
            dofs = (dofs[:, :, None] * bs + torch.arange(bs, device=dev)[None, None, :]).reshape(n, -1)
        parts.append(slot[1][dofs])
    new = (parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)).contiguous()
    if d["coeffs"] is not None and d["coeffs"].shape == new.shape:
        d["coeffs"].copy_(new)  # in place: argument blocks / captured graphs keep reading the same tensor
    else:
        d["coeffs"] = new


def bc_markers(V: FunctionSpace, bcs, cache: dict):
    """int8 marker over unrolled dofs for the bcs living in V
    (cpp/assemble_matrix.cpp:688-705); None if no bc applies."""
    mine = [bc for bc in bcs if V.contains(bc.function_space)]
    if not mine:
        return None, None
    dev = _native.require_gpu()

    def build():
        m = np.zeros(V.num_dofs, dtype=np.int8)
        for bc in mine:
            bc.mark_dofs(m)
        return (m, _to_dev(m, dev))

    return cached(cache, "bcm", [V] + mine, str(dev), build)


def bc_values(V: FunctionSpace, bcs, cache: dict, dtype=np.float64):
    """(host markers, device markers, device values) over the unrolled dofs of V for lifting
    (cpp/lifting.h:166-180).  The markers and dof lists are structural and cached per (space, bcs);
    the VALUES are read from the live bc objects on every call -- a time-dependent boundary function
    or a constant changed between solves must reach the kernel, as set_bc re-reads them too -- and
    re-uploaded only when they differ from what the device holds."""
    import torch

    dev = _native.require_gpu()
    bcs = list(bcs)

    def build():
        markers = np.zeros(V.num_dofs, dtype=np.int8)
        for bc in bcs:
            bc.dof_indices()  # settles the owned-first order of bc._dofs
            bc.mark_dofs(markers)
        return {"markers": markers, "d_markers": _to_dev(markers, dev),
                "d_values": torch.from_numpy(np.zeros(V.num_dofs, dtype=dtype)).to(dev),
                "d_dofs": [_to_dev(bc._dofs.astype(np.int64), dev) for bc in bcs], "g": [None] * len(bcs)}

    s = cached(cache, "lift", [V] + bcs, (str(dev), np.dtype(dtype).name), build)
    for k, bc in enumerate(bcs):
        g = bc.values_at_dofs()
        if s["g"][k] is None or not np.array_equal(g, s["g"][k]):
            # later bcs override earlier ones on shared dofs: refresh this and every later bc in order
            for k2 in range(k, len(bcs)):
                g2 = bcs[k2].values_at_dofs()
                s["g"][k2] = g2.copy()
                s["d_values"][s["d_dofs"][k2]] = _to_dev(np.asarray(g2).astype(dtype, copy=False), dev)
            break
    return s["markers"], s["d_markers"], s["d_values"]


def ptr(t):
    return None if t is None else t.data_ptr()


def stream_ptr():
    """raw handle of torch's current stream on the current device (the fast accessor: this is read ~15 times per step)"""
    import torch

    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def cell_info_ptr(V, kernel):
    """device pointer of the cell permutation words of V's mesh when ``kernel`` (an imported one) carries dof transformations
    (include/mpcx.h cell_info0 / cell_info1), else None; uploaded once per mesh"""
    if getattr(kernel, "ufcx_transforms", None) is None:
        return None
    mesh = V.mesh
    info = getattr(mesh, "cell_permutation_info", None)
    if info is None:
        raise ValueError("imported kernel with dof transformations: mesh.cell_permutation_info is not set")
    dev = _native.require_gpu()
    key = ("cell_info", str(dev))
    hit = mesh._device.get(key)
    if hit is None or hit[0] is not info:
        hit = mesh._device[key] = (info, _to_dev(np.ascontiguousarray(info, dtype=np.uint32).view(np.int32), dev))
    return hit[1].data_ptr()
