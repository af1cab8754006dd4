"""Device mirrors of the flat host arrays (torch is the allocator / stream
provider; the kernels only ever see raw pointers through the C ABI)."""

from __future__ import annotations

import numpy as np

from . import _native
from .fem import Form, FunctionSpace, Integral


def _to_dev(a: np.ndarray, dev):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def mesh_device(mesh):
    dev = _native.require_gpu()
    key = str(dev)
    if key not in mesh._device:
        mesh._device[key] = {
            "x": _to_dev(mesh.geometry.x, dev),
            "x_dofmap": _to_dev(mesh.geometry.dofmap, dev),
        }
    return mesh._device[key]


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
    return V._device[key]


def integral_device(form: Form, i: int):
    """entities / coefficients / constants / quadrature tables of integral i."""
    dev = _native.require_gpu()
    key = (str(dev), i)
    if key not in form._device:
        integ: Integral = form.integrals[i]
        k = integ.kernel
        d = {
            "entities": _to_dev(integ.entities.astype(np.int32).reshape(-1), dev),
            "coeffs": None if integ.coeffs is None else _to_dev(integ.coeffs.astype(np.float64), dev),
            "constants": None if integ.constants is None else _to_dev(integ.constants.astype(np.float64), dev),
            "qpts": _to_dev(k.qpts.astype(np.float64).reshape(-1), dev),
            "qwts": _to_dev(k.qwts.astype(np.float64), dev),
            "fqpts": _to_dev(k.fqpts.astype(np.float64).reshape(-1), dev),
            "fqwts": _to_dev(k.fqwts.astype(np.float64), dev),
        }
        # cell integral over cells 0..n-1 in order: the kernels skip the indirection
        ident = integ.itype == "cell" and integ.entities.size > 0 and int(integ.entities[0]) == 0 and \
            int(integ.entities[-1]) == integ.entities.size - 1 and bool(np.all(np.diff(integ.entities) == 1))
        d["entities_ptr"] = None if ident else d["entities"].data_ptr()
        d["kernel"] = _native.KernelT(
            k.form, k.celltype, k.degree, k.bs, k.degree1 or k.degree, k.bs1 or k.bs, k.fn_id, k.coeff_degree,
            int(k.qwts.size), int(k.fqwts.size),
            d["qpts"].data_ptr(), d["qwts"].data_ptr(), d["fqpts"].data_ptr(), d["fqwts"].data_ptr(),
        )
        form._device[key] = d
    return form._device[key]


def bc_markers(V: FunctionSpace, bcs, cache: dict):
    """int8 marker over unrolled dofs for the bcs living in V
    (cpp/assemble_matrix.cpp:688-705); None if no bc applies."""
    mine = [bc for bc in bcs if V.contains(bc.function_space)]
    if not mine:
        return None, None
    dev = _native.require_gpu()
    key = ("bcm", str(dev), id(V), tuple(id(bc) for bc in mine))
    if key not in cache:
        m = np.zeros(V.num_dofs, dtype=np.int8)
        for bc in mine:
            bc.mark_dofs(m)
        cache[key] = (m, _to_dev(m, dev))
    return cache[key]


def ptr(t):
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch

    return torch.cuda.current_stream().cuda_stream
