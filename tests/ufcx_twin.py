"""The small parity cases with their cell integrals re-expressed as IMPORTED UFCx kernels (C text written by
dolfinx_mpc_amd/codegen.py in the shape FFCx gives its output: baked tables + quadrature loop), so that the whole parity
suite also runs through the reference's real seam -- a ``tabulate_tensor`` per integral
(cpp/assemble_matrix.cpp:438-439) -- on every kernel variant.  The oracle side keeps the built-in operators: the
comparison pins the generated kernels against them as well."""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from dolfinx_mpc_amd import fem  # noqa: E402
from dolfinx_mpc_amd.codegen import generate  # noqa: E402

FN_C = {i: fem.fn_c_expression(i) for i in range(6)}  # one table for every generated kernel (fem.py)
_KIND = {fem.FORM_STIFFNESS: "stiffness", fem.FORM_MASS: "mass", fem.FORM_SOURCE: "source", fem.FORM_ELASTICITY: "elasticity"}


def twin_integral(integ, spaces):
    """the same integral with an imported kernel, or the integral itself where the generator has no counterpart
    (exterior facets, Taylor-Hood coupling blocks)"""
    k = integ.kernel
    if integ.itype != "cell" or k.form not in _KIND or (k.degree1 or k.degree) != k.degree or (k.bs1 or k.bs) != k.bs:
        return integ
    cell = "tetrahedron" if k.celltype == fem.CELL_TETRAHEDRON else "triangle"
    kind = _KIND[k.form]
    has_c = integ.constant is not None
    if kind == "source" and k.fn_id == fem.FN_CONSTANT_VEC:
        has_c = True
    src, name = generate(kind, cell, k.degree, k.bs, (k.qpts, k.qwts), coefficient_degree=k.coeff_degree,
                         use_constant=has_c and kind != "elasticity", fexpr=FN_C.get(k.fn_id, "1.0"))
    ks = fem.KernelSpec(fem.FORM_UFCX, k.celltype, k.degree, k.bs, ufcx_source=src, ufcx_name=name)
    if len(spaces) > 1:
        ks.degree1, ks.bs1 = spaces[1].degree, spaces[1].dofmap.bs
    return fem.Integral("cell", integ.entities, ks, integ.coefficient, integ.constant)


def twin_form(form):
    if form is None:
        return None
    return fem.Form(form.function_spaces, [twin_integral(i, form.function_spaces) for i in form.integrals])


def twin_case(case):
    import copy

    out = copy.copy(case)
    out.a, out.L = twin_form(case.a), twin_form(case.L)
    out.name = case.name + "_ufcx"
    return out


def num_imported(case) -> int:
    return sum(i.kernel.form == fem.FORM_UFCX for f in (case.a, case.L) if f is not None for i in f.integrals)
