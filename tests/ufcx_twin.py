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
from dolfinx_mpc_amd.codegen import twin_form, twin_integral  # noqa: E402,F401  (the re-expression lives in the package: bench.py uses it too)


def twin_case(case, layout=None):
    import copy

    out = copy.copy(case)
    out.a, out.L = twin_form(case.a, layout), twin_form(case.L, layout)
    out.name = case.name + "_ufcx" + ("_ffcx_layout" if layout == "ffcx" else "")
    return out


def num_imported(case) -> int:
    return sum(i.kernel.form == fem.FORM_UFCX for f in (case.a, case.L) if f is not None for i in f.integrals)
