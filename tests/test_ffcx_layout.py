"""Whole FFCx output FILES as the element-kernel seam (VERDICT r5 "missing" 2 / item 3a).

The reference never sees a bare function: DOLFINx hands it ``form->form_integrals[i]->tabulate_tensor_float64`` out of the
objects FFCx writes behind the functions (cpp/assemble_matrix.cpp:438-439, call at :504-506;
python/src/dolfinx_mpc/numba/assemble_matrix.py:282-290).  tests/ufcx/ffcx_layout_*.c reproduce that file layout
(hand-written / written by dolfinx_mpc_amd.codegen.ffcx_file: FFCx is absent from this image): include block with
``<ufcx.h>``, static tables inside the functions, ``ufcx_integral`` objects with ``#ifndef __STDC_NO_COMPLEX__`` members,
the per-form arrays, ``ufcx_form`` objects and alias pointers, SEVERAL forms per file.

  * ``mpcx_ufcx_resolve`` / ``mpcx_ufcx_compile`` accept the whole file and find the function from a function name, an
    integral object, a form object or its alias (CPU: resolution + gfx950 cross-compilation);
  * the oracle compiles the same file with gcc against oracle/include/ufcx.h and reaches the kernel THROUGH the compiled
    objects with ctypes (the reference's way) -- two independent readings of the file that must agree;
  * on the GPU the imported file runs on every imported-kernel path and equals the oracle."""

import ctypes as C
import os

import numpy as np
import pytest

from dolfinx_mpc_amd import _native, fem
from problems import Case, _walls_yz, case_cube_periodic, oracle_outputs, periodic_raw, product_outputs

HERE = os.path.dirname(os.path.abspath(__file__))
A_FN = "tabulate_tensor_integral_2f1c9a7be3d04a5fb0a1c2d3e4f5a6b7c8d9e0f1"
L_FN = "tabulate_tensor_integral_7b02d5c86e914f3a9d8c7b6a5f4e3d2c1b0a9f8e"


def _file(name="ffcx_layout_poisson_p1"):
    return open(os.path.join(HERE, "ufcx", name + ".c")).read()


def _resolve(src, name):
    L = _native.lib()
    out = C.create_string_buffer(512)
    rc = L.mpcx_ufcx_resolve(src.encode(), None if name is None else name.encode(), out, 512)
    return rc, (out.value.decode() if rc == 0 else L.mpcx_last_error().decode())


def test_names_resolve_through_the_objects():
    src = _file()
    assert _resolve(src, "form_poisson_a") == (0, A_FN)  # alias -> form -> form_integrals[0] -> .tabulate_tensor_float64
    assert _resolve(src, "form_poisson_L") == (0, L_FN)
    assert _resolve(src, "form_d41a6c0e8b7f4e2d9c3b5a1f0e6d7c8b9a2f3e4d") == (0, A_FN)  # the ufcx_form object itself
    assert _resolve(src, "integral_7b02d5c86e914f3a9d8c7b6a5f4e3d2c1b0a9f8e") == (0, L_FN)  # a ufcx_integral object
    assert _resolve(src, A_FN) == (0, A_FN)  # a plain function name still works
    rc, msg = _resolve(src, None)
    assert rc != 0 and "2 found" in msg  # two integrals: a name is required
    rc, msg = _resolve(src, "form_poisson_b")
    assert rc != 0 and "form_poisson_b" in msg
    # the bare-function files of earlier rounds: the only function is found without a name
    for f in ("laplace_p1_tet", "source_p1_tet", "slip_facet_p2p1_tet"):
        assert _resolve(_file(f), None) == (0, "tabulate_tensor_" + f)


def test_whole_file_cross_compiles_for_gfx950():
    L = _native.lib()
    src = _file().encode()
    for name, rank in ((b"form_poisson_a", 2), (b"form_poisson_L", 1), (b"integral_2f1c9a7be3d04a5fb0a1c2d3e4f5a6b7c8d9e0f1", 2)):
        d = _native.UfcxDescT(src, name, rank, 4, 1, 4 if rank == 2 else 0, 1 if rank == 2 else 0, 4, None, None)
        h = L.mpcx_ufcx_compile(d)
        assert h, L.mpcx_last_error().decode()
        assert L.mpcx_ufcx_code_size(h) > 1000
        L.mpcx_ufcx_free(h)
    bad = _native.UfcxDescT(src, b"form_poisson_b", 2, 4, 1, 4, 1, 4, None, None)
    assert not L.mpcx_ufcx_compile(bad) and "form_poisson_b" in L.mpcx_last_error().decode()


def _cases(n=4, reorder=None):
    base = case_cube_periodic(n, 1, 0.3, reorder=reorder) if reorder else case_cube_periodic(n, 1, 0.3)
    V = fem.functionspace(base.mesh, ("Lagrange", 1))
    bc = fem.dirichletbc(0.3, fem.locate_dofs_geometrical(V, _walls_yz), V)
    fh = fem.Function(V)
    fh.interpolate(lambda x: 1.0 + 0.5 * x[0] - x[2] * x[1])
    src = _file()
    a = fem.form_ufcx([V, V], src, "form_poisson_a")
    Lf = fem.form_ufcx([V], src, "form_poisson_L", coefficient=fh, constant=fem.Constant(0.7))
    imported = Case("ffcx_layout_poisson_p1", V, a, Lf, [bc], periodic_raw(V, [bc]))
    builtin = Case("builtin_poisson_p1", V, fem.form_stiffness(V), fem.form_source(V, fem.FN_ONE, constant=0.7, coefficient=fh),
                   [bc], periodic_raw(V, [bc]))
    return imported, builtin


def test_oracle_reaches_the_kernels_through_the_compiled_objects():
    from oracle import pyoracle as po

    imported, builtin = _cases()
    got = oracle_outputs(po, imported)
    want = oracle_outputs(po, builtin)
    assert abs(got["A"] - want["A"]).max() <= 1e-13 * abs(want["A"]).max()
    for k in ("b", "b_lifted"):
        assert abs(got[k] - want[k]).max() <= 1e-13 * max(1.0, abs(want[k]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("alg", [None, "rowblock", "atomic"])
@pytest.mark.parametrize("n, reorder", [(4, None), (6, (2, 2, 2))])
def test_gpu_runs_the_file_on_every_imported_kernel_path(oracle, alg, n, reorder):
    imported, _ = _cases(n, reorder)
    ref = oracle_outputs(oracle, imported)
    out = product_outputs(imported, algorithm=alg)
    assert np.array_equal(out["A"].indptr, ref["A"].indptr) and np.array_equal(out["A"].indices, ref["A"].indices)
    assert abs(out["A"].data - ref["A"].data).max() <= 1e-12 * abs(ref["A"].data).max()
    for k in ("b", "b_lifted"):
        assert abs(out[k] - ref[k]).max() <= 1e-12 * max(1.0, abs(ref[k]).max())
