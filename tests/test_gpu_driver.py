"""``dolfinx_mpc_amd/mpcx_driver`` (examples/mpcx_driver.cpp): the constrained assembly through the C ABI ALONE -- a host
program that links libmpcx.so and the HIP runtime, no Python, no torch -- against the oracle and against the Python host
layer on the same problems: the drop-in boundary exercised from the language a dolfinx_mpc binding would be
written in.  The driver gets a problem file (mesh, add_constraint arrays, Dirichlet data, kernel descriptors), builds the
constraint, the sparsity pattern and the plans itself and writes A and b.

Tolerance: 1e-12 of the largest entry against the oracle (different summation order), the pattern bit for bit."""

import os
import struct
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

from problems import case_cube_periodic, oracle_outputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "dolfinx_mpc_amd", "mpcx_driver")
_DT = {np.dtype(np.int8): 0, np.dtype(np.int32): 1, np.dtype(np.int64): 2, np.dtype(np.float64): 3}
_NP = {0: np.int8, 1: np.int32, 2: np.int64, 3: np.float64}


def write_bundle(path, arrays):
    with open(path, "wb") as f:
        f.write(b"MPCX1\0\0\0" + struct.pack("<q", len(arrays)))
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            f.write(name.encode().ljust(32, b"\0") + struct.pack("<iq", _DT[a.dtype], a.size) + a.tobytes())


def read_bundle(path):
    out = {}
    with open(path, "rb") as f:
        assert f.read(8) == b"MPCX1\0\0\0"
        (count,) = struct.unpack("<q", f.read(8))
        for _ in range(count):
            name = f.read(32).split(b"\0")[0].decode()
            dt, n = struct.unpack("<iq", f.read(12))
            out[name] = np.frombuffer(f.read(n * np.dtype(_NP[dt]).itemsize), dtype=_NP[dt]).copy()
    return out


def _kernel_arrays(prefix, integ):
    k = integ.kernel
    nq = int(k.qwts.size)
    d = {f"{prefix}_kernel": np.array([k.form, k.celltype, k.degree, k.bs, getattr(k, "degree1", k.degree) or k.degree,
                                        getattr(k, "bs1", k.bs) or k.bs, k.fn_id, k.coeff_degree, nq], dtype=np.int32),
         f"{prefix}_qpts": np.asarray(k.qpts, dtype=np.float64).reshape(-1), f"{prefix}_qwts": np.asarray(k.qwts, dtype=np.float64)}
    c = integ.constants
    if c is not None:
        d[f"{prefix}_constants"] = np.asarray(c, dtype=np.float64)
    return d


def problem_file(case, path):
    V = case.V
    mesh = V.mesh
    markers = np.zeros(V.num_dofs, dtype=np.int8)
    values = np.zeros(V.num_dofs, dtype=np.float64)
    for bc in case.bcs:
        bc.mark_dofs(markers)
        bc.set(values, None, 1.0)
    sl, ms, co, ow, off = case.raw
    arrays = {"x": mesh.geometry.x.reshape(-1), "cells": mesh.geometry.dofmap.astype(np.int32).reshape(-1),
              "slaves": np.asarray(sl, np.int32), "masters": np.asarray(ms, np.int64), "coeffs": np.asarray(co, np.float64),
              "owners": np.asarray(ow, np.int32), "offsets": np.asarray(off, np.int32), "bc_markers": markers, "bc_values": values,
              "params": np.array([512, 9216, 2048], dtype=np.int32)}
    if V.dof_tile_offsets is not None:
        arrays["hints"] = V.dof_tile_offsets.astype(np.int32)
    arrays.update(_kernel_arrays("mat", case.a.integrals[0]))
    arrays.update(_kernel_arrays("vec", case.L.integrals[0]))
    write_bundle(path, arrays)


@pytest.mark.gpu
@pytest.mark.parametrize("kwargs,expect_left", [(dict(reorder=(4, 4, 4)), False), (dict(reorder=(4, 4, 4), warp="half"), False),
                                                (dict(numbering="shuffled"), False)], ids=["tiled", "half-warped", "shuffled"])
def test_driver_matches_oracle_and_python_layer(oracle, tmp_path, kwargs, expect_left):
    import dolfinx_mpc_amd as dm
    from problems import product_mpc

    assert os.path.exists(DRIVER), "mpcx_driver is built by __graft_entry__.build() (make -C dolfinx_mpc_amd/csrc)"
    case = case_cube_periodic(12, 1, 0.3, **kwargs)
    pin, pout = str(tmp_path / "problem.bin"), str(tmp_path / "result.bin")
    problem_file(case, pin)
    run = subprocess.run([DRIVER, pin, pout, "2"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "clusters" in run.stdout
    res = read_bundle(pout)
    n = case.V.num_dofs
    A = sp.csr_matrix((res["vals"], res["cols"], res["rowptr"]), shape=(n, n))
    ref = oracle_outputs(oracle, case)
    refA = ref["A"].tocsr()
    refA.sort_indices()
    assert np.array_equal(res["rowptr"], refA.indptr) and np.array_equal(res["cols"], refA.indices)  # the MPC pattern
    assert abs(res["vals"] - refA.data).max() <= 1e-12 * abs(refA.data).max()
    b_ref = ref["b_lifted"].copy()
    for bc in case.bcs:
        bc.set(b_ref, None, 1.0)
    assert abs(res["b"] - b_ref).max() <= 1e-12 * max(1.0, abs(b_ref).max())
    assert res["timings"][4] > 0  # clusters were found: the cluster kernels ran
    # ... and the Python host layer on the same problem
    mpc = product_mpc(case)
    Ap = dm.assemble_matrix(case.a, mpc, bcs=case.bcs)
    bp = dm.assemble_vector(case.L, mpc)
    dm.apply_lifting(bp, [case.a], [case.bcs], mpc)
    dm.set_bc(bp, case.bcs)
    assert np.array_equal(Ap.rowptr, res["rowptr"]) and np.array_equal(Ap.cols, res["cols"])
    assert abs(Ap.vals.cpu().numpy() - res["vals"]).max() <= 1e-13 * abs(refA.data).max()
    assert abs(bp.numpy() - res["b"]).max() <= 1e-13 * max(1.0, abs(b_ref).max())


def test_driver_binary_links_only_the_library_and_hip():
    """built on CPU by build(); its dynamic dependencies are libmpcx.so, the HIP runtime and the C / C++ runtimes"""
    assert os.path.exists(DRIVER)
    out = subprocess.run(["ldd", DRIVER], capture_output=True, text=True).stdout
    libs = [ln.split()[0] for ln in out.splitlines() if "=>" in ln or ln.strip().startswith("/")]
    assert any(name.startswith("libmpcx.so") for name in libs) and any(name.startswith("libamdhip64") for name in libs)
    assert not any("torch" in name or "python" in name for name in libs), libs
