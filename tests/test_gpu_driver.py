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


@pytest.mark.gpu
def test_driver_on_a_mesh_without_clusters(oracle, tmp_path):
    """a Delaunay tetrahedral mesh with a non-matching periodic pair (multi-master slaves, fat rows, no six-tet fans): the
    driver takes the per-cell LDS row-block plan (mpcx_cell_plan_create) and the owner-computes vector plan over the cells"""
    from problems import case_delaunay_periodic

    case = case_delaunay_periodic(3, 1, 6, seed=3, bc_value=0.4)
    pin, pout = str(tmp_path / "problem.bin"), str(tmp_path / "result.bin")
    problem_file(case, pin)
    run = subprocess.run([DRIVER, pin, pout], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    res = read_bundle(pout)
    # (a random mesh may hold a chance six-tet fan or two: those go to the cluster kernel, the rest are per-cell cells)
    assert res["timings"][5] >= 0.9 * case.V.mesh.num_cells and 6 * res["timings"][4] + res["timings"][5] == case.V.mesh.num_cells
    ref = oracle_outputs(oracle, case)
    refA = ref["A"].tocsr()
    refA.sort_indices()
    assert np.array_equal(res["rowptr"], refA.indptr) and np.array_equal(res["cols"], refA.indices)
    assert abs(res["vals"] - refA.data).max() <= 1e-12 * abs(refA.data).max()
    b_ref = ref["b_lifted"].copy()
    for bc in case.bcs:
        bc.set(b_ref, None, 1.0)
    assert abs(res["b"] - b_ref).max() <= 1e-12 * max(1.0, abs(b_ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("make", ["cube_p2", "delaunay_p1", "stokes_a01"])
def test_cell_plan_from_the_c_abi_equals_the_torch_built_plan(make):
    """mpcx_cell_plan_create (row ranges, entity lists grouped by local rows, scatter offsets, masked dofmaps in library-owned
    memory) against assemble_matrix._rowblock_plan + _masked_dofmap, array by array"""
    import ctypes as C
    import importlib

    import torch

    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import _device as D
    from dolfinx_mpc_amd import _native
    from problems import case_delaunay_periodic, product_mpc
    from test_gpu_cluster_plan import _dev_array

    am = importlib.import_module("dolfinx_mpc_amd.assemble_matrix")
    if make == "stokes_a01":
        from dolfinx_mpc_amd.workloads import stokes_slip_problem

        V, Q, bcs, raw_v, forms, _L0 = stokes_slip_problem(3, 3, None)
        mv = dm.MultiPointConstraint(V)
        mv.add_constraint(V, *raw_v)
        mv.finalize()
        mq = dm.MultiPointConstraint(Q)
        mq.finalize()
        form, mpc0, mpc1 = forms[(0, 1)], mv, mq
    else:
        case = case_cube_periodic(6, 2, 0.3, reorder=(2, 2, 2)) if make == "cube_p2" else case_delaunay_periodic(3, 1, 5, seed=7)
        form, bcs = case.a, case.bcs
        mpc0 = mpc1 = product_mpc(case)
    V0, V1 = form.function_spaces
    A = dm.create_matrix(form, mpc0, mpc1)
    plan_t, t, info = am._rowblock_plan(A, form, 0, V0)
    group_rows = 1  # (what _rowblock_plan takes for these forms)
    _, bc0 = D.bc_markers(V0, bcs, form._device)
    _, bc1 = D.bc_markers(V1, bcs, form._device)
    md0 = am._masked_dofmap(form, V0, bc0, mpc0, 0)
    md1 = am._masked_dofmap(form, V1, bc1, mpc1, 1)
    s0, s1 = D.space_device(V0), D.space_device(V1)
    _, k0 = mpc0._device()
    _, k1 = mpc1._device()
    L = _native.lib()
    hints = None if V0.dof_tile_offsets is None else np.ascontiguousarray(V0.dof_tile_offsets.astype(np.int32) * V0.dofmap.bs)
    rowptr_h = np.ascontiguousarray(A.rowptr.astype(np.int64))
    nc = V0.dofmap.list.shape[0]
    h = C.c_void_p()
    rc = L.mpcx_cell_plan_create(A.shape[0], A.d_rowptr.data_ptr(), rowptr_h.ctypes.data, A.d_cols.data_ptr(), nc, 1, None, nc,
                                 s0["dofmap"].data_ptr(), V0.element_ndofs, V0.dofmap.bs, D.ptr(bc0), k0["is_slave"].data_ptr(),
                                 s1["dofmap"].data_ptr(), V1.element_ndofs, V1.dofmap.bs, D.ptr(bc1), k1["is_slave"].data_ptr(),
                                 am.ROWBLOCK_MAX_ROWS, am.ROWBLOCK_MAX_NNZ, None if hints is None else hints.ctypes.data,
                                 0 if hints is None else hints.size, group_rows, D.stream_ptr(), C.byref(h))
    _native.check(rc, "mpcx_cell_plan_create")
    torch.cuda.synchronize()
    try:
        a = _native.MatrixArgs()
        _native.check(L.mpcx_cell_plan_fill(h, C.byref(a)), "mpcx_cell_plan_fill")
        assert (a.plan.num_blocks, a.plan.max_rows, a.plan.max_nnz, a.plan.row_pairs) == (plan_t.num_blocks, plan_t.max_rows, plan_t.max_nnz, 0)
        assert a.algorithm == 2 and a.lean == 0
        nb, nslots = plan_t.num_blocks, int(t[2].numel())
        assert L.mpcx_cell_plan_num_slots(h) == nslots and L.mpcx_cell_plan_num_blocks(h) == nb
        assert np.array_equal(_dev_array(a.plan.block_row0, nb + 1, np.int32), t[0].cpu().numpy())
        assert np.array_equal(_dev_array(a.plan.block_ent_off, nb + 1, np.int64), t[1].cpu().numpy())
        assert np.array_equal(_dev_array(a.plan.block_ents, nslots, np.int32), t[2].cpu().numpy())
        assert np.array_equal(_dev_array(a.plan.ent_offs, t[3].numel(), np.uint8), t[3].cpu().numpy())
        assert np.array_equal(_dev_array(a.mdofmap0, md0.numel(), np.int32), md0.cpu().numpy().reshape(-1))
        assert np.array_equal(_dev_array(a.mdofmap1, md1.numel(), np.int32), md1.cpu().numpy().reshape(-1))
    finally:
        L.mpcx_cell_plan_destroy(h)


def test_driver_binary_links_only_the_library_and_hip():
    """built on CPU by build(); its dynamic dependencies are libmpcx.so, the HIP runtime and the C / C++ runtimes"""
    assert os.path.exists(DRIVER)
    out = subprocess.run(["ldd", DRIVER], capture_output=True, text=True).stdout
    libs = [ln.split()[0] for ln in out.splitlines() if "=>" in ln or ln.strip().startswith("/")]
    assert any(name.startswith("libmpcx.so") for name in libs) and any(name.startswith("libamdhip64") for name in libs)
    assert not any("torch" in name or "python" in name for name in libs), libs


# ---------------------------------------------------------------------------------------------------------
# Round 6: ``dolfinx_mpc_amd/mpcx_driver_blocks`` (examples/mpcx_driver_blocks.cpp) -- the other workloads behind the C ABI
# alone: pair-record, node-block, per-cell and master-contribution plans each from ONE library call (mpcx_pairs_plan_create,
# mpcx_nodeblock_plan_create, mpcx_cell_plan_create, mpcx_master_plan_create), blocks over one or two spaces.
# ---------------------------------------------------------------------------------------------------------
DRIVER_BLOCKS = os.path.join(ROOT, "dolfinx_mpc_amd", "mpcx_driver_blocks")


def _space_arrays(k, V, raw, bcs):
    markers = np.zeros(V.num_dofs, dtype=np.int8)
    values = np.zeros(V.num_dofs, dtype=np.float64)
    for bc in bcs:
        if V.contains(bc.function_space):
            bc.mark_dofs(markers)
            bc.set(values, None, 1.0)
    sl, ms, co, ow, off = raw
    p = f"s{k}"
    return {p + "_shape": np.array([V.element_ndofs, V.dofmap.bs, V.num_dofs], dtype=np.int32),
            p + "_dofmap": V.dofmap.list.astype(np.int32).reshape(-1), p + "_slaves": np.asarray(sl, np.int32),
            p + "_masters": np.asarray(ms, np.int64), p + "_coeffs": np.asarray(co, np.float64), p + "_owners": np.asarray(ow, np.int32),
            p + "_offsets": np.asarray(off, np.int32), p + "_bc_markers": markers, p + "_bc_values": values}


def _vector_arrays(k, space, form, rows):
    from dolfinx_mpc_amd.quadrature import lagrange_basis

    integ = form.integrals[0]
    kk = integ.kernel
    d = {f"v{k}_space": np.array([space, rows], dtype=np.int32)}
    d.update(_kernel_arrays(f"v{k}", integ))
    if kk.degree == 2:
        d[f"v{k}_qphi"] = lagrange_basis(form.mesh.cell_name, 2, kk.qpts).reshape(-1)
    return d


def _run_blocks(tmp_path, arrays, env=None):
    assert os.path.exists(DRIVER_BLOCKS), "mpcx_driver_blocks is built by __graft_entry__.build() (make -C dolfinx_mpc_amd/csrc)"
    pin, pout = str(tmp_path / "problem.bin"), str(tmp_path / "result.bin")
    write_bundle(pin, arrays)
    run = subprocess.run([DRIVER_BLOCKS, pin, pout, "2"], capture_output=True, text=True, timeout=600,
                         env=None if env is None else dict(os.environ, **env))
    assert run.returncode == 0, run.stdout + run.stderr
    return read_bundle(pout), run.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [1, 0], ids=["pair records", "per-cell row blocks"])
def test_blocks_driver_config5_p2_poisson(oracle, tmp_path, kind):
    """BASELINE configs[4] at test size (scalar P2, periodic + Dirichlet walls) without Python: pattern, pair-record plan,
    device master plan, owner-computes vector -- against the oracle (pattern bit for bit, values 1e-12)"""
    case = case_cube_periodic(6, 2, 0.0, reorder=(2, 2, 2))
    V = case.V
    arrays = {"x": V.mesh.geometry.x.reshape(-1), "cells": V.mesh.geometry.dofmap.astype(np.int32).reshape(-1)}
    arrays.update(_space_arrays(0, V, case.raw, case.bcs))
    arrays.update({"b0_spaces": np.array([0, 0, kind], dtype=np.int32), "b0_params": np.array([96, 4608], dtype=np.int32)})
    arrays.update(_kernel_arrays("b0", case.a.integrals[0]))
    arrays.update(_vector_arrays(0, 0, case.L, 1024))
    res, log = _run_blocks(tmp_path, arrays)
    ref = oracle_outputs(oracle, case)
    refA = ref["A"].tocsr()
    refA.sort_indices()
    assert np.array_equal(res["A0_rowptr"], refA.indptr) and np.array_equal(res["A0_cols"], refA.indices)
    assert abs(res["A0_vals"] - refA.data).max() <= 1e-12 * abs(refA.data).max()
    assert abs(res["b0"] - ref["b"]).max() <= 1e-12 * max(1.0, abs(ref["b"]).max())
    if kind == 1:
        # the right-hand side came from per-interval tables (mpcx_cell_grid_plan_create); point by point it rounds differently
        res2, _log = _run_blocks(tmp_path, arrays, env={"MPCX_DRIVER_NO_GRID": "1"})
        assert abs(res2["b0"] - res["b0"]).max() <= 1e-14 * abs(ref["b"]).max() and not np.array_equal(res2["b0"], res["b0"])


@pytest.mark.gpu
def test_blocks_driver_config3_taylor_hood(oracle, tmp_path):
    """the Taylor-Hood blocks of BASELINE configs[2] at test size without Python: a00 through the node-block plan (CSR values),
    a01 / a10 through pair records (rectangular, two constraints), b0 -- against the oracle's nest assembly"""
    from problems import empty_raw, stokes_slip_problem

    po = oracle
    V, Q, bcs, raw_v, forms, L0 = stokes_slip_problem(3, 3)
    arrays = {"x": V.mesh.geometry.x.reshape(-1), "cells": V.mesh.geometry.dofmap.astype(np.int32).reshape(-1)}
    arrays.update(_space_arrays(0, V, raw_v, bcs))
    arrays.update(_space_arrays(1, Q, empty_raw(), bcs))
    for k, ((i, j), kind, prm) in enumerate((((0, 0), 2, (96, 4608)), ((0, 1), 1, (96, 2304)), ((1, 0), 1, (96, 2304)))):
        arrays[f"b{k}_spaces"] = np.array([i, j, kind], dtype=np.int32)
        arrays[f"b{k}_params"] = np.array(prm, dtype=np.int32)
        arrays.update(_kernel_arrays(f"b{k}", forms[(i, j)].integrals[0]))
    arrays.update(_vector_arrays(0, 0, L0, 1536))
    res, log = _run_blocks(tmp_path, arrays)
    mv = po.OracleMPC.from_raw(V, *raw_v)
    mq = po.OracleMPC.from_raw(Q, *empty_raw())
    mpcs = [mv, mq]
    for k, (i, j) in enumerate(((0, 0), (0, 1), (1, 0))):
        ref = po.assemble_matrix(forms[(i, j)], mpcs[i], mpcs[j], bcs=bcs).tocsr()
        ref.sort_indices()
        assert np.array_equal(res[f"A{k}_rowptr"], ref.indptr) and np.array_equal(res[f"A{k}_cols"], ref.indices), (i, j)
        assert abs(res[f"A{k}_vals"] - ref.data).max() <= 1e-12 * max(1.0, abs(ref.data).max()), (i, j)
    b0 = po.assemble_vector(L0, mv)
    assert abs(res["b0"] - b0).max() <= 1e-12 * max(1.0, abs(b0).max())
