"""mpcx_cluster_plan_*: the cluster set-up of MPCX_ALG_CUBE behind one C-ABI call (device memory owned by the library, no
torch in the build) against the plan dolfinx_mpc_amd/assemble_matrix.py builds through torch -- array by array -- and
the matrix assembled from it against the oracle."""
import ctypes as C
import importlib

import numpy as np
import pytest

from problems import case_cube_periodic, oracle_outputs, product_mpc

pytestmark = pytest.mark.gpu


def _c_plan(case, mpc, A, max_rows, max_nnz):
    import torch

    from dolfinx_mpc_amd import _device as D
    from dolfinx_mpc_amd import _native

    L = _native.lib()
    V = case.V
    md = D.mesh_device(V.mesh)
    _, bc = D.bc_markers(V, case.bcs, case.a._device)
    _, t = mpc._device()
    hints = None if V.dof_tile_offsets is None else np.ascontiguousarray(V.dof_tile_offsets.astype(np.int32))
    rowptr_h = np.ascontiguousarray(A.rowptr.astype(np.int64))
    h = C.c_void_p()
    rc = L.mpcx_cluster_plan_create(V.mesh.num_owned_cells, md["x_dofmap"].data_ptr(), V.mesh.num_nodes, md["x"].data_ptr(), A.shape[0],
                                    A.d_rowptr.data_ptr(), rowptr_h.ctypes.data, A.d_cols.data_ptr(), D.ptr(bc), t["is_slave"].data_ptr(),
                                    max_rows, max_nnz, None if hints is None else hints.ctypes.data, 0 if hints is None else hints.size,
                                    D.stream_ptr(), C.byref(h))
    _native.check(rc, "mpcx_cluster_plan_create")
    torch.cuda.synchronize()
    return h


def _dev_array(ptr, count, dtype):
    import torch

    if count == 0:
        return np.zeros(0, dtype=dtype)
    nbytes = count * np.dtype(dtype).itemsize
    out = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(out.data_ptr(), ptr, nbytes, 3) == 0  # device to device
    return out.cpu().numpy().view(dtype)


@pytest.mark.parametrize("kwargs", [dict(reorder=(4, 4, 4)), dict(reorder=(4, 4, 4), warp="half"), dict(numbering="shuffled")],
                         ids=["tiled", "half-warped", "shuffled"])
def test_cluster_plan_from_the_c_abi_equals_the_torch_built_plan(oracle, kwargs, monkeypatch):
    import torch

    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import _native

    am = importlib.import_module("dolfinx_mpc_amd.assemble_matrix")
    monkeypatch.setattr(am, "CUBE_MAX_ROWS", 64)
    monkeypatch.setattr(am, "CUBE_MAX_NNZ", 64 * 16)
    case = case_cube_periodic(12, 1, 0.3, **kwargs)
    mpc = product_mpc(case)
    A = dm.assemble_matrix(case.a, mpc, bcs=case.bcs, diagval=case.diagval)
    (parts, keep, info), = [v[1] for v in A._plans[("objcache", "cubes")].values()]
    L = _native.lib()
    h = _c_plan(case, mpc, A, 64, 64 * 16)
    try:
        assert L.mpcx_cluster_plan_num_clusters(h) == info["clusters"]
        assert L.mpcx_cluster_plan_num_parts(h) == len(parts)
        verts = _dev_array(L.mpcx_cluster_plan_verts(h), info["clusters"] * 8, np.int32)
        assert np.array_equal(verts, keep[2].cpu().numpy().reshape(-1))
        vals2 = torch.zeros_like(A.vals)
        base, base_keep = am.matrix_args(case.a, 0, A, mpc, mpc, case.bcs, 2, store_mode=1)
        chain, u = [], base
        while u is not None:
            chain.append(u)
            u = u.second
        assert len(chain) == len(parts)
        for p, (part, ref_args) in enumerate(zip(parts, chain)):
            a = _native.MatrixArgs.from_buffer_copy(ref_args)
            _native.check(L.mpcx_cluster_plan_part(h, p, C.byref(a)), "mpcx_cluster_plan_part")
            plan_t, recs, nbytes, ids, off, flags = part[:6]
            assert part[6] is None  # (no slot -> record index: per-slot records, like the C builder's)
            assert (a.cube_rec_bytes, a.cube_flags, a.plan.num_blocks) == (nbytes, flags, plan_t.num_blocks)
            assert (a.plan.max_rows, a.plan.max_nnz) == (plan_t.max_rows, plan_t.max_nnz)
            nslots = recs.numel() // nbytes
            assert np.array_equal(_dev_array(a.cube_recs, nslots * nbytes, np.uint8), recs.cpu().numpy())
            assert np.array_equal(_dev_array(a.plan.block_ent_off, plan_t.num_blocks + 1, np.int64), off.cpu().numpy())
            if ids is not None:
                assert np.array_equal(_dev_array(a.cube_block_ids, plan_t.num_blocks, np.int32), ids.cpu().numpy())
            else:
                assert not a.cube_block_ids
            # ... and the launch from the C-built plan writes the same values
            a.vals = vals2.data_ptr()
            _native.check(L.mpcx_assemble_matrix(C.byref(a)), "mpcx_assemble_matrix")
        cells = C.c_void_p()
        nleft = L.mpcx_cluster_plan_leftover(h, C.byref(cells))
        assert nleft == case.V.mesh.num_owned_cells - 6 * info["clusters"]
        if nleft:  # cells in no cluster (distorted regions): the per-cell kernel adds them, as assemble_matrix does
            assert np.array_equal(_dev_array(cells.value, nleft, np.int32), base.leftover)
            fl = am._leftover_form(case.a, 0, base.leftover)
            al, _kl = am.matrix_args(fl, 0, A, mpc, mpc, case.bcs, 2, 0, with_mpc_kernel=False, allow_cubes=False)
            al.vals = vals2.data_ptr()
            _native.check(L.mpcx_assemble_matrix(C.byref(al)), "mpcx_assemble_matrix")
        else:
            assert base.leftover is None
        torch.cuda.synchronize()
        ref = oracle_outputs(oracle, case)["A"]
        # (bulk + master contributions of the last launch; the diagonals of slave / Dirichlet rows are added by the wrapper)
        got = A.to_scipy().copy()
        got.data = vals2.cpu().numpy()
        diag_rows = np.flatnonzero(abs(ref.diagonal() - got.diagonal()) > 1e-13)
        assert abs((ref - got)).max() <= 1.0 + 1e-12 and set(diag_rows) <= set(np.flatnonzero(ref.diagonal() == case.diagval))
        off_diag = (ref - got).tolil()
        off_diag.setdiag(0.0)
        assert abs(off_diag.tocsr()).max() <= 1e-12 * max(1.0, abs(ref).max())
    finally:
        L.mpcx_cluster_plan_destroy(h)


@pytest.mark.parametrize("kwargs", [dict(reorder=(4, 4, 4)), dict(numbering="shuffled")], ids=["tiled", "shuffled"])
def test_owner_plan_from_the_c_abi_equals_the_torch_allocated_plan(oracle, kwargs):
    """mpcx_owner_plan_create (the owner-computes plan of the cluster vector kernel in library-owned memory) against
    assemble_vector._owner_plan_from_rows, array by array; the vector assembled from it against the oracle"""
    import torch

    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import _device as D
    from dolfinx_mpc_amd import _native
    from dolfinx_mpc_amd.la import create_vector

    av = importlib.import_module("dolfinx_mpc_amd.assemble_vector")
    case = case_cube_periodic(12, 1, 0.3, **kwargs)
    mpc = product_mpc(case)
    b = create_vector(case.V)
    args, keep = av.vector_args(case.L, 0, b, mpc, 0)
    assert args.kernel_name == "cube_own"
    pk = [k for k in keep if isinstance(k, tuple) and len(k) == 9][0]  # the torch-allocated plan: (row0, off, order, lmap, hoff, spill, src, rows, seg)
    V = case.V
    L = _native.lib()
    nc = int(args.n_cubes)
    verts = torch.empty((nc, 8), dtype=torch.int32, device="cuda")
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(verts.data_ptr(), args.cube_verts, nc * 32, 3) == 0
    _, t = mpc._device()
    mrow = torch.empty_like(verts)
    _native.check(L.mpcx_mask_dofmap(verts.data_ptr(), nc, 8, 1, None, t["is_slave"].data_ptr(), 0, mrow.data_ptr(), D.stream_ptr()),
                  "mpcx_mask_dofmap")
    rows = int(np.diff(pk[0].cpu().numpy()).max())  # own rows per block of the reference plan
    hints = None if V.dof_tile_offsets is None else np.ascontiguousarray(V.dof_tile_offsets.astype(np.int32))
    h = C.c_void_p()
    rows_cfg = av._even_rows(V, V._vcube_rows)  # (the cap the plan builder chose for this problem size)
    rc = L.mpcx_owner_plan_create(nc, 8, mrow.data_ptr(), 1, V.num_dofs, rows_cfg, None if hints is None else hints.ctypes.data,
                                  0 if hints is None else hints.size, av.VECTOR_LDS_ROWS, D.stream_ptr(), C.byref(h))
    _native.check(rc, "mpcx_owner_plan_create")
    try:
        a2 = _native.VectorArgs.from_buffer_copy(args)
        _native.check(L.mpcx_owner_plan_fill(h, C.byref(a2)), "mpcx_owner_plan_fill")
        nb = int(a2.plan.num_blocks)
        assert nb == pk[0].numel() - 1 and rows <= rows_cfg and int(a2.n_own_rows) == int(args.n_own_rows)
        assert int(a2.plan.max_rows) == int(args.plan.max_rows)
        for name, ptr, ref in (("row0", a2.plan.block_row0, pk[0]), ("off", a2.plan.block_ent_off, pk[1]), ("order", a2.plan.block_ents, pk[2]),
                               ("lmap", a2.own_lmap, pk[3].reshape(-1)), ("hoff", a2.own_hoff, pk[4]), ("src", a2.own_src, pk[6]),
                               ("rows", a2.own_rows, pk[7]), ("seg", a2.own_seg, pk[8])):
            refh = ref.cpu().numpy()
            n = refh.size if name != "src" else int(pk[4][-1].item())
            got = _dev_array(ptr, n, refh.dtype)
            assert np.array_equal(got, refh[:n]), name
        b2 = create_vector(V)
        a2.b = b2.array.data_ptr()
        _native.check(L.mpcx_assemble_vector(C.byref(a2)), "mpcx_assemble_vector")
        torch.cuda.synchronize()
        ref = dm.assemble_vector(case.L, mpc).numpy()
        assert abs(b2.numpy() - ref).max() <= 1e-12 * max(1.0, abs(ref).max())
    finally:
        L.mpcx_owner_plan_destroy(h)


@pytest.mark.parametrize("shape", ["tiled", "untiled", "stretched", "warped"])
def test_grid_plan_from_the_c_abi_equals_the_torch_built_plan(oracle, shape):
    """mpcx_grid_plan_create (the tensor grid under a mesh of box clusters, in library-owned memory) against
    assemble_vector._cluster_grid / _block_rows, array by array; the vector assembled from it against the oracle; a mesh whose
    clusters are not boxes has no grid (return code 1, no plan)"""
    import torch

    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import _device as D
    from dolfinx_mpc_amd import _native, fem
    from dolfinx_mpc_amd.la import create_vector
    from dolfinx_mpc_amd.mesh import create_unit_cube
    from problems import Case, _walls_yz, periodic_raw

    av = importlib.import_module("dolfinx_mpc_amd.assemble_vector")
    if shape == "warped":
        case = case_cube_periodic(8, 1, 0.0, reorder=(4, 4, 4), warp=True)
    else:
        mesh = create_unit_cube(12, 9, 10, reorder=None if shape == "untiled" else (4, 4, 4))
        if shape == "stretched":
            x = mesh.geometry.x.copy()
            x[:, 0] = x[:, 0] * (1.0 + 0.3 * x[:, 0] * (1.0 - x[:, 0]))  # (not uniform; the faces x = 0, 1 stay)
            x[:, 2] = 1.0 - 0.8 * x[:, 2]  # (mirrored: the clusters' corner 7 lies below corner 0 in z)
            mesh.geometry.x = x
        V = fem.functionspace(mesh, ("Lagrange", 1))
        bc = fem.dirichletbc(0.0, fem.locate_dofs_geometrical(V, _walls_yz), V)
        case = Case("grid_" + shape, V, fem.form_stiffness(V), fem.form_source(V, fem.FN_BENCH_PERIODIC, constant=1.1), [bc],
                    periodic_raw(V, [bc]))
    mpc = product_mpc(case)
    args, keep = av.vector_args(case.L, 0, create_vector(case.V), mpc, 0)
    assert args.kernel_name == "cube_own"
    L = _native.lib()
    h = C.c_void_p()
    rc = L.mpcx_grid_plan_create(args.cube_verts, int(args.n_cubes), args.x, C.byref(args.plan), D.stream_ptr(), C.byref(h))
    if shape == "warped":
        assert rc == 1 and not h and not bool(args.grid_idx)
        return
    _native.check(rc, "mpcx_grid_plan_create")
    try:
        assert bool(args.grid_idx)
        nc = int(args.n_cubes)
        ns = [L.mpcx_grid_plan_num_intervals(h, d) for d in range(3)]
        assert ns == [int(args.grid_n[d]) for d in range(3)]
        a2 = _native.VectorArgs.from_buffer_copy(args)
        _native.check(L.mpcx_grid_plan_fill(h, C.byref(a2)), "mpcx_grid_plan_fill")
        assert np.array_equal(_dev_array(a2.grid_iv, 2 * sum(ns), np.float64), _dev_array(args.grid_iv, 2 * sum(ns), np.float64))
        assert bool(a2.grid_block_rows) == bool(args.grid_block_rows)
        assert int(a2.grid_block_rows_max) == int(args.grid_block_rows_max) == L.mpcx_grid_plan_block_rows(h)
        got_idx, ref_idx = _dev_array(a2.grid_idx, 4 * nc, np.int32).reshape(nc, 4), _dev_array(args.grid_idx, 4 * nc, np.int32).reshape(nc, 4)
        assert np.array_equal(got_idx[:, :3], ref_idx[:, :3])
        if bool(args.grid_block_rows):
            nb = int(args.plan.num_blocks)
            assert np.array_equal(_dev_array(a2.grid_block_rows, 128 * nb, np.int32), _dev_array(args.grid_block_rows, 128 * nb, np.int32))
        b2 = create_vector(case.V)
        a2.b = b2.array.data_ptr()
        _native.check(L.mpcx_assemble_vector(C.byref(a2)), "mpcx_assemble_vector")
        torch.cuda.synchronize()
        ref = oracle_outputs(oracle, case)["b"]
        # (the launch alone: the rows of slave dofs go to their masters in the same call; Dirichlet rows are the wrapper's)
        got = dm.assemble_vector(case.L, mpc).numpy()
        assert abs(got - ref).max() <= 1e-12 * max(1.0, abs(ref).max())
        assert abs(b2.numpy() - got).max() <= 1e-13 * max(1.0, abs(ref).max())
    finally:
        L.mpcx_grid_plan_destroy(h)


@pytest.mark.parametrize("degree,warp", [(2, False), (1, False), (2, True)])
def test_cell_grid_plan_from_the_c_abi_equals_the_torch_built_plan(oracle, degree, warp, monkeypatch):
    """mpcx_cell_grid_plan_create (per-cell tensor-grid tables in library-owned memory: intervals, rows per block, cell types, the
    rule's subset sums) against assemble_vector._cell_grid, array by array; the vector assembled from it against the product's;
    a mesh with warped cells has no such plan (return code 1)"""
    import torch

    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd import _device as D
    from dolfinx_mpc_amd import _native
    from dolfinx_mpc_amd.la import create_vector

    av = importlib.import_module("dolfinx_mpc_amd.assemble_vector")
    if degree == 1:
        monkeypatch.setenv("MPCX_FORCE_KERNEL", "vector=ownblock")
    case = case_cube_periodic(6, degree, 0.0, reorder=(2, 2, 2), warp=warp)
    mpc = product_mpc(case)
    args, keep = av.vector_args(case.L, 0, create_vector(case.V), mpc, 0)
    assert args.kernel_name == "ownblock"
    L = _native.lib()
    md = D.mesh_device(case.V.mesh)
    q = np.ascontiguousarray(case.L.integrals[0].kernel.qpts, dtype=np.float64).reshape(-1)
    h = C.c_void_p()
    rc = L.mpcx_cell_grid_plan_create(md["x_dofmap"].data_ptr(), case.V.mesh.num_cells, md["x"].data_ptr(), C.byref(args.plan), q.ctypes.data,
                                      q.size // 3, D.stream_ptr(), C.byref(h))
    if warp:
        assert rc == 1 and not h and not bool(args.grid_J)
        return
    _native.check(rc, "mpcx_cell_grid_plan_create")
    try:
        assert bool(args.grid_J)
        a2 = _native.VectorArgs.from_buffer_copy(args)
        a2.grid_idx = a2.grid_J = a2.grid_eta = a2.grid_iv = a2.grid_tab = a2.grid_block_rows = None
        _native.check(L.mpcx_cell_grid_plan_fill(h, C.byref(a2)), "mpcx_cell_grid_plan_fill")
        n = case.V.mesh.num_cells
        ns = [int(a2.grid_n[d]) for d in range(3)]
        assert ns == [int(args.grid_n[d]) for d in range(3)]
        assert (int(a2.grid_ng), int(a2.grid_ntypes), int(a2.grid_block_rows_max)) == (int(args.grid_ng), int(args.grid_ntypes),
                                                                                         int(args.grid_block_rows_max))
        assert np.array_equal(_dev_array(a2.grid_iv, 2 * sum(ns), np.float64), _dev_array(args.grid_iv, 2 * sum(ns), np.float64))
        assert np.allclose(_dev_array(a2.grid_eta, int(a2.grid_ng), np.float64), _dev_array(args.grid_eta, int(args.grid_ng), np.float64),
                           rtol=0, atol=1e-15)
        nj = int(a2.grid_ntypes) * (q.size // 3)
        assert np.array_equal(_dev_array(a2.grid_J, nj, np.int32), _dev_array(args.grid_J, nj, np.int32))
        assert np.array_equal(_dev_array(a2.grid_idx, 4 * n, np.int32), _dev_array(args.grid_idx, 4 * n, np.int32))
        nb = int(args.plan.num_blocks)
        assert np.array_equal(_dev_array(a2.grid_block_rows, 128 * nb, np.int32), _dev_array(args.grid_block_rows, 128 * nb, np.int32))
        b2 = create_vector(case.V)
        a2.b = b2.array.data_ptr()
        _native.check(L.mpcx_assemble_vector(C.byref(a2)), "mpcx_assemble_vector")
        torch.cuda.synchronize()
        got = dm.assemble_vector(case.L, mpc).numpy()
        ref = oracle_outputs(oracle, case)["b"]
        assert abs(got - ref).max() <= 1e-12 * max(1.0, abs(ref).max())
        assert abs(b2.numpy() - got).max() <= 1e-13 * max(1.0, abs(ref).max())
    finally:
        L.mpcx_cell_grid_plan_destroy(h)
