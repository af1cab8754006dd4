"""world_size-2 / 3 gloo tests (CPU) of the multi-GPU partition and the
interface-row exchange (dolfinx_mpc_amd/distributed.py).  The per-rank local
assembly is done by the CPU oracle here (the HIP kernels need a GPU); what is
under test is the host logic the N>1 path adds: slab meshes with owned/ghost
numbering, the integration domain (owned cells), owned-only diagonals, and the
neighbour exchange + scatter-add.  The result (owned rows, in global numbering)
must equal a single-process assembly of the global mesh."""

import os
import socket
import sys

import numpy as np
import pytest
import scipy.sparse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem(mesh, world, N, kind="poisson"):
    """poisson: BASELINE config 2's problem on the slab; elasticity: vector P1 (bs = 3)
    elasticity with a periodic tie of every component (BASELINE config 4's tensor shapes)."""
    from dolfinx_mpc_amd import fem
    from problems import periodic_raw

    zmax = float(world)
    if kind == "poisson":
        V = fem.functionspace(mesh, ("Lagrange", 1))
        dofs = fem.locate_dofs_geometrical(
            V, lambda x: np.isclose(x[1], 0) | np.isclose(x[1], 1) | np.isclose(x[2], 0) | np.isclose(x[2], zmax))
        bc = fem.dirichletbc(0.7, dofs, V)
        raw = periodic_raw(V, [bc])
        return V, bc, raw, fem.form_stiffness(V), fem.form_source(V, fem.FN_BENCH_PERIODIC)
    if kind == "p2":
        # BASELINE config 5's space: periodic Poisson, P2
        V = fem.functionspace(mesh, ("Lagrange", 2))
        dofs = fem.locate_dofs_geometrical(
            V, lambda x: np.isclose(x[1], 0) | np.isclose(x[1], 1) | np.isclose(x[2], 0) | np.isclose(x[2], zmax))
        bc = fem.dirichletbc(-0.4, dofs, V)
        raw = periodic_raw(V, [bc])
        return V, bc, raw, fem.form_stiffness(V), fem.form_source(V, fem.FN_POLY3)
    V = fem.functionspace(mesh, ("Lagrange", 1, (3,)))
    dofs = fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[1], 0) | np.isclose(x[2], zmax))
    bc = fem.dirichletbc(np.array([0.0, 0.1, -0.2]), dofs, V)
    raw = periodic_raw(V, [bc], scale=0.8)
    return V, bc, raw, fem.form_elasticity(V, 500.0, 300.0), fem.form_source(V, fem.FN_LINEAR)


def _worker(rank, world, port, N, reorder, outdir, kind="poisson"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    from dolfinx_mpc_amd.distributed import SlabExchange, create_slab_mesh
    from oracle import pyoracle as po

    # file rendezvous inside the test's tmp dir: no port to race for
    dist.init_process_group("gloo", init_method=f"file://{outdir}/rendezvous", rank=rank, world_size=world)
    mesh = create_slab_mesh(N, rank, world, reorder)
    V, bc, raw, a, L = _problem(mesh, world, N, kind)
    bs = V.dofmap.bs
    mpc = po.OracleMPC.from_raw(V, *raw)
    pattern = po.create_pattern(a, mpc, mpc)
    A = po.assemble_matrix(a, mpc, bcs=[bc], pattern=pattern)  # owned cells, owned-only diagonals
    b = po.assemble_vector(L, mpc)
    po.apply_lifting(b, [a], [[bc]], mpc)
    ex = SlabExchange(mesh, pattern[0], pattern[1], rank, world, space=V)
    vals = torch.from_numpy(A.data.copy())
    bt = torch.from_numpy(b.copy())
    ex.reduce_matrix(vals)
    ex.reduce_vector(bt)
    # the same reduction behind the reference's own calls (python/src/dolfinx_mpc/assemble_matrix.py:64 A.assemble(),
    # bench_periodic.py:108 b.ghostUpdate(ADD, REVERSE)) on the la containers, then the forward scatter
    from dolfinx_mpc_amd.distributed import exchange_for
    from dolfinx_mpc_amd.la import InsertMode, MPCMatrix, ScatterMode, Vector

    cpu = torch.device("cpu")
    Am = MPCMatrix(pattern[0], pattern[1], V.num_dofs, device=cpu)
    Am.vals.copy_(torch.from_numpy(A.data))
    Am.attach_exchange(exchange_for(V, Am))
    Am.assemble()
    bv = Vector(V.num_dofs, device=cpu)
    bv._array.copy_(torch.from_numpy(b))
    bv.attach_exchange(exchange_for(V))
    bv.ghostUpdate(addv=InsertMode.ADD, mode=ScatterMode.REVERSE)
    assert torch.equal(Am.vals, vals) and torch.equal(bv.array, bt)
    bv.ghostUpdate(addv=InsertMode.INSERT, mode=ScatterMode.FORWARD)
    A = scipy.sparse.csr_matrix((vals.numpy(), A.indices, A.indptr), shape=A.shape)
    g = (V.dof_global[:, None] * bs + np.arange(bs)[None, :]).reshape(-1)  # global unrolled dof ids
    nown = V.dofmap.index_map.size_local * bs
    Aown = A[:nown].tocoo()
    blk_send = V.dof_send_up if V.degree == 2 else mesh.node_send_up
    ghost = np.flatnonzero(np.repeat(blk_send, bs))
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), row=g[Aown.row], col=g[Aown.col], val=Aown.data,
             brow=g[:nown], bval=bt.numpy()[:nown], nslaves=mpc.num_local_slaves,
             ghost_g=g[ghost], ghost_val=bv.array.numpy()[ghost])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,N,reorder,kind", [(2, 4, None, "poisson"), (2, 4, (2, 2, 2), "poisson"),
                                                   (3, 3, (2, 2, 2), "poisson"), (2, 3, (2, 2, 2), "elasticity"),
                                                   (2, 3, (2, 2, 2), "p2"), (3, 2, None, "p2")])
def test_slab_partition_matches_global_assembly(oracle, tmp_path, world, N, reorder, kind):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(world, port, N, reorder, str(tmp_path), kind), nprocs=world, join=True)
    _check_against_global_assembly(oracle, tmp_path, world, N, kind)


def _check_against_global_assembly(oracle, tmp_path, world, N, kind):
    """the ranks' owned rows (rank{r}.npz, global dof ids) against the oracle on the unpartitioned mesh"""
    from dolfinx_mpc_amd.mesh import create_box

    gmesh = create_box((0, 0, 0), (1, 1, float(world)), (N, N, N * world))
    V, bc, raw, a, L = _problem(gmesh, world, N, kind)
    mpc = oracle.OracleMPC.from_raw(V, *raw)
    Aref = oracle.assemble_matrix(a, mpc, bcs=[bc])
    bref = oracle.assemble_vector(L, mpc)
    oracle.apply_lifting(bref, [a], [[bc]], mpc)

    n = V.num_dofs
    # global ids of the reference (single-process) numbering: nodes are numbered lexicographically,
    # P2 edges get the same structured id the slab spaces use
    bs = V.dofmap.bs
    if V.degree == 1:
        gkey = np.arange(V.num_dofs // bs, dtype=np.int64)
    else:
        from dolfinx_mpc_amd.fem import kuhn_edge_global_ids

        _, ev = gmesh.edges()
        gkey = np.concatenate([np.arange(gmesh.num_nodes, dtype=np.int64),
                               kuhn_edge_global_ids(ev[:, 0], ev[:, 1], N + 1, gmesh.num_nodes)])
    gkey = (gkey[:, None] * bs + np.arange(bs)[None, :]).reshape(-1)
    order = np.argsort(gkey)

    def to_ref(q):  # global id -> index in the reference numbering
        p = np.searchsorted(gkey[order], q)
        assert np.array_equal(gkey[order][p], q)
        return order[p]

    rows, cols, vals, brow, bval, nsl = [], [], [], [], [], 0
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        rows.append(to_ref(d["row"])), cols.append(to_ref(d["col"])), vals.append(d["val"])
        brow.append(to_ref(d["brow"])), bval.append(d["bval"])
        nsl += int(d["nslaves"])
    A = scipy.sparse.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsr()
    # every row is owned exactly once
    assert np.array_equal(np.sort(np.concatenate(brow)), np.arange(n))
    assert nsl == mpc.num_local_slaves
    assert abs(A - Aref).max() < 1e-12 * abs(Aref).max()
    b = np.zeros(n)
    b[np.concatenate(brow)] = np.concatenate(bval)
    assert np.allclose(b, bref, rtol=0, atol=1e-13 * abs(bref).max())
    # forward scatter: every ghost row of the upper interface planes holds its owner's value
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        if "ghost_g" in d and d["ghost_g"].size:
            assert np.array_equal(d["ghost_val"], b[to_ref(d["ghost_g"])])


def _gpu_worker(rank, world, port, N, reorder, outdir, kind="poisson"):
    """Same as _worker but the local assembly runs through the HIP kernels (both
    ranks share cuda:0; transport is gloo with host staging because RCCL refuses
    two ranks on one device)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd.distributed import SlabExchange, create_slab_mesh

    torch.cuda.set_device(0)
    # file rendezvous inside the test's tmp dir: no port to race for
    dist.init_process_group("gloo", init_method=f"file://{outdir}/rendezvous", rank=rank, world_size=world)
    mesh = create_slab_mesh(N, rank, world, reorder)
    V, bc, raw, a, L = _problem(mesh, world, N, kind)
    bs = V.dofmap.bs
    mpc = dm.MultiPointConstraint(V)
    mpc.add_constraint(V, *raw)
    mpc.finalize()
    # a reference-style driver, unchanged: A.assemble() inside assemble_matrix (assemble_matrix.py:64) and
    # b.ghostUpdate (bench_periodic.py:108) do the interface reduction through the attached exchange
    from dolfinx_mpc_amd.la import InsertMode, ScatterMode

    A = dm.assemble_matrix(a, mpc, bcs=[bc], algorithm="rowblock")
    b = dm.assemble_vector(L, mpc)
    dm.apply_lifting(b, [a], [[bc]], mpc)
    b.ghostUpdate(addv=InsertMode.ADD, mode=ScatterMode.REVERSE)
    torch.cuda.synchronize()
    S = A.to_scipy()
    g = (V.dof_global[:, None] * bs + np.arange(bs)[None, :]).reshape(-1)  # global unrolled dof ids
    nown = V.dofmap.index_map.size_local * bs
    Aown = S[:nown].tocoo()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), row=g[Aown.row], col=g[Aown.col], val=Aown.data,
             brow=g[:nown], bval=b.numpy()[:nown], nslaves=mpc.num_local_slaves)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,N", [("poisson", 6), ("p2", 4), ("elasticity", 4)])
def test_slab_partition_gpu_kernels_two_ranks(oracle, tmp_path, kind, N):
    """the HIP kernels on two ranks (cluster / row-pair / row-block matrix kernels, cluster and owner-computes
    vector kernels on slab meshes with ghost cells) + the interface exchange against the unpartitioned oracle"""
    import torch.multiprocessing as mp

    world, reorder = 2, (4, 4, 4)
    port = _free_port()
    mp.spawn(_gpu_worker, args=(world, port, N, reorder, str(tmp_path), kind), nprocs=world, join=True)
    _check_against_global_assembly(oracle, tmp_path, world, N, kind)


# ---------------------------------------------------------------------------------------------
# strong-scaling partition of ONE box (create_box_slab) and the two-body contact mesh (config 4)
# ---------------------------------------------------------------------------------------------
def _strong_problem(mesh, kind):
    from dolfinx_mpc_amd import fem
    from problems import periodic_raw

    if kind == "poisson":
        V = fem.functionspace(mesh, ("Lagrange", 1))
        dofs = fem.locate_dofs_geometrical(
            V, lambda x: np.isclose(x[1], 0) | np.isclose(x[1], 1) | np.isclose(x[2], 0) | np.isclose(x[2], 1))
        bc = fem.dirichletbc(0.3, dofs, V)
        return V, [bc], periodic_raw(V, [bc]), fem.form_stiffness(V), fem.form_source(V, fem.FN_BENCH_PERIODIC)
    if kind == "p2":  # BASELINE config 5's space on a strong-scaling partition
        V = fem.functionspace(mesh, ("Lagrange", 2))
        dofs = fem.locate_dofs_geometrical(
            V, lambda x: np.isclose(x[1], 0) | np.isclose(x[1], 1) | np.isclose(x[2], 0) | np.isclose(x[2], 1))
        bc = fem.dirichletbc(-0.4, dofs, V)
        return V, [bc], periodic_raw(V, [bc]), fem.form_stiffness(V), fem.form_source(V, fem.FN_POLY3)
    raise ValueError(kind)


def _strong_worker(rank, world, outdir, kind, n, axis, reorder):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    from dolfinx_mpc_amd.distributed import SlabExchange, create_box_slab, create_stacked_cubes_slab
    from oracle import pyoracle as po

    dist.init_process_group("gloo", init_method=f"file://{outdir}/rendezvous", rank=rank, world_size=world)
    if kind == "contact":
        import dolfinx_mpc_amd as dm
        from dolfinx_mpc_amd import fem
        from dolfinx_mpc_amd.mesh import (CONTACT_BOTTOM, CONTACT_BOTTOM_INTERFACE, CONTACT_TOP,
                                          CONTACT_TOP_INTERFACE)

        mesh, ft = create_stacked_cubes_slab(n, rank, world, theta=0.3, reorder=reorder, axis=axis)
        V = fem.functionspace(mesh, ("Lagrange", 1, (3,)))
        u_top = fem.Function(V)
        u_top.x.array[2::3] = -4.25e-1
        bcs = [fem.dirichletbc(fem.Function(V), fem.locate_dofs_topological(V, 2, ft.find(CONTACT_BOTTOM)), V),
               fem.dirichletbc(u_top, fem.locate_dofs_topological(V, 2, ft.find(CONTACT_TOP)), V)]
        a = fem.form_elasticity(V, 500.0, 0.0)
        L = fem.form_source(V, fem.FN_CONSTANT_VEC, constant=[1.0, 0.3, -0.2, -1.0])
        # the product's builder on the LOCAL mesh (owned + ghost cells): every local slave finds its masters locally
        pm = dm.MultiPointConstraint(V)
        pm.create_contact_inelastic_condition(ft, CONTACT_BOTTOM_INTERFACE, CONTACT_TOP_INTERFACE)
        raw = (pm._slaves, pm._masters, pm._coeffs, pm._owners, pm._offsets)
    else:
        mesh = create_box_slab((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), (n, n, n), rank, world, axis, reorder)
        V, bcs, raw, a, L = _strong_problem(mesh, kind)
    bs = V.dofmap.bs
    mpc = po.OracleMPC.from_raw(V, *raw)
    pattern = po.create_pattern(a, mpc, mpc)
    A = po.assemble_matrix(a, mpc, bcs=bcs, pattern=pattern)
    b = po.assemble_vector(L, mpc)
    po.apply_lifting(b, [a], [bcs], mpc)
    ex = SlabExchange(mesh, pattern[0], pattern[1], rank, world, bs=bs, space=V if V.degree == 2 else None)
    vals = torch.from_numpy(A.data.copy())
    bt = torch.from_numpy(b.copy())
    ex.reduce_matrix(vals)
    ex.reduce_vector(bt)
    A = scipy.sparse.csr_matrix((vals.numpy(), A.indices, A.indptr), shape=A.shape)
    blk_global = V.dof_global if V.degree == 2 else mesh.node_global
    g = (blk_global[:, None] * bs + np.arange(bs)[None, :]).reshape(-1)
    nown = V.dofmap.index_map.size_local * bs
    Aown = A[:nown].tocoo()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), row=g[Aown.row], col=g[Aown.col], val=Aown.data,
             brow=g[:nown], bval=bt.numpy()[:nown], nslaves=mpc.num_local_slaves, ncells=mesh.num_owned_cells)
    dist.barrier()
    dist.destroy_process_group()


def _gather(tmp_path, world, n):
    rows, cols, vals, brow, bval, nsl, ncells = [], [], [], [], [], 0, 0
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        rows.append(d["row"]), cols.append(d["col"]), vals.append(d["val"])
        brow.append(d["brow"]), bval.append(d["bval"])
        nsl += int(d["nslaves"])
        ncells += int(d["ncells"])
    A = scipy.sparse.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsr()
    assert np.array_equal(np.sort(np.concatenate(brow)), np.arange(n))  # every row owned exactly once
    b = np.zeros(n)
    b[np.concatenate(brow)] = np.concatenate(bval)
    return A, b, nsl, ncells


@pytest.mark.parametrize("world,n,axis,reorder", [(2, 4, 2, None), (3, 5, 2, (2, 2, 2)), (2, 5, 1, (2, 2, 2)), (4, 6, 2, None)])
def test_strong_scaling_partition_matches_global_assembly(oracle, tmp_path, world, n, axis, reorder):
    """ONE n^3 unit cube cut into `world` slabs of layers (uneven when world does not divide n): the
    union of the ranks' owned rows after the exchange is the single-process matrix and vector."""
    import torch.multiprocessing as mp

    from dolfinx_mpc_amd.mesh import create_unit_cube

    mp.spawn(_strong_worker, args=(world, str(tmp_path), "poisson", n, axis, reorder), nprocs=world, join=True)
    gmesh = create_unit_cube(n, n, n)
    V, bcs, raw, a, L = _strong_problem(gmesh, "poisson")
    mpc = oracle.OracleMPC.from_raw(V, *raw)
    Aref = oracle.assemble_matrix(a, mpc, bcs=bcs)
    bref = oracle.assemble_vector(L, mpc)
    oracle.apply_lifting(bref, [a], [bcs], mpc)
    A, b, nsl, ncells = _gather(tmp_path, world, V.num_dofs)
    assert ncells == gmesh.num_cells and nsl == mpc.num_local_slaves
    assert abs(A - Aref).max() < 1e-12 * abs(Aref).max()
    assert np.allclose(b, bref, rtol=0, atol=1e-13 * max(1.0, abs(bref).max()))


@pytest.mark.parametrize("world,n", [(2, 3), (3, 4)])
def test_strong_scaling_partition_p2(oracle, tmp_path, world, n):
    """P2 (config 5's space) on an uneven strong-scaling z-partition against the single-process assembly."""
    import torch.multiprocessing as mp

    from dolfinx_mpc_amd.fem import kuhn_edge_global_ids
    from dolfinx_mpc_amd.mesh import create_unit_cube

    mp.spawn(_strong_worker, args=(world, str(tmp_path), "p2", n, 2, None), nprocs=world, join=True)
    gmesh = create_unit_cube(n, n, n)
    V, bcs, raw, a, L = _strong_problem(gmesh, "p2")
    mpc = oracle.OracleMPC.from_raw(V, *raw)
    Aref = oracle.assemble_matrix(a, mpc, bcs=bcs)
    bref = oracle.assemble_vector(L, mpc)
    oracle.apply_lifting(bref, [a], [bcs], mpc)
    _, ev = gmesh.edges()
    gkey = np.concatenate([np.arange(gmesh.num_nodes, dtype=np.int64),
                           kuhn_edge_global_ids(ev[:, 0], ev[:, 1], n + 1, gmesh.num_nodes)])
    order = np.argsort(gkey)

    def to_ref(q):
        p = np.searchsorted(gkey[order], q)
        assert np.array_equal(gkey[order][p], q)
        return order[p]

    nd = V.num_dofs
    rows, cols, vals, brow, bval = [], [], [], [], []
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        rows.append(to_ref(d["row"])), cols.append(to_ref(d["col"])), vals.append(d["val"])
        brow.append(to_ref(d["brow"])), bval.append(d["bval"])
    A = scipy.sparse.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nd, nd)).tocsr()
    assert np.array_equal(np.sort(np.concatenate(brow)), np.arange(nd))
    assert abs(A - Aref).max() < 1e-12 * abs(Aref).max()
    b = np.zeros(nd)
    b[np.concatenate(brow)] = np.concatenate(bval)
    assert np.allclose(b, bref, rtol=0, atol=1e-13 * max(1.0, abs(bref).max()))


@pytest.mark.parametrize("world,n_top,reorder", [(2, 2, None), (2, 4, (2, 2, 2))])
def test_contact_two_body_partition_matches_global_assembly(oracle, tmp_path, world, n_top, reorder):
    """BASELINE config 4 on `world` ranks: both bodies cut along y at the same positions, so the slave
    and master layers of a slab are on one rank and the constraint needs no exchange of its own; the
    interface rows (including master rows that collected slave contributions) travel in the usual
    neighbour exchange.  Against a single-process assembly with the brute-force constraint."""
    import torch.multiprocessing as mp

    from problems import case_contact_two_body

    mp.spawn(_strong_worker, args=(world, str(tmp_path), "contact", n_top, 1, reorder), nprocs=world, join=True)
    case = case_contact_two_body(n_top, None, 0.3)
    mpc = oracle.OracleMPC.from_raw(case.V, *case.raw)
    a = __import__("dolfinx_mpc_amd").fem.form_elasticity(case.V, 500.0, 0.0)
    Aref = oracle.assemble_matrix(a, mpc, bcs=case.bcs)
    bref = oracle.assemble_vector(case.L, mpc)
    oracle.apply_lifting(bref, [a], [case.bcs], mpc)
    A, b, nsl, ncells = _gather(tmp_path, world, case.V.num_dofs)
    assert ncells == case.mesh.num_cells and nsl == mpc.num_local_slaves
    assert abs(A - Aref).max() < 1e-12 * abs(Aref).max()
    assert np.allclose(b, bref, rtol=0, atol=1e-12 * max(1.0, abs(bref).max()))


def _rccl_worker(rank, world, outdir, n):
    """strong-scaling Poisson on `world` GPUs, HIP kernels, exchange over RCCL (backend "nccl")"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd.distributed import SlabExchange, create_box_slab

    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", init_method=f"file://{outdir}/rendezvous", rank=rank, world_size=world,
                            device_id=torch.device("cuda", rank))
    mesh = create_box_slab((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), (n, n, n), rank, world, 2, (4, 4, 4))
    V, bcs, raw, a, L = _strong_problem(mesh, "poisson")
    mpc = dm.MultiPointConstraint(V)
    mpc.add_constraint(V, *raw)
    mpc.finalize()
    from dolfinx_mpc_amd.la import InsertMode, ScatterMode

    A = dm.assemble_matrix(a, mpc, bcs=bcs)  # A.assemble() posts the RCCL send/recv of the interface rows
    b = dm.assemble_vector(L, mpc)  # ... which travel while this runs
    dm.apply_lifting(b, [a], [bcs], mpc)
    b.ghostUpdate(addv=InsertMode.ADD, mode=ScatterMode.REVERSE)
    torch.cuda.synchronize()
    S = A.to_scipy()
    g = mesh.node_global
    nown = mesh.num_owned_nodes
    Aown = S[:nown].tocoo()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), row=g[Aown.row], col=g[Aown.col], val=Aown.data,
             brow=g[:nown], bval=b.numpy()[:nown], nslaves=mpc.num_local_slaves, ncells=mesh.num_owned_cells)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_strong_scaling_partition_over_rccl(oracle, tmp_path):
    """the exchange on backend "nccl" (RCCL over xGMI), one rank per GPU: needs at least two visible devices,
    skipped on the single-GPU boxes (the gloo tests above cover the logic; this covers the transport)"""
    import torch
    import torch.multiprocessing as mp

    from dolfinx_mpc_amd.mesh import create_unit_cube

    world = min(torch.cuda.device_count(), 4)
    if world < 2:
        pytest.skip("needs >= 2 GPUs")
    n = 12
    mp.spawn(_rccl_worker, args=(world, str(tmp_path), n), nprocs=world, join=True)
    gmesh = create_unit_cube(n, n, n)
    V, bcs, raw, a, L = _strong_problem(gmesh, "poisson")
    mpc = oracle.OracleMPC.from_raw(V, *raw)
    Aref = oracle.assemble_matrix(a, mpc, bcs=bcs)
    bref = oracle.assemble_vector(L, mpc)
    oracle.apply_lifting(bref, [a], [bcs], mpc)
    A, b, nsl, ncells = _gather(tmp_path, world, V.num_dofs)
    assert ncells == gmesh.num_cells and nsl == mpc.num_local_slaves
    assert abs(A - Aref).max() < 1e-12 * abs(Aref).max()
    assert np.allclose(b, bref, rtol=0, atol=1e-12 * max(1.0, abs(bref).max()))
