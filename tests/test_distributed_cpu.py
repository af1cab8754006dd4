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
    A = scipy.sparse.csr_matrix((vals.numpy(), A.indices, A.indptr), shape=A.shape)
    g = (V.dof_global[:, None] * bs + np.arange(bs)[None, :]).reshape(-1)  # global unrolled dof ids
    nown = V.dofmap.index_map.size_local * bs
    Aown = A[:nown].tocoo()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), row=g[Aown.row], col=g[Aown.col], val=Aown.data,
             brow=g[:nown], bval=bt.numpy()[:nown], nslaves=mpc.num_local_slaves)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,N,reorder,kind", [(2, 4, None, "poisson"), (2, 4, (2, 2, 2), "poisson"),
                                                   (3, 3, (2, 2, 2), "poisson"), (2, 3, (2, 2, 2), "elasticity"),
                                                   (2, 3, (2, 2, 2), "p2"), (3, 2, None, "p2")])
def test_slab_partition_matches_global_assembly(oracle, tmp_path, world, N, reorder, kind):
    import torch.multiprocessing as mp

    from dolfinx_mpc_amd.mesh import create_box

    port = _free_port()
    mp.spawn(_worker, args=(world, port, N, reorder, str(tmp_path), kind), nprocs=world, join=True)

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


def _gpu_worker(rank, world, port, N, reorder, outdir):
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
    V, bc, raw, a, L = _problem(mesh, world, N)
    mpc = dm.MultiPointConstraint(V)
    mpc.add_constraint(V, *raw)
    mpc.finalize()
    A = dm.assemble_matrix(a, mpc, bcs=[bc], algorithm="rowblock")
    b = dm.assemble_vector(L, mpc)
    dm.apply_lifting(b, [a], [[bc]], mpc)
    ex = SlabExchange(mesh, A.rowptr, A.cols, rank, world, device=torch.device("cuda", 0))
    ex.reduce_matrix(A)
    ex.reduce_vector(b)
    torch.cuda.synchronize()
    S = A.to_scipy()
    g = mesh.node_global
    nown = mesh.num_owned_nodes
    Aown = S[:nown].tocoo()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), row=g[Aown.row], col=g[Aown.col], val=Aown.data,
             brow=g[:nown], bval=b.numpy()[:nown], nslaves=mpc.num_local_slaves)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_slab_partition_gpu_kernels_two_ranks(oracle, tmp_path):
    import torch.multiprocessing as mp

    from dolfinx_mpc_amd.mesh import create_box

    world, N, reorder = 2, 6, (4, 4, 4)
    port = _free_port()
    mp.spawn(_gpu_worker, args=(world, port, N, reorder, str(tmp_path)), nprocs=world, join=True)
    gmesh = create_box((0, 0, 0), (1, 1, float(world)), (N, N, N * world))
    V, bc, raw, a, L = _problem(gmesh, world, N)
    mpc = oracle.OracleMPC.from_raw(V, *raw)
    Aref = oracle.assemble_matrix(a, mpc, bcs=[bc])
    bref = oracle.assemble_vector(L, mpc)
    oracle.apply_lifting(bref, [a], [[bc]], mpc)
    n = V.num_dofs
    rows, cols, vals, brow, bval = [], [], [], [], []
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        rows.append(d["row"]), cols.append(d["col"]), vals.append(d["val"])
        brow.append(d["brow"]), bval.append(d["bval"])
    A = scipy.sparse.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsr()
    assert abs(A - Aref).max() < 1e-12 * abs(Aref).max()
    b = np.zeros(n)
    b[np.concatenate(brow)] = np.concatenate(bval)
    assert np.allclose(b, bref, rtol=0, atol=1e-12 * abs(bref).max())
