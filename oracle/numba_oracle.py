"""oracle/numba_oracle.py -- TEST INFRASTRUCTURE ONLY: a SECOND, independently structured
statement of the constrained assembly, following the reference's numba assemblers
(python/src/dolfinx_mpc/numba/assemble_matrix.py:100-213, 216-449 and
python/src/dolfinx_mpc/numba/assemble_vector.py:172-349) instead of its C++ ones:

1. the form is assembled WITHOUT the constraint into the MPC sparsity pattern (Dirichlet rows and
   columns of every element tensor zeroed) -- the reference calls DOLFINx's own assembler here
   (numba/assemble_matrix.py:99-100), we call the oracle's C loops with an empty constraint;
2. a correction pass over the slave entities only (plain numpy below, written from the numba source):
   ``modify_mpc_cell`` adds the master row / column / master-master terms, and the entity's own
   contribution is replaced by ``A_local - A_local_copy`` (:300-320), i.e. the slave rows and columns
   inserted in step 1 are taken out again;
3. slave diagonal and Dirichlet diagonal.

The C++ restatement (oracle/mpc_oracle.c) eliminates inside the one loop over all entities.  The two
must agree to rounding on every case (tests/test_oracle_numba.py); real arithmetic, so the numba
version's missing conjugation (:383 ``coeff**2``) makes no difference.  PARITY UNPINNED like the first
oracle (no FFCx, no DOLFINx here): this removes the risk of one shared misreading of modify_mpc_cell,
not the missing reference run.
"""

from __future__ import annotations

import numpy as np
import scipy.sparse

from . import pyoracle as po


def _element_tensor(form, integ, e, rank):
    """A_local / b_local of entity e of one integral: the kernel call of numba/assemble_matrix.py:282-290"""
    mesh = form.mesh
    ent = integ.entities[e]
    cell = int(ent if integ.itype == "cell" else ent[0])
    lf = 0 if integ.itype == "cell" else int(ent[1])
    cd = mesh.geometry.x[mesh.geometry.dofmap[cell]]
    # (Integral.coeffs packs from the live dof values on every read: pack once per assembly call, like
    # pack_coefficients in numba/assemble_matrix.py)
    packed = getattr(integ, "_oracle_pack", None)
    w = None if packed is None else packed[e]
    return cell, np.array(po.tabulate_one(integ.kernel, cd, w=w, c=integ.constants, local_facet=lf), dtype=np.float64)


def _add(store, rows, cols, block):
    """MatSetValuesLocal(ADD_VALUES) on a dict-of-keys accumulator"""
    for a, r in enumerate(rows):
        for b, c in enumerate(cols):
            store[(int(r), int(c))] = store.get((int(r), int(c)), 0.0) + float(block[a, b])


def modify_mpc_cell(store, num_dofs, bs, Ae, local_blocks, slaves, mpc: po.OracleMPC):
    """numba/assemble_matrix.py:324-449, statement by statement"""
    masters, coefficients, offsets, is_slave = mpc.masters, mpc.coeffs, mpc.masters_offsets, mpc.is_slave
    n = bs * num_dofs
    local_index0 = np.empty(len(slaves), dtype=np.int32)
    for i in range(num_dofs):
        for j in range(bs):
            slave = local_blocks[i] * bs + j
            if is_slave[slave]:
                local_index0[np.flatnonzero(slaves == slave)[0]] = i * bs + j
    Ae_original = Ae.copy()
    Ae_stripped = np.zeros((n, n))
    for i in range(num_dofs):
        for b in range(bs):
            s0 = is_slave[local_blocks[i] * bs + b]
            for j in range(num_dofs):
                for c in range(bs):
                    s1 = is_slave[local_blocks[j] * bs + c]
                    Ae_stripped[i * bs + b, j * bs + c] = (not (s0 and s1)) * Ae_original[i * bs + b, j * bs + c]
    fl_masters, fl_slaves, fl_coeffs = [], [], []
    for i, slave in enumerate(slaves):
        for k in range(offsets[slave], offsets[slave + 1]):
            fl_slaves.append(int(local_index0[i]))
            fl_masters.append(int(masters[k]))
            fl_coeffs.append(float(coefficients[k]))
    mpc_dofs = np.zeros(n, dtype=np.int64)
    for i in range(len(fl_masters)):
        li, master, coeff = fl_slaves[i], fl_masters[i], fl_coeffs[i]
        Ae[:, li] = 0
        Ae[li, :] = 0
        for j in range(num_dofs):
            for k in range(bs):
                mpc_dofs[j * bs + k] = local_blocks[j] * bs + k
        mpc_dofs[li] = master
        _add(store, mpc_dofs, [master], (coeff * Ae_stripped[:, li])[:, None])  # :400-409
        _add(store, [master], mpc_dofs, (coeff * Ae_stripped[li, :])[None, :])  # :412-421
        _add(store, [master], [master], np.array([[coeff**2 * Ae_original[li, li]]]))  # :423
        for j in range(len(fl_masters)):  # :426-436
            if i == j:
                continue
            _add(store, [master], [fl_masters[j]], np.array([[coeff * fl_coeffs[j] * Ae_original[li, fl_slaves[j]]]]))


def assemble_matrix(form, mpc: po.OracleMPC, bcs=(), diagval=1.0):
    """numba/assemble_matrix.py:47-213 for a square form on one constraint; returns scipy CSR on the MPC
    pattern (explicit zeros kept where the pattern has entries)."""
    V = form.function_spaces[0]
    assert form.function_spaces[1] is V and mpc.V is V
    bs, nd = V.dofmap.bs, V.element_ndofs
    rowptr, cols = po.create_pattern(form, mpc, mpc)
    # 1. everything, unconstrained, into the MPC pattern (no diagonals)
    A0 = po.assemble_matrix(form, po.OracleMPC.empty(V), bcs=bcs, diagval=0.0, pattern=(rowptr, cols))
    store = {}
    is_bc = np.zeros(V.num_dofs, dtype=bool)
    for bc in bcs:
        if V.contains(bc.function_space):
            m = np.zeros(V.num_dofs, dtype=np.int8)
            bc.mark_dofs(m)
            is_bc |= m.astype(bool)
    slave_cells = np.flatnonzero(np.diff(mpc.c2s_offsets) > 0)
    # 2. correction over the slave entities (cells :144-163, exterior facets :165-199)
    for integ in form.integrals:
        active = np.flatnonzero(np.isin(integ.cells, slave_cells))
        for e in active:
            if e == 0:
                integ._oracle_pack = integ.coeffs
            cell, A_local = _element_tensor(form, integ, e, 2)
            local_blocks = V.dofmap.list[cell]
            for j in range(nd):  # :294-298
                for k in range(bs):
                    if is_bc[local_blocks[j] * bs + k]:
                        A_local[j * bs + k, :] = 0
                        A_local[:, j * bs + k] = 0
            A_copy = A_local.copy()
            slaves = mpc.c2s[mpc.c2s_offsets[cell] : mpc.c2s_offsets[cell + 1]]
            modify_mpc_cell(store, nd, bs, A_local, local_blocks, slaves, mpc)
            dofs = (local_blocks[:, None] * bs + np.arange(bs)[None, :]).reshape(-1)
            _add(store, dofs, dofs, A_local - A_copy)  # :307-320
    # 3. diagonals (:201-211)
    for s in mpc.slaves[: mpc.num_local_slaves]:
        store[(int(s), int(s))] = store.get((int(s), int(s)), 0.0) + diagval
    for bc in bcs:
        if V.contains(bc.function_space):
            d_all, nowned = bc.dof_indices()
            for d in d_all[:nowned]:
                store[(int(d), int(d))] = store.get((int(d), int(d)), 0.0) + diagval
    if store:
        k = np.array(list(store.keys()), dtype=np.int64)
        corr = scipy.sparse.coo_matrix((np.array(list(store.values())), (k[:, 0], k[:, 1])), shape=A0.shape).tocsr()
    else:
        corr = scipy.sparse.csr_matrix(A0.shape)
    return A0, corr


def assemble_vector(form, mpc: po.OracleMPC):
    """numba/assemble_vector.py:40-169: full unconstrained vector, then for the slave entities
    modify_mpc_contributions (:297-349) and ``b += b_local - b_local_copy`` (:226-230)."""
    V = form.function_spaces[0]
    bs, nd = V.dofmap.bs, V.element_ndofs
    b = po.assemble_vector(form, po.OracleMPC.empty(V))
    slave_cells = np.flatnonzero(np.diff(mpc.c2s_offsets) > 0)
    for integ in form.integrals:
        for e in np.flatnonzero(np.isin(integ.cells, slave_cells)):
            if e == 0:
                integ._oracle_pack = integ.coeffs
            cell, b_local = _element_tensor(form, integ, e, 1)
            b_copy = b_local.copy()
            cell_slaves = mpc.c2s[mpc.c2s_offsets[cell] : mpc.c2s_offsets[cell + 1]]
            blocks = V.dofmap.list[cell]
            local_index = np.empty(len(cell_slaves), dtype=np.int32)
            for i in range(nd):
                for j in range(bs):
                    dof = blocks[i] * bs + j
                    if mpc.is_slave[dof]:
                        local_index[np.flatnonzero(cell_slaves == dof)[0]] = i * bs + j
            for local, slave in zip(local_index, cell_slaves):
                for k in range(mpc.masters_offsets[slave], mpc.masters_offsets[slave + 1]):
                    b[mpc.masters[k]] += mpc.coeffs[k] * b_copy[local]
                    b_local[local] = 0
            for j in range(nd):
                for k in range(bs):
                    b[blocks[j] * bs + k] += b_local[j * bs + k] - b_copy[j * bs + k]
    return b
