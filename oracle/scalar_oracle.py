"""oracle/scalar_oracle.py -- TEST INFRASTRUCTURE ONLY: the constrained assembly for ANY scalar type T (float32, float64,
complex64, complex128), a plain-Python restatement of the reference's templated C++ loops with the Hermitian transpose on
the row side for complex T:

    cpp/assemble_matrix.cpp:99-268   modify_mpc_cell  (coeff_i = conj(coeff) for complex T, :219-223)
    cpp/assemble_matrix.cpp:417-548  assemble_cells_impl / :271-415 exterior facets, :662-726 driver (slave diagonal)
    cpp/assemble_vector.h:35-69      modify_mpc_vec   (conj for complex T, :59-65)
    cpp/assemble_vector.cpp:34-91    _assemble_entities_impl
    cpp/lifting.h:45-134             lift_bc_entities
    python/src/dolfinx_mpc/assemble_matrix.py:43-65 (Dirichlet diagonal), assemble_vector.py:25-104

The real oracle (mpc_oracle.c) is fp64-real.  Element tensors of type T come from ITS real element kernels through the
multilinearity of the built-in forms in their data (scale constant, coefficient function, the vector constant of
FN_CONSTANT_VEC; elasticity: linear in (mu, lambda)): real and imaginary parts are tabulated separately and combined --
geometry and basis functions are real in the reference as well (U = real(T)).  Everything else (Dirichlet zeroing,
stripped / original tensors, flattened masters, conjugation, insertion, lifting) is written out here, statement by
statement, in numpy scalars of type T.  Small meshes only (pure-Python loops).  PARITY UNPINNED like the other oracles."""

from __future__ import annotations

import numpy as np
import scipy.sparse

from . import pyoracle as po


class ScalarMPC:
    """finalized constraint with coefficients of type T (cpp/MultiPointConstraint.h:36-126)"""

    def __init__(self, V, slaves, masters, coeffs, owners, offsets, dtype):
        self.V = V
        self.dtype = np.dtype(dtype)
        n = V.num_dofs
        slaves = np.asarray(slaves, dtype=np.int64)
        self.is_slave = np.zeros(n, dtype=bool)
        self.is_slave[slaves] = True
        self.masters_of = {}
        coeffs = np.asarray(coeffs).astype(self.dtype)
        for i, s in enumerate(slaves):
            self.masters_of[int(s)] = [(int(masters[j]), coeffs[j]) for j in range(offsets[i], offsets[i + 1])]
        self.slaves = np.flatnonzero(self.is_slave)
        nowned = V.dofmap.index_map.size_local * V.dofmap.index_map_bs
        self.num_local_slaves = int(np.searchsorted(self.slaves, nowned))
        self.real = po.OracleMPC.from_raw(V, slaves.astype(np.int32), masters, np.ones(len(masters)), owners, offsets)  # structure only

    def links(self, dof):
        return self.masters_of.get(int(dof), [])


def _real_tensor(kernel, cd, w, c, lf):
    return np.asarray(po.tabulate_one(kernel, cd, w=w, c=c, local_facet=lf), dtype=np.float64)


def element_tensor(integ, cd, w, lf, dtype):
    """tensor of one entity in T (see the module docstring); w: packed coefficient values of the entity (T) or None"""
    k = integ.kernel
    c = integ.constants
    T = np.dtype(dtype)
    if not np.issubdtype(T, np.complexfloating):
        wr = None if w is None else np.asarray(w, dtype=np.float64)
        cr = None if c is None else np.asarray(c.real if np.iscomplexobj(c) else c, dtype=np.float64)
        return _real_tensor(k, cd, wr, cr, lf).astype(T)
    c = None if c is None else np.asarray(c, dtype=np.complex128)
    if k.form == 3:  # elasticity: A = mu T(1, 0) + lambda T(0, 1)
        return c[0] * _real_tensor(k, cd, None, np.array([1.0, 0.0]), lf) + c[1] * _real_tensor(k, cd, None, np.array([0.0, 1.0]), lf)
    vecconst = k.form in (2, 5) and k.fn_id == 5 and c is not None
    c0 = 1.0 + 0.0j if c is None else c[0]
    out = 0.0
    w_parts = [(1.0, None)] if w is None else [(1.0, np.asarray(w).real.copy()), (1.0j, np.asarray(w).imag.copy())]
    g_parts = [(1.0, None)] if not vecconst else [(1.0, c[1:].real.copy()), (1.0j, c[1:].imag.copy())]
    for fw, wr in w_parts:
        for fg, gr in g_parts:
            cr = np.array([1.0]) if gr is None else np.concatenate([[1.0], gr])
            out = out + (fw * fg) * _real_tensor(k, cd, wr, cr, lf)
    return (c0 * out).astype(T)


def _entity(form, integ, e):
    mesh = form.mesh
    ent = integ.entities[e]
    cell = int(ent if integ.itype == "cell" else ent[0])
    lf = 0 if integ.itype == "cell" else int(ent[1])
    return cell, lf, mesh.geometry.x[mesh.geometry.dofmap[cell]]


def _unrolled(V, cell):
    bs = V.dofmap.bs
    return (V.dofmap.list[cell].astype(np.int64)[:, None] * bs + np.arange(bs)[None, :]).reshape(-1)


def _bc_markers(V, bcs):
    m = np.zeros(V.num_dofs, dtype=bool)
    for bc in bcs:
        if V.contains(bc.function_space):
            t = np.zeros(V.num_dofs, dtype=np.int8)
            bc.mark_dofs(t)
            m |= t.astype(bool)
    return m


def assemble_matrix(form, mpc0: ScalarMPC, mpc1: ScalarMPC = None, bcs=(), diagval=1.0):
    """scipy CSR of type T on the MPC sparsity pattern (the real oracle's pattern builder)"""
    mpc1 = mpc0 if mpc1 is None else mpc1
    T = np.dtype(form.dtype)
    V0, V1 = form.function_spaces
    rowptr, cols = po.create_pattern(form, mpc0.real, mpc1.real)
    store = {}

    def mat_set(rows, cs, block):
        for a, r in enumerate(rows):
            for b, c in enumerate(cs):
                store[(int(r), int(c))] = store.get((int(r), int(c)), T.type(0)) + block[a, b]

    bc0, bc1 = _bc_markers(V0, bcs), _bc_markers(V1, bcs)
    for integ in form.integrals:
        packed = integ.coeffs
        for e in range(integ.num_entities):
            cell, lf, cd = _entity(form, integ, e)
            Ae = element_tensor(integ, cd, None if packed is None else packed[e], lf, T).copy()
            d0, d1 = _unrolled(V0, cell), _unrolled(V1, cell)
            # Dirichlet rows and columns first (:510-533)
            Ae[bc0[d0], :] = 0
            Ae[:, bc1[d1]] = 0
            s0 = [p for p, d in enumerate(d0) if mpc0.is_slave[d]]
            s1 = [q for q, d in enumerate(d1) if mpc1.is_slave[d]]
            if s0 or s1:
                # modify_mpc_cell (:99-268)
                O = Ae.copy()
                S = Ae.copy()
                for p in s0:
                    for q in s1:
                        S[p, q] = 0  # fill_stripped_matrix (:33-77): slave-slave entries removed
                Ae[s0, :] = 0
                Ae[:, s1] = 0
                F0 = [(p, m, c) for p in s0 for (m, c) in mpc0.links(d0[p])]
                F1 = [(q, m, c) for q in s1 for (m, c) in mpc1.links(d1[q])]
                for (p, m, c) in F0:
                    ci = np.conj(c)  # Hermitian transpose on the row side (:219-223)
                    mat_set([m], d1, (ci * S[p, :])[None, :])
                    for (q, m2, c2) in F1:
                        mat_set([m], [m2], np.array([[ci * c2 * O[p, q]]]))
                for (q, m, c) in F1:
                    mat_set(d0, [m], (c * S[:, q])[:, None])
            mat_set(d0, d1, Ae)
    # slave diagonal (:711-724) and Dirichlet diagonal (assemble_matrix.py:59-62), square forms only
    if V0 is V1 and mpc0.V is mpc1.V:
        for s in mpc0.slaves[: mpc0.num_local_slaves]:
            store[(int(s), int(s))] = store.get((int(s), int(s)), T.type(0)) + T.type(diagval)
    if V0 is V1:
        for bc in bcs:
            if V0.contains(bc.function_space):
                dofs, nowned = bc.dof_indices()
                for d in dofs[:nowned]:
                    store[(int(d), int(d))] = store.get((int(d), int(d)), T.type(0)) + T.type(diagval)
    vals = np.zeros(cols.size, dtype=T)
    for (r, c), v in store.items():
        lo, hi = rowptr[r], rowptr[r + 1]
        pos = lo + np.searchsorted(cols[lo:hi], c)
        assert pos < hi and cols[pos] == c, "entry outside the MPC pattern"
        vals[pos] = v
    return scipy.sparse.csr_matrix((vals, cols, rowptr), shape=(V0.num_dofs, V1.num_dofs))


def _modify_mpc_vec(b, be, dofs, mpc: ScalarMPC):
    """cpp/assemble_vector.h:35-69; be[slave] is zeroed INSIDE the master loop, as there"""
    be_copy = be.copy()
    for p, d in enumerate(dofs):
        if mpc.is_slave[d]:
            for (m, c) in mpc.links(d):
                b[m] += np.conj(c) * be_copy[p]
                be[p] = 0


def assemble_vector(form, mpc: ScalarMPC):
    T = np.dtype(form.dtype)
    V = form.function_spaces[0]
    b = np.zeros(V.num_dofs, dtype=T)
    for integ in form.integrals:
        packed = integ.coeffs
        for e in range(integ.num_entities):
            cell, lf, cd = _entity(form, integ, e)
            be = element_tensor(integ, cd, None if packed is None else packed[e], lf, T).copy()
            dofs = _unrolled(V, cell)
            _modify_mpc_vec(b, be, dofs, mpc)
            np.add.at(b, dofs, be)
    return b


def apply_lifting(b, forms, bcs, mpc: ScalarMPC, x0=None, scale=1.0):
    """cpp/lifting.h:45-134, 151-416: b -= scale K^H A (g - x0), A the raw kernel output"""
    for j, a in enumerate(forms):
        if a is None or len(bcs[j]) == 0:
            continue
        T = np.dtype(a.dtype)
        V0, V1 = a.function_spaces
        marker = _bc_markers(V1, bcs[j])
        g = np.zeros(V1.num_dofs, dtype=T)
        for bc in bcs[j]:
            bc.set(g)
        x0j = None if not x0 else np.asarray(x0[j])
        for integ in a.integrals:
            packed = integ.coeffs
            for e in range(integ.num_entities):
                cell, lf, cd = _entity(a, integ, e)
                d1 = _unrolled(V1, cell)
                if not marker[d1].any():
                    continue
                Ae = element_tensor(integ, cd, None if packed is None else packed[e], lf, T)
                d0 = _unrolled(V0, cell)
                be = np.zeros(d0.size, dtype=T)
                for q, dj in enumerate(d1):
                    if marker[dj]:
                        be -= Ae[:, q] * (scale * (g[dj] - (0 if x0j is None else x0j[dj])))
                _modify_mpc_vec(b, be, d0, mpc)
                np.add.at(b, d0, be)
    return b


def backsubstitution(mpc: ScalarMPC, u):
    for s in mpc.slaves:
        u[s] = sum(c * u[m] for (m, c) in mpc.links(s))
    return u
