#!/usr/bin/env python
"""Generate the golden fixtures in tests/golden/*.npz.

The reference (dolfinx_mpc C++/numba assemblers) cannot be built or imported in
this container (DOLFINx, Basix, FFCx, PETSc, MPI absent -- SURVEY.md section
8c), so the vectors are produced by the CPU oracle (oracle/mpc_oracle.c), whose
MPC algebra is pinned by the reference's own K^T A K / K^T b identities
(tests/test_oracle_identities.py).  A fixture is data only: the assembled CSR
matrix and vectors of each small configuration of tests/problems.py, plus
checksums of BASELINE config 1 (periodic Poisson P1, 32^3).

    python tests/golden/generate.py
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import pyoracle as po  # noqa: E402
from problems import all_small_cases, case_cube_periodic, oracle_outputs  # noqa: E402


def checksums(out):
    A = out["A"].tocsr()
    n = A.shape[0]
    rng = np.random.default_rng(1234)
    v = rng.standard_normal(n)
    idx = rng.choice(A.nnz, size=min(512, A.nnz), replace=False)
    idx.sort()
    d = dict(frob=np.sqrt((A.data**2).sum()), Av=(A @ v)[:: max(1, n // 256)].copy(), vAv=v @ (A @ v), diag_sum=A.diagonal().sum(),
             sample_idx=idx, sample_val=A.data[idx], nnz=A.nnz)
    for k in ("b", "b_lifted"):
        d[k + "_sum"] = out[k].sum()
        d[k + "_norm"] = np.linalg.norm(out[k])
        d[k + "_sample"] = out[k][:: max(1, n // 256)].copy()
    return d


def main():
    for make in all_small_cases():
        case = make()
        out = oracle_outputs(po, case)
        d = {}
        if "A" in out:
            A = out["A"].tocsr()
            d.update(A_indptr=A.indptr.astype(np.int32), A_indices=A.indices.astype(np.int32), A_data=A.data)
        for k in ("b", "b_lifted"):
            if k in out:
                d[k] = out[k]
        np.savez_compressed(os.path.join(HERE, case.name + ".npz"), **d)
        print("wrote", case.name, {k: v.shape for k, v in d.items()})
    case = case_cube_periodic(32, 1, 0.0)
    out = oracle_outputs(po, case, fast=True)
    np.savez_compressed(os.path.join(HERE, "config1_cube32_checksums.npz"), **checksums(out))
    print("wrote config1 checksums")


if __name__ == "__main__":
    main()
