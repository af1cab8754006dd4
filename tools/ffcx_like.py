#!/usr/bin/env python
"""Command line of dolfinx_mpc_amd/codegen.py (the FFCx stand-in): print the C text of a UFCx tabulate_tensor.

    python tools/ffcx_like.py stiffness tetrahedron 2 [bs]
    python tools/ffcx_like.py stiffness hexahedron 1 [bs]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dolfinx_mpc_amd.codegen import BENCH_PERIODIC_F, generate, generate_hex  # noqa: E402,F401
from dolfinx_mpc_amd.quadrature import make_quadrature  # noqa: E402

if __name__ == "__main__":
    kind, cell, degree = sys.argv[1], sys.argv[2], int(sys.argv[3])
    bs = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    if cell == "hexahedron":
        print(generate_hex(kind, bs)[0])
    else:
        qdeg = {"stiffness": 2 * (degree - 1), "mass": 2 * degree, "elasticity": 2 * (degree - 1), "source": degree + 2}[kind]
        print(generate(kind, cell, degree, bs, make_quadrature(cell, qdeg))[0])
