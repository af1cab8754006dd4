"""The reference's own assembly tests, statement for statement, with ``dolfinx_mpc_amd`` in the place of ``dolfinx_mpc``
(python/tests/test_matrix_assembly.py:23-102, python/tests/test_vector_assembly.py:22-63): the unconstrained system
comes from the same HIP assembler with an empty constraint (the reference takes DOLFINx's), the check is the reference's
``dolfinx_mpc.utils.compare_mpc_lhs / compare_mpc_rhs`` (product side: dolfinx_mpc_amd/utils.py) -- no oracle involved.
Sweep: triangles and quadrilaterals, degree 1-3, both master choices."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def l2b(li):
    return np.array(li, dtype=np.float64).tobytes()


def _s_m_c(master_point):
    return {l2b([1, 0]): {l2b([0, 1]): 0.43, l2b([1, 1]): 0.11}, l2b([0, 0]): {l2b(master_point): 0.69}}


@pytest.mark.parametrize("master_point", [[1, 1], [0, 1]])
@pytest.mark.parametrize("degree", range(1, 4))
@pytest.mark.parametrize("celltype", ["quadrilateral", "triangle"])
@pytest.mark.parametrize("shape", [(5, 3), (1, 8)], ids=["5x3", "slaves-on-one-cell"])
def test_mpc_assembly_matrix(master_point, degree, celltype, shape):
    import dolfinx_mpc_amd
    import dolfinx_mpc_amd.utils
    from dolfinx_mpc_amd import fem
    from dolfinx_mpc_amd.mesh import create_unit_square

    mesh = create_unit_square(shape[0], shape[1], celltype)
    V = fem.functionspace(mesh, ("Lagrange", degree))
    bilinear_form = fem.form_stiffness(V)  # inner(grad(u), grad(v)) * dx
    mpc = dolfinx_mpc_amd.MultiPointConstraint(V)
    mpc.create_general_constraint(_s_m_c(master_point))
    mpc.finalize()
    A_mpc = dolfinx_mpc_amd.assemble_matrix(bilinear_form, mpc)
    # the globally reduced system
    plain = dolfinx_mpc_amd.MultiPointConstraint(V)
    plain.finalize()
    A_org = dolfinx_mpc_amd.assemble_matrix(bilinear_form, plain)
    dolfinx_mpc_amd.utils.compare_mpc_lhs(A_org, A_mpc, mpc)


@pytest.mark.parametrize("master_point", [[1, 1], [0, 1]])
@pytest.mark.parametrize("degree", range(1, 4))
@pytest.mark.parametrize("celltype", ["quadrilateral", "triangle"])
def test_mpc_assembly_vector(master_point, degree, celltype):
    import dolfinx_mpc_amd
    import dolfinx_mpc_amd.utils
    from dolfinx_mpc_amd import fem
    from dolfinx_mpc_amd.la import InsertMode, ScatterMode
    from dolfinx_mpc_amd.mesh import create_unit_square

    mesh = create_unit_square(3, 5, celltype)
    V = fem.functionspace(mesh, ("Lagrange", degree))
    linear_form = fem.form_source(V, fem.FN_SIN2D)  # f = sin(2 pi x) sin(pi y)
    mpc = dolfinx_mpc_amd.MultiPointConstraint(V)
    mpc.create_general_constraint(_s_m_c(master_point))
    mpc.finalize()
    b = dolfinx_mpc_amd.assemble_vector(linear_form, mpc)
    b.ghostUpdate(addv=InsertMode.ADD, mode=ScatterMode.REVERSE)
    plain = dolfinx_mpc_amd.MultiPointConstraint(V)
    plain.finalize()
    L_org = dolfinx_mpc_amd.assemble_vector(linear_form, plain)
    L_org.ghostUpdate(addv=InsertMode.ADD, mode=ScatterMode.REVERSE)
    dolfinx_mpc_amd.utils.compare_mpc_rhs(L_org, b, mpc, root=0)
