"""tests/test_independent_checks.py on the GPU product: the HIP element kernels, dof tables and facet
conventions against analytic integrals of polynomials (no oracle in between)."""

import numpy as np
import pytest

from dolfinx_mpc_amd import fem
from dolfinx_mpc_amd.mesh import create_unit_cube
from test_independent_checks import gauss_cube, poly3, u_fun, v_fun

pytestmark = pytest.mark.gpu


def _none(V):
    import dolfinx_mpc_amd as dm

    m = dm.MultiPointConstraint(V)
    m.finalize()
    return m


@pytest.mark.parametrize("alg", ["atomic", "rowblock"])
@pytest.mark.parametrize("degree", [1, 2])
@pytest.mark.parametrize("reorder", [None, (2, 2, 2)])
def test_product_forms_reproduce_analytic_integrals(degree, reorder, alg):
    import dolfinx_mpc_amd as dm

    mesh = create_unit_cube(3, 3, 3, reorder=reorder)
    V = fem.functionspace(mesh, ("Lagrange", degree))
    X = V.tabulate_dof_coordinates().T
    (u, gu), (v, gv) = u_fun(degree), v_fun(degree)
    U, W = u(X), v(X)
    none = _none(V)
    A = dm.assemble_matrix(fem.form_stiffness(V), none, algorithm=alg).to_scipy()
    M = dm.assemble_matrix(fem.form_mass(V), none, algorithm=alg).to_scipy()
    b = dm.assemble_vector(fem.form_source(V, fem.FN_POLY3), none, algorithm=alg).numpy()
    assert U @ (A @ W) == pytest.approx(gauss_cube(lambda x: np.sum(gu(x) * gv(x), axis=0)), rel=1e-12)
    assert U @ (M @ W) == pytest.approx(gauss_cube(lambda x: u(x) * v(x)), rel=1e-12)
    assert U @ b == pytest.approx(gauss_cube(lambda x: poly3(x) * u(x)), rel=1e-12)
    right = mesh.locate_exterior_facets(lambda x: np.isclose(x[0], 1.0))
    Mf = dm.assemble_matrix(fem.form_facet_mass(V, right), none, algorithm=alg).to_scipy()
    bf = dm.assemble_vector(fem.form_facet_source(V, right, fem.FN_LINEAR), none, algorithm=alg).numpy()
    on_face = lambda g: (lambda s: g(np.stack([np.ones_like(s[0]), s[0], s[1]])))
    assert U @ (Mf @ W) == pytest.approx(gauss_cube(on_face(lambda x: u(x) * v(x)), dim=2), rel=1e-12)
    lin = lambda x: 1.0 + x[0] - 2.0 * x[1] + 0.5 * x[2]
    assert U @ bf == pytest.approx(gauss_cube(on_face(lambda x: lin(x) * u(x)), dim=2), rel=1e-12)


@pytest.mark.parametrize("degree", [1, 2])
def test_product_vector_valued_forms(degree):
    import dolfinx_mpc_amd as dm

    mesh = create_unit_cube(2, 2, 2)
    V = fem.functionspace(mesh, ("Lagrange", degree, (3,)))
    Q = fem.functionspace(mesh, ("Lagrange", 1))
    X = V.tabulate_dof_coordinates().T
    (u, gu), (v, gv) = u_fun(degree), v_fun(degree)
    Uv = np.stack([u(X), 2 * v(X), u(X) - v(X)], axis=1).reshape(-1)
    Wv = np.stack([v(X), -u(X), 0.5 * u(X)], axis=1).reshape(-1)
    none = _none(V)
    A = dm.assemble_matrix(fem.form_stiffness(V), none).to_scipy()
    exact = gauss_cube(lambda x: np.sum(gu(x) * gv(x), axis=0) - 2 * np.sum(gv(x) * gu(x), axis=0)
                       + 0.5 * np.sum((gu(x) - gv(x)) * gu(x), axis=0))
    assert Uv @ (A @ Wv) == pytest.approx(exact, rel=1e-12)
    for alg in ("rowblock", "atomic"):
        mu, lam = 1.3, 0.7
        E = dm.assemble_matrix(fem.form_elasticity(V, mu, lam), none, algorithm=alg).to_scipy()

        def integrand(x):
            GU = np.stack([gu(x), 2 * gv(x), gu(x) - gv(x)])
            GW = np.stack([gv(x), -gu(x), 0.5 * gu(x)])
            eU, eW = 0.5 * (GU + GU.transpose(1, 0, 2)), 0.5 * (GW + GW.transpose(1, 0, 2))
            return 2 * mu * np.sum(eU * eW, axis=(0, 1)) + lam * np.trace(GU) * np.trace(GW)

        assert Uv @ (E @ Wv) == pytest.approx(gauss_cube(integrand), rel=1e-12)
    if degree == 2:
        mq = _none(Q)
        A10 = dm.assemble_matrix(fem.form_div_trial(Q, V, constant=-1.0), (mq, none)).to_scipy()
        q = lambda x: 0.2 + x[0] - x[1] + 3.0 * x[2]
        Qh = q(Q.tabulate_dof_coordinates().T)
        divU = lambda x: gu(x)[0] + 2 * gv(x)[1] + gu(x)[2] - gv(x)[2]
        assert Qh @ (A10 @ Uv) == pytest.approx(-gauss_cube(lambda x: divU(x) * q(x)), rel=1e-12)
