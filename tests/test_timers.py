"""The reference's named timer scopes round the library's entry points (dolfinx_mpc_amd/common.py): same names as
cpp/assemble_matrix.cpp:677, python/src/dolfinx_mpc/assemble_vector.py:47,99, cpp/utils.h:149,388."""
import pytest

import dolfinx_mpc_amd as dm
from dolfinx_mpc_amd import common


def test_timer_registry_and_table():
    common.reset_timings()
    with common.Timer("~MPC: unit test scope"):
        pass
    t = common.Timer("~MPC: unit test scope")
    t.start()
    assert t.elapsed()[0] >= 0.0
    assert t.stop() >= 0.0
    n, wall = common.timing("~MPC: unit test scope")
    assert n == 2 and wall >= 0.0
    text = common.list_timings()
    assert "~MPC: unit test scope" in text and "reps" in text
    with pytest.raises(KeyError):
        common.timing("no such scope")
    common.reset_timings()
    assert common.timings() == {}


def test_entry_points_carry_the_reference_scope_names():
    """(without a GPU the calls raise before doing anything -- the scope is still entered and recorded)"""
    from problems import case_cube_periodic, product_mpc

    common.reset_timings()
    case = case_cube_periodic(2, 1, 0.0)
    mpc = product_mpc(case)
    for fn, name in ((lambda: dm.assemble_matrix(case.a, mpc, bcs=case.bcs), "~MPC: Assemble matrix (C++)"),
                     (lambda: dm.assemble_vector(case.L, mpc), "~MPC: Assemble vector (C++)"),
                     (lambda: dm.create_sparsity_pattern(case.a, mpc, where="host"), "~MPC: Create sparsity pattern")):
        try:
            fn()
        except Exception:  # noqa: BLE001  (no device here)
            pass
        assert common.timing(name)[0] >= 1


@pytest.mark.gpu
def test_scopes_on_the_device():
    from dolfinx_mpc_amd.la import create_vector
    from problems import case_cube_periodic, product_mpc

    common.reset_timings()
    case = case_cube_periodic(4, 1, 0.3)
    mpc = product_mpc(case)
    A = dm.assemble_matrix(case.a, mpc, bcs=case.bcs)
    b = dm.assemble_vector(case.L, mpc)
    dm.apply_lifting(b, [case.a], [case.bcs], mpc)
    for name in ("~MPC: Assemble matrix (C++)", "~MPC: Assemble vector (C++)", "~MPC: Apply lifting (C++)", "~MPC: Create Matrix"):
        assert common.timing(name)[0] >= 1
    del A, create_vector


def test_twin_caches_die_with_the_callers_objects(monkeypatch):
    """the reordered twin (dolfinx_mpc_amd/locality.py) keeps twins of Forms / Functions / constraints only as long
    as the caller's objects live -- a time loop that rebuilds its forms every step does not accumulate them"""
    import gc

    from dolfinx_mpc_amd import fem, locality
    from problems import case_cube_periodic, product_mpc

    monkeypatch.setenv("MPCX_AUTO_REORDER", "1")
    case = case_cube_periodic(3, 1, 0.0, numbering="shuffled")
    mesh = case.V.mesh
    assert locality.wanted(mesh)
    try:
        tw = locality.Twin(mesh)
    except Exception:  # noqa: BLE001  (torch without a device takes the numpy branch; anything else is a real failure)
        raise
    V = case.V
    for _ in range(5):
        f = fem.Function(V)
        f.interpolate(lambda x: 1.0 + x[0])
        form = fem.form_stiffness(V, coefficient=f)
        tw.form(form)
        assert len(tw._forms) == 1 and len(tw._functions) == 1
        del form, f
        gc.collect()
        assert len(tw._forms) == 0 and len(tw._functions) == 0
    mpc = product_mpc(case)
    tw.mpc(mpc)
    assert len(tw._mpcs) == 1
    del mpc
    gc.collect()
    assert len(tw._mpcs) == 0
    assert len(tw._spaces) == 1  # the space lives on (case.V)
