"""The example scripts (reference-demo shaped drivers) run end to end on the GPU and meet their own checks."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "examples", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("N,degree", [(12, 1), (8, 2)])
def test_demo_periodic_poisson(N, degree):
    info = _load("demo_periodic_poisson").main(N, degree, verbose=False)
    assert info["converged"] and info["slaves"] > 0 and info["periodic_gap"] < 1e-13 and info["u_max"] > 1e-3


def test_demo_stokes_nest():
    info = _load("demo_stokes_nest").main(4, verbose=False)
    assert info["converged"] and info["velocity_error"] < 1e-8 and info["pressure_ptp"] < 1e-6
