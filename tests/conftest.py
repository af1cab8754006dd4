import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the host set-up routines live in the same shared library as the kernels: build it once if a fresh
    # checkout has not run __graft_entry__.build() yet (hipcc cross-compiles gfx950 without a GPU)
    lib = os.path.join(ROOT, "dolfinx_mpc_amd", "libmpcx.so")
    if not os.path.exists(lib):
        import __graft_entry__

        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure only)."""
    from oracle import pyoracle

    pyoracle.lib()
    return pyoracle
