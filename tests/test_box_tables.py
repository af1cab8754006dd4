"""Host checks of the tensor-grid tables of the right-hand side (no GPU): the generated header of the 14-point cluster path is
what its generator writes and reproduces the 84 points of a box; the data-driven (point, vertex subset) table of the per-cell
path reproduces the quadrature points of cells of a box for every rule the library uses."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dolfinx_mpc_amd.quadrature import make_quadrature  # noqa: E402

FAN = [[0, 1, 3, 7], [0, 1, 7, 5], [0, 5, 7, 4], [0, 3, 2, 7], [0, 6, 4, 7], [0, 2, 6, 7]]  # csrc/mpcx_fan.hpp


def _header_arrays():
    text = open(os.path.join(ROOT, "dolfinx_mpc_amd", "csrc", "mpcx_box14.hpp")).read()

    def floats(name):
        body = re.search(name + r"\[[^\]]*\](?:\[[^\]]*\])? = \{(.*?)\};", text, re.S).group(1)
        return np.array([float.fromhex(t) for t in re.findall(r"-?0x[0-9a-fA-F.]+p[-+]?\d+", body)])

    def ints(name):
        body = re.search(name + r"(?:\[[^\]]*\])+ = \{(.*?)\};", text, re.S).group(1)
        return np.array([int(t) for t in re.findall(r"\d+", body)])

    return text, floats, ints


def test_generated_header_is_current(tmp_path):
    """tools/gen_box14.py writes exactly the committed header (the rule or the fan changed without regenerating it otherwise)"""
    text = open(os.path.join(ROOT, "dolfinx_mpc_amd", "csrc", "mpcx_box14.hpp")).read()
    src = open(os.path.join(ROOT, "tools", "gen_box14.py")).read().replace(
        'os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dolfinx_mpc_amd", "csrc", "mpcx_box14.hpp")',
        repr(str(tmp_path / "out.hpp")))
    script = tmp_path / "gen.py"
    script.write_text(src.replace("os.path.dirname(os.path.dirname(os.path.abspath(__file__)))", repr(ROOT)))
    subprocess.run([sys.executable, str(script)], check=True, capture_output=True)
    assert (tmp_path / "out.hpp").read_text() == text


def test_cluster_tables_reproduce_the_points_of_a_box():
    _text, floats, ints = _header_arrays()
    grid, xq, wq, wu, lam, wl = floats("GRID"), floats("XQ").reshape(14, 3), floats("WQ"), floats("WU"), floats("LAM"), None
    idx = ints("IDX").reshape(6, 14, 3)
    pts, wts = make_quadrature("tetrahedron", 5)
    assert np.array_equal(xq, pts) and np.array_equal(wq, wts)  # bit for bit: the kernel compares
    corner = np.array([[(v >> d) & 1 for d in range(3)] for v in range(8)], dtype=float)
    for t, verts in enumerate(FAN):
        for q, p in enumerate(pts):
            bary = np.array([1.0 - p.sum(), *p])
            assert np.abs(grid[idx[t, q]] - bary @ corner[verts]).max() < 1e-15
    assert grid.size == 19 and np.all(np.diff(grid) > 0) and np.abs(grid + grid[::-1] - 1.0).max() < 1e-15  # symmetric about 1/2
    widx, lidx = ints("WIDX"), ints("LIDX").reshape(14, 3)
    assert np.array_equal(wu[widx], wts) and np.array_equal(lam[lidx], pts)
    # weight x barycentric weight of every (point, vertex): the constants of the table kernel's accumulation
    tables = re.search(r"constexpr Tables TABLES = \{(.*)\};", _text, re.S).group(1)
    groups = [np.array([float.fromhex(t) for t in re.findall(r"-?0x[0-9a-fA-F.]+p[-+]?\d+", g)]) for g in re.findall(r"\{([^{}]*)\}", tables)]
    wl = groups[3]
    wlidx = ints("WLIDX").reshape(14, 4)
    bary = np.concatenate([1.0 - pts.sum(axis=1, keepdims=True), pts], axis=1)
    assert np.abs(wl[wlidx] - wts[:, None] * bary).max() < 1e-16
    assert abs((wl[wlidx]).sum() - 1.0 / 6.0) < 1e-15


@pytest.mark.parametrize("degree", [1, 2, 5, 6, 8])
def test_subset_table_reproduces_the_points_of_box_cells(degree):
    """rule_subset_table on every tetrahedron rule the library hands out: for random cells with their vertices on two values per
    axis (cells of boxes: Kuhn tetrahedra and the central tetrahedron of a five-cell cut) the coordinates lo + h eta[J[q][m]] are
    the affine images of the rule's points"""
    from dolfinx_mpc_amd.assemble_vector import rule_subset_table

    pts, _w = make_quadrature("tetrahedron", degree)
    table = rule_subset_table(pts)
    if table is None:  # collapsed Gauss rules of high degree: more than 250 distinct sums, the per-point evaluation stays
        assert degree > 6
        return
    eta, J = table
    assert eta.size <= 250 and np.all(np.diff(eta) > 0)
    rng = np.random.default_rng(degree)
    cells = [np.array(c, dtype=float) for c in ([[0, 0, 0], [1, 0, 0], [1, 1, 0], [1, 1, 1]], [[0, 0, 0], [1, 1, 0], [1, 0, 1], [0, 1, 1]],
                                                [[1, 1, 1], [0, 1, 1], [0, 0, 1], [0, 0, 0]], [[0, 1, 0], [0, 0, 0], [1, 0, 1], [1, 1, 1]])]
    for c in cells:
        lo, h = rng.uniform(-1, 1, 3), rng.uniform(0.1, 2.0, 3)
        X = lo + h * c[rng.permutation(4)]
        low, high = X.min(axis=0), X.max(axis=0)
        m = [(int(sum(1 << v for v in range(4) if X[v, d] == high[d]))) for d in range(3)]
        for q, p in enumerate(pts):
            x = X[0] + (X[1:] - X[0]).T @ p
            got = np.array([low[d] + (high[d] - low[d]) * eta[J[q, m[d]]] for d in range(3)])
            assert np.abs(got - x).max() < 1e-14
