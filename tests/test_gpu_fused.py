"""mpcx_assemble_fused (include/mpcx.h): the matrix and vector cluster kernels of config 2 in one launch give the values of
the two separate launches.  The entry point is an experiment the product path does not take (3.72 ms
against 3.50 ms for the two launches on two streams); this keeps it honest."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [32, 48])
def test_fused_launch_matches_separate_launches(n):
    # a child process: the probe sets MPCX_VCUBE_ROWS before the library is imported
    code = (f"import sys; sys.path.insert(0, {os.path.join(ROOT, 'tools', 'probes')!r}); import fused_probe as p; "
            f"r = p.run({n}, timing=False); assert r is not None, 'plans differ'; "
            "assert r['dA'] < 1e-14 and r['db'] < 1e-13, r; print('ok', r)")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
