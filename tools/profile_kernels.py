#!/usr/bin/env python
"""Run a config's hot kernels a few times, for rocprofv3 (--kernel-trace / --pmc):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o name -- python tools/profile_kernels.py [N] [config] [more bench.py options]
(bench.py in its child mode: set-up, first call, two more steps, no timing or baselines)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = sys.argv[1] if len(sys.argv) > 1 else "256"
config = sys.argv[2] if len(sys.argv) > 2 else "2"
extra = sys.argv[3:]  # e.g. --cell hex
sys.exit(subprocess.call([sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--size", N, "--steps", "1",
                          "--warmup", "0", "--no-cpu-baseline", "--no-traffic"] + extra, env=dict(os.environ, MPCX_BENCH_CHILD="1")))
