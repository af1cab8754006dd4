"""bench.py's command line: `--gpus N` launches N ranks itself when no launcher did, and the
N > 1 line carries what the driver needs (n_gpus, rccl_ranks, cells per rank)."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_launches_the_ranks_itself(monkeypatch):
    """without WORLD_SIZE, `bench.py --gpus 2` re-executes itself through torch.distributed.run on 127.0.0.1 with
    two ranks; with fewer visible devices than ranks the transport falls back to gloo"""
    sys.path.insert(0, ROOT)
    import bench

    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("MPCX_DIST_BACKEND", raising=False)
    assert bench.launch_ranks(2) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "2", "--steps", "3"] and cmd[-5].endswith("bench.py")
    import torch

    if torch.cuda.device_count() < 2:
        assert seen["env"]["MPCX_DIST_BACKEND"] == "gloo"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


@pytest.mark.gpu
def test_bench_gpus_2_runs_two_ranks_without_a_launcher():
    """`python bench.py --gpus 2` on whatever is there (one GPU: both ranks share it over gloo): ONE JSON line from
    rank 0 with n_gpus = 2, both ranks in the process group, the cells of both slabs adding up to the mesh"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--size", "16", "--no-cpu-baseline", "--no-traffic"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2
    assert sum(out["config"]["cells_per_gpu"]) == 6 * 16 ** 3
    assert out["config"]["dofs_global"] == 17 ** 3 and out["value"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("nranks, extra", [(4, ["--config", "2", "--size", "48", "--scaling", "strong"]),
                                           (2, ["--config", "2", "--size", "40", "--scaling", "weak"]),
                                           (2, ["--config", "4", "--size", "16"]),
                                           (2, ["--config", "5", "--size", "24"])])
def test_multirank_path_of_every_partitioned_config(nranks, extra):
    """tools/multirank_smoke.sh inside the suite: bench.py's N > 1 path -- launched the way the driver
    launches it (torch.distributed.run, one rank per 'GPU') -- for configs 2, 4 and 5 and both scalings on whatever devices
    are there (one GPU: the ranks share it, transport gloo with host staging; RCCL refuses two ranks on one device).  One JSON
    line from rank 0, all ranks in the group, the pre-flight exchange passed, the global dof count of the stated problem."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch

    if torch.cuda.device_count() < nranks:
        env["MPCX_DIST_BACKEND"] = "gloo"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--steps", "3", "--warmup", "1",
           "--no-traffic", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert "error" not in out and out["n_gpus"] == nranks and out["rccl_ranks"] == nranks and out["value"] > 0
    assert "pre-flight over" in r.stderr
    assert out["transport"] is not None
