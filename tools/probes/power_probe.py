#!/usr/bin/env python
"""Is the step of config 5 the SUM of its matrix and vector kernels because the package is power-limited?  Socket power
and shader clock (hwmon sysfs, sampled every 20 ms by a thread; rocm-smi as fall-back) while the GPU runs, for ~2 s each:
the matrix assembly alone, the vector assembly alone, both on their two streams (the benchmark step), and both with the
first part of the matrix launch capped so that the two kernels are co-resident (MPCX_CORUN=1).

    python tools/probes/power_probe.py --config 5
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.power = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")
                            + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))
        self.freq = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
        self.samples, self.on, self.stop = [], False, False

    def read(self):
        p = f = None
        try:
            if self.power:
                p = max(int(open(x).read()) for x in self.power) / 1e6
            if self.freq:
                f = max(int(open(x).read()) for x in self.freq) / 1e9
        except (OSError, ValueError):
            pass
        return p, f

    def run(self):
        while not self.stop:
            if self.on:
                self.samples.append(self.read())
            time.sleep(0.02)


def smi_once():
    try:
        r = subprocess.run(["rocm-smi", "-P", "-c", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=20)
        return json.loads(r.stdout)
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=5)
    ap.add_argument("--size", dest="n", type=int, default=0)
    ap.add_argument("--seconds", type=float, default=2.0)
    pa = ap.parse_args()
    import torch

    import bench
    import dolfinx_mpc_amd as dm
    from dolfinx_mpc_amd.la import create_vector, wait_assembly

    args = argparse.Namespace(n=pa.n or {2: 256, 3: 128, 4: 56, 5: 246}[pa.config], no_tile=False, tile=[8, 8, 8], scaling="strong",
                              cell="tet", numbering="tiled", ufcx=None, config=pa.config)
    w = bench.poisson_workload(args, 0, 1, 1 if pa.config == 2 else 2) if pa.config in (2, 5) else (
        bench.stokes_workload(args, 0, 1) if pa.config == 3 else bench.contact_workload(args, 0, 1))
    mats = {label: dm.create_matrix(f, m0, m1) for label, f, (m0, m1) in w.blocks}
    vecs = {label: create_vector(m.function_space) for label, _f, m in w.vectors}

    def mat():
        for label, f, (m0, m1) in w.blocks:
            dm.assemble_matrix(f, (m0, m1), bcs=w.bcs, A=mats[label], algorithm="rowblock")

    def vec():
        for label, f, m in w.vectors:
            dm.assemble_vector(f, m, b=vecs[label])

    def both():
        mat()
        vec()

    s = Sampler()
    s.start()
    print("# hwmon files:", s.power, s.freq, flush=True)
    print("# rocm-smi idle:", json.dumps(smi_once())[:600], flush=True)
    os.environ["MPCX_CORUN"] = "0"
    for _ in range(3):
        both()
    torch.cuda.synchronize()
    big = torch.empty(1 << 28, dtype=torch.float64, device="cuda")
    big2 = torch.empty_like(big)
    small = torch.empty(1 << 20, dtype=torch.float64, device="cuda")

    def copy():
        big2.copy_(big)

    def light():
        for _ in range(50):
            small.add_(1.0)

    try:
        r = subprocess.run(["rocm-smi", "--showmaxpower", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=20)
        print("# rocm-smi --showmaxpower:", r.stdout.strip()[:400], flush=True)
    except Exception as e:  # noqa: BLE001
        print("# rocm-smi --showmaxpower failed:", e, flush=True)
    arms = [("device copy 2 GiB (HBM streaming only)", copy, {"MPCX_CORUN": "0"}), ("tiny kernels (launch-bound)", light, {"MPCX_CORUN": "0"}),
            ("matrix alone", mat, {"MPCX_CORUN": "0"}), ("vector alone", vec, {"MPCX_CORUN": "0"}),
            ("both, two streams", both, {"MPCX_CORUN": "0"}),
            ("both, matrix capped (co-resident)", both, {"MPCX_CORUN": "1", "MPCX_CORUN_FRAC": "0.999", "MPCX_CORUN_MATRIX_WGS": "3"})]
    av = sys.modules["dolfinx_mpc_amd.assemble_vector"]
    for name, fn, env in arms:
        os.environ.update(env)
        av.VECTOR_OWNER_ROWS = 3072 if env.get("MPCX_CORUN") == "1" else 8192
        for _ in range(3):
            fn()
        wait_assembly()
        torch.cuda.synchronize()
        s.samples, s.on = [], True
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < pa.seconds:
            for _ in range(5):
                fn()
            n += 5
            wait_assembly()
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        s.on = False
        ps = [p for p, _ in s.samples if p is not None]
        fs = [f for _, f in s.samples if f is not None]
        smi = smi_once() if not ps else None
        print("ARM " + json.dumps({"arm": name, "ms_per_call": dt / n * 1e3, "samples": len(s.samples),
                                   "power_W_mean": sum(ps) / len(ps) if ps else None, "power_W_max": max(ps) if ps else None,
                                   "sclk_GHz_mean": sum(fs) / len(fs) if fs else None, "sclk_GHz_min": min(fs) if fs else None,
                                   "rocm_smi_after": smi}), flush=True)
    s.stop = True


if __name__ == "__main__":
    main()
