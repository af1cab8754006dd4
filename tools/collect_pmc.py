#!/usr/bin/env python
"""Collect rocprofv3 PMC counters for the config-2 hot kernels, one counter group
per pass (FETCH_SIZE and WRITE_SIZE cannot share a pass: TCC has 4 slots,
MI355X_MICROARCH.md "rocprofv3 PMC slots"), and write

    <out>/pmc_summary.json   per kernel: avg counter value per dispatch (+ derived HBM bytes)
    <out>/pmc_summary.txt    human-readable table

Run on the GPU box:   python tools/collect_pmc.py gpurun_out/pmc_rNN [N] [config]
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE
reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section);
WRITE_SIZE matched the known byte count of our coalesced stores 1:1 (2.04 GB of
CSR values per launch).
"""
import glob
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [
    "FETCH_SIZE",
    "WRITE_SIZE",
    "TCC_HIT_sum TCC_MISS_sum",
    "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS",
    "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU",
    "GRBM_GUI_ACTIVE",
]


def main():
    out = os.path.abspath(sys.argv[1])
    N = sys.argv[2] if len(sys.argv) > 2 else "256"
    global CONFIG
    CONFIG = sys.argv[3] if len(sys.argv) > 3 else "2"
    extra = sys.argv[4:]  # further bench.py options, e.g. --cell hex
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    for g in GROUPS:
        name = g.split()[0]
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + g.split() + ["-d", out, "-o", "p_" + name, "--",
                                                                      sys.executable, os.path.join(ROOT, "tools", "profile_kernels.py"), N, CONFIG] + extra
        with open(os.path.join(out, f"log_{name}.txt"), "w") as fh:
            try:
                subprocess.run(cmd, stdout=fh, stderr=subprocess.STDOUT, env=env, cwd="/tmp", timeout=240)
            except subprocess.TimeoutExpired:
                fh.write("\nTIMEOUT\n")
    summary = {}
    for f in sorted(glob.glob(os.path.join(out, "p_*_results.db"))):
        cur = sqlite3.connect(f).cursor()
        rows = cur.execute(
            "select k.name, p.counter_name, count(distinct p.dispatch_id), sum(p.counter_value), avg(k.end - k.start) "
            "from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name").fetchall()
        for name, counter, ndisp, total, dur in rows:
            d = summary.setdefault(name, {})
            d[counter] = total / ndisp  # summed over SEs/XCDs, per dispatch
            d.setdefault("avg_duration_us_profiled", dur / 1e3)
    for name, d in summary.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_bytes_per_launch"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
    summary["_workload_n"] = int(N)
    json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
    summary.pop("_workload_n")
    with open(os.path.join(out, "pmc_summary.txt"), "w") as fh:
        fh.write(f"# rocprofv3 PMC summary, tools/profile_kernels.py N={N}; values = sum over SEs per dispatch\n")
        for name, d in sorted(summary.items(), key=lambda kv: -kv[1].get("avg_duration_us_profiled", 0)):
            fh.write(f"\n{name}\n")
            for k, v in sorted(d.items()):
                fh.write(f"    {k:32s} {v:.6g}\n")
    print(open(os.path.join(out, "pmc_summary.txt")).read())
    # the rocprofv3 databases are large (hundreds of MB at full size) and gpurun_out/ only travels back below 64 MiB:
    # the summaries are what is kept
    for f in glob.glob(os.path.join(out, "p_*")):
        try:
            os.remove(f)
        except OSError:
            pass


if __name__ == "__main__":
    main()
