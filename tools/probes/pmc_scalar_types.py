#!/usr/bin/env python
"""SQ counters of the scalar-type row-block kernels (tools/bench_scalar_types.py N): one rocprofv3 --pmc pass per counter group,
a per-kernel table on stdout.  python tools/probes/pmc_scalar_types.py gpurun_out/pmc_sc [N=128]"""
import glob
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GROUPS = ["SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS",
          "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU",
          "FETCH_SIZE", "WRITE_SIZE"]


def main():
    out = os.path.abspath(sys.argv[1])
    N = sys.argv[2] if len(sys.argv) > 2 else "128"
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    summary = {}
    for g in GROUPS:
        name = g.split()[0]
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + g.split() + ["-d", out, "-o", "p_" + name, "--", sys.executable,
                                                                      os.path.join(ROOT, "tools", "bench_scalar_types.py"), N]
        subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env, cwd="/tmp", timeout=400)
    for f in sorted(glob.glob(os.path.join(out, "p_*_results.db")) + glob.glob(os.path.join(out, "*", "p_*_results.db"))):
        cur = sqlite3.connect(f).cursor()
        rows = cur.execute("select k.name, p.counter_name, count(distinct p.dispatch_id), sum(p.counter_value), avg(k.end - k.start) "
                           "from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name").fetchall()
        for kname, counter, nd, total, dur in rows:
            if "rowblock" not in kname:
                continue
            d = summary.setdefault(kname, {})
            d[counter] = total / nd
            d["us"] = dur / 1e3
        os.remove(f)
    for kname, d in sorted(summary.items()):
        cyc = d["us"] * 1e-6 * 2.4e9
        line = {k: f"{v:.3g}" for k, v in d.items()}
        der = {}
        if "SQ_INSTS_VALU" in d:
            der["valu_issue"] = round(4 * d["SQ_INSTS_VALU"] / 1024 / cyc, 3)
        if "SQ_LDS_IDX_ACTIVE" in d:
            der["lds_busy"] = round(d["SQ_LDS_IDX_ACTIVE"] / 256 / cyc, 3)
            der["lds_conflict"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0) / 256 / cyc, 3)
        if "SQ_WAVE_CYCLES" in d:
            der["waiting"] = round(d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], 3)
            der["waves_per_simd"] = round(d["SQ_WAVE_CYCLES"] * 4 / 1024 / cyc, 2)
        if "FETCH_SIZE" in d:
            der["hbm_GB"] = round((2 * d["FETCH_SIZE"] + d.get("WRITE_SIZE", 0)) * 1024 / 1e9, 3)
        print(kname[:150])
        print("   ", line)
        print("   ", der)


if __name__ == "__main__":
    main()
