#!/usr/bin/env python
"""Summarise a rocprofv3 (--kernel-trace [--pmc ...]) results .db into text:
per-kernel launch count / avg / min / max / total time, and per-kernel PMC
counter sums if counters were collected.  Usage:
    python tools/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/rNN_name.txt
"""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    print(f"# rocprofv3 summary of {path}")
    rows = cur.execute(
        "select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
        "max(grid_x), max(workgroup_x) from kernels group by name order by 6 desc").fetchall()
    tot = sum(r[5] for r in rows) or 1
    print(f"{'kernel':100s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>10s} {'%':>6s} "
          f"{'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>7s} {'scratch':>7s} {'grid':>10s} {'wg':>5s}")
    for r in rows:
        print(f"{r[0][:100]:100s} {r[1]:6d} {r[2] / 1e3:10.1f} {r[3] / 1e3:10.1f} {r[4] / 1e3:10.1f} {r[5] / 1e6:10.3f} "
              f"{100 * r[5] / tot:6.2f} {r[6]:5d} {r[7]:5d} {r[8]:5d} {r[9]:7d} {r[10]:7d} {r[11]:10d} {r[12]:5d}")
    try:
        pm = cur.execute(
            "select k.name, p.counter_name, count(*), sum(p.value), avg(p.value) from pmc_events p "
            "join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name order by k.name").fetchall()
    except sqlite3.Error as e:
        pm = []
        print(f"# no pmc data ({e})")
    if pm:
        print("\n# PMC counters (per kernel: dispatches, sum, avg per dispatch)")
        for r in pm:
            print(f"{r[0][:90]:90s} {r[1]:28s} n={r[2]:5d} sum={r[3]:.6g} avg={r[4]:.6g}")


if __name__ == "__main__":
    main(sys.argv[1])
