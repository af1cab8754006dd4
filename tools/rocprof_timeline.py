#!/usr/bin/env python
"""Timeline of the last kernels of a rocprofv3 --kernel-trace results .db: start offset, duration, queue, name -- to see which
launches of a step overlap.  Usage: python tools/rocprof_timeline.py x_results.db [how_many=40] [min_us=20] [anchor: show the kernels up to the last one whose name contains this]"""
import sqlite3
import sys


def main(path, count=40, min_us=20.0, anchor=None):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    s = "stream_id" if "stream_id" in cols else "0"
    rows = cur.execute(f"select start, end, {q}, {s}, name from kernels where end - start >= ? order by start", (min_us * 1e3,)).fetchall()
    if anchor:
        anchor, _, nth = anchor.partition("#")  # "name#k": the k-th occurrence instead of the last
        hits = [i for i, r in enumerate(rows) if anchor in r[4]]
        last = (hits[int(nth)] if nth else hits[-1]) if hits else len(rows) - 1
        rows = rows[:last + 3]
    rows = rows[-count:]
    t0 = rows[0][0]
    print(f"# {path}: last {len(rows)} kernels of at least {min_us} us (columns: {cols})")
    print(f"{'start_ms':>10s} {'end_ms':>10s} {'dur_ms':>8s} {'queue':>6s} {'stream':>7s}  kernel")
    for st, en, qu, sm, name in rows:
        print(f"{(st - t0) / 1e6:10.3f} {(en - t0) / 1e6:10.3f} {(en - st) / 1e6:8.3f} {qu!s:>6s} {sm!s:>7s}  {name[:90]}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40, float(sys.argv[3]) if len(sys.argv) > 3 else 20.0,
         sys.argv[4] if len(sys.argv) > 4 else None)
