#!/usr/bin/env python
"""Rows of DESIGN.md's section-5 table from a bench.py line (the driver-style record with its config3/4/5 sub-records):
    python tools/design_table.py profiles/r06_bench_default.json"""
import json
import sys


def kern(k):
    s = f"`{k['kernel']}` {k['launch_ms']:.2f} ms: hbm {k.get('hbm_frac', 0):.2f}"
    if k.get("fp64_frac") is not None:
        s += f" / fp64 {k['fp64_frac']:.2f}"
    if k.get("traffic_over_algorithmic") is not None:
        s += f", executed {k['traffic_over_algorithmic']:.2f} x = {k.get('frac_hbm_executed', 0):.2f} of 8 TB/s"
    if k.get("valu_issue_frac") is not None:
        s += f", VALU {k['valu_issue_frac']:.2f}"
    if k.get("value_storage") == "block-scalar":
        s += " (block-scalar storage)"
    return s


def row(name, d, ndofs=None):
    r = d.get("roofline", {})
    cells = [name, f"**{d['ms_per_step']:.2f}**", f"{d['value'] / 1e9:.2f} G"]
    extra = []
    if d.get("ms_per_step_csr_valued"):
        extra.append(f"CSR-valued {d['ms_per_step_csr_valued']:.1f} ms = {d['value_csr_valued'] / 1e9:.2f} G")
    if d.get("ms_per_step_graph"):
        extra.append(f"graph replay {d['ms_per_step_graph']:.2f} ms")
    u = d.get("roofline_ufcx_text") or {}
    if u.get("ms_per_step"):
        extra.append(f"imported FFCx-layout text {u['ms_per_step']:.2f} ms ({u['ms_per_step'] / d['ms_per_step']:.2f} x)")
    cells.append("; ".join(extra))
    cells.append("; ".join(kern(k) for k in d.get("roofline_kernels", [])))
    cells.append(f"`{r.get('kernel')}`: {r.get('bound')} {r.get('frac', 0):.2f} (counters: {r.get('bound_by_counters')}); step hbm {r.get('frac_hbm_step', 0):.2f}")
    return "| " + " | ".join(cells) + " |"


def main(path):
    d = json.loads(open(path).read().strip().splitlines()[-1])
    print("| run | step ms | DoFs/s | beside it | kernels (HIP-event launch ms; ALGORITHMIC fractions of 8 TB/s / 78.6 TF; executed traffic; VALU issue) | `roofline` |")
    print("|---|---|---|---|---|---|")
    print(row("config 2, default", d))
    for key, name in (("roofline_ufcx_text", "config 2, imported FFCx-layout files"), ("roofline_ufcx", "config 2, stated built-in twin"),
                      ("roofline_generic", "config 2, cluster kernels off"), ("roofline_spatial", "config 2, nodes and cells shuffled")):
        s = d.get(key)
        if s and s.get("ms_per_step"):
            t = s.get("timings_ms") or {}
            print(f"| {name} | **{s['ms_per_step']:.2f}** | {s['value'] / 1e9:.2f} G | " + ", ".join(f"{k} {v:.2f} ms" for k, v in t.items()) + " | | |")
    for c in (3, 4, 5):
        s = d.get(f"config{c}")
        if s and "ms_per_step" in s:
            print(row(f"config {c}", s))
    cb = d.get("cpu_baseline") or {}
    ca = d.get("cpu_baseline_allcores") or {}
    print("\nCPU:", cb.get("value"), cb.get("sample"), "| all cores:", ca.get("value"), ca.get("cores"))
    for c in (3, 4, 5):
        s = (d.get(f"config{c}") or {}).get("cpu_baseline") or {}
        print(f"config {c} CPU:", s.get("value"), s.get("sample"))
    print("one_shot:", json.dumps(d.get("one_shot", {}).get("first_call_split")), d.get("one_shot", {}).get("pattern_s"), "graph", d.get("ms_per_step_graph"))


if __name__ == "__main__":
    main(sys.argv[1])
