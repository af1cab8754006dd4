"""Short view of a bench.py line: step, timings, one-shot costs, per-kernel roofline entries."""
import json
import sys

for path in sys.argv[1:]:
    d = json.load(open(path))
    print(path, "ms/step %.3f  value %.4g %s" % (d["ms_per_step"], d["value"], d["unit"]))
    print("  timings", {k: round(v, 3) for k, v in d.get("timings_ms", {}).items()})
    o = d.get("one_shot", {})
    print("  one_shot", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in o.items() if k != "note"})
    for k in d.get("roofline_kernels", []):
        print("  ", k["kernel"], "%.3f ms" % k["launch_ms"], "hbm %.3f" % k["hbm_frac"], "fp64 %.3f" % k.get("fp64_frac", 0.0), k.get("bound"))
    r = d.get("roofline", {})
    print("  roofline", r.get("kernel"), r.get("bound"), "frac %.3f" % r.get("frac", 0.0), "traffic", r.get("traffic"), "alg", r.get("algorithmic_bytes"))
    g = d.get("roofline_generic")
    if g:
        print("  generic %.3f ms" % g["ms_per_step"], [(k["kernel"], round(k["launch_ms"], 3)) for k in g["kernels"]])
