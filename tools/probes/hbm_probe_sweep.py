#!/usr/bin/env python
"""What this box's HBM streams (the library's probe reached 4.6-5.0 TB/s where the hardware guide quotes
6.29 TB/s for a float4 copy): the library's probes (mpcx_hbm_probe modes 0-4) over buffer sizes from inside the 256 MiB
Infinity Cache to 4 GiB, next to torch's device copy (hipMemcpyDtoD) and a few grid sizes.  Prints one JSON line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch

    from dolfinx_mpc_amd import _native

    L = _native.lib()

    def t(fn, reps=10):
        for _ in range(3):
            fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for s, e in ev:
            s.record()
            fn()
            e.record()
        torch.cuda.synchronize()
        return min(s.elapsed_time(e) for s, e in ev), sum(s.elapsed_time(e) for s, e in ev) / reps

    out = {"wgs": os.environ.get("MPCX_HBM_PROBE_WGS", "2048"), "sizes": {}}
    for mib in (64, 256, 1024, 2048, 4096):
        n = mib << 20
        a = torch.empty(n // 8, dtype=torch.float64, device="cuda").fill_(1.0)
        b = torch.empty_like(a)
        rec = {}
        for mode, name, moved in ((0, "copy", 2 * n), (3, "copy_x4", 2 * n), (4, "copy_x4_nt", 2 * n), (1, "read", n), (2, "write", n)):
            best, mean = t(lambda: L.mpcx_hbm_probe(a.data_ptr(), b.data_ptr(), n, mode, None))
            rec[name] = [round(moved / best / 1e6, 1), round(moved / mean / 1e6, 1)]  # GB/s (best, mean)
        best, mean = t(lambda: b.copy_(a))
        rec["torch_copy"] = [round(2 * n / best / 1e6, 1), round(2 * n / mean / 1e6, 1)]
        best, mean = t(lambda: b.zero_())
        rec["torch_memset"] = [round(n / best / 1e6, 1), round(n / mean / 1e6, 1)]
        out["sizes"][f"{mib} MiB"] = rec
        del a, b
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "grids":
        for g in (1024, 2048, 4096, 8192, 16384):
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, MPCX_HBM_PROBE_WGS=str(g)))
    else:
        main()
