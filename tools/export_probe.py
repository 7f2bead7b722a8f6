#!/usr/bin/env python3
"""Where does the export kernel's time go?  fq_sym_export on the metric tensor for every container, beside the streaming
ceilings of the same byte ratios (tools/kbench ... ceilings: shrink 2:1 / 4:1, copy).   python tools/export_probe.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    wl = bench.Workload(torch.device("cuda", 0))
    wl.prime_bounds()
    L, lib_, st = wl.L, wl._lib, wl.stream
    rows, cols, n = wl.rows, wl.cols, wl.n
    for s in wl.sets:
        s["bins"] = torch.empty(n * 2, dtype=torch.uint8, device=wl.device)
        s["scales"] = torch.empty(rows, 2, device=wl.device)
        s["over"] = torch.empty(rows, dtype=torch.int32, device=wl.device)
    out = {}
    for key, bits in (("w", 4), ("w", 8), ("a", 8)):
        for cname, cont, bpe in (("int4", lib_.BINS_INT4, 0.5), ("int8", lib_.BINS_INT8, 1), ("int16", lib_.BINS_INT16, 2)):
            for over in (True, False):
                def fn(s, key=key, bits=bits, cont=cont, over=over):
                    rc = L.fq_sym_export(s[key].data_ptr(), s["bins"].data_ptr(), s["scales"].data_ptr(), s["over"].data_ptr() if over else None, rows, cols, bits,
                                         cont, lib_.DTYPE_BF16, 0, 0, st)
                    assert rc == 0, rc
                ms, pct = wl.time_kernel(fn, 60)
                moved = n * (2 + bpe)
                ovf = int((wl.sets[0]["over"] != 0).sum().item()) if over else None
                out[f"{key}{bits}->{cname}{'' if over else ' (no overflow out)'}"] = {"us": round(ms * 1e3, 2), "GBs": round(moved / ms / 1e6, 1), "rows_with_overflow": ovf}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
