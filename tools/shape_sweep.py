#!/usr/bin/env python3
"""Kernel throughput over the model shapes SURVEY §8(d) lists (LLaMA-7B / 13B weights, activations, KV; tiny config),
bf16 and fp32, Sym and Asym, forward and backward -- raw C-ABI launches, HIP events, rotating buffers.

    python tools/shape_sweep.py      -> gpurun_out/shape_sweep.json (+ table on stdout)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from llm_qat_amd import _lib  # noqa: E402

SHAPES = [
    ("7B down_proj.weight", 4096, 11008, 4, "w"), ("7B gate/up.weight", 11008, 4096, 4, "w"), ("7B attn.weight", 4096, 4096, 4, "w"),
    ("7B act [2048,4096]", 2048, 4096, 8, "a"), ("7B act [2048,11008]", 2048, 11008, 8, "a"), ("7B KV [2048,4096] b4", 2048, 4096, 4, "a"),
    ("13B down_proj.weight", 5120, 13824, 4, "w"), ("13B gate/up.weight", 13824, 5120, 4, "w"), ("13B act [2048,5120]", 2048, 5120, 8, "a"),
    ("13B act [2048,13824]", 2048, 13824, 8, "a"), ("tiny weight [688,256]", 688, 256, 8, "w"), ("tiny act [256,256]", 256, 256, 8, "a"),
]


def main():
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for dtype, code, esz in ((torch.bfloat16, _lib.DTYPE_BF16, 2), (torch.float32, _lib.DTYPE_F32, 4), (torch.float16, _lib.DTYPE_F16, 2)):
        for label, rows, cols, bits, style in (SHAPES if dtype != torch.float16 else SHAPES[:6]):
            n = rows * cols
            nsets = max(2, min(8, int(600e6 // (n * esz * 4)) + 1))   # > 256 MiB of distinct buffers when possible
            g = torch.Generator(device=dev).manual_seed(1)
            sets = []
            for _ in range(nsets):
                x = torch.randn(rows, cols, generator=g, device=dev) * (0.02 if style == "w" else 1.0)
                if style == "a":
                    x[torch.rand(rows, cols, generator=g, device=dev) < 1e-3] *= 20
                x = x.to(dtype)
                mb = L.fq_ste_mask_bytes(rows, cols, code)
                sets.append(dict(x=x, y=torch.empty_like(x), g=torch.randn_like(x), gx=torch.empty_like(x),
                                 b=torch.empty(rows, 2, device=dev), m=torch.empty(max(mb, 8), dtype=torch.uint8, device=dev), mb=mb))

            def t(fn, iters=60, rounds=3):
                # Every kind gets its own warm-up over ALL buffer sets and the median of several rounds: the first batch
                # that touches freshly allocated buffers runs 30-50 % slow on small tensors (tools/small_shape_probe.py,
                # profiles/r02_small_shape_probe.json: [2048,4096] 10.3 us in the first round, 6.7 us afterwards), which
                # round 1's sweep (5 warm-up launches, Sym timed first) reported as if it were the Sym kernel's time.
                for i in range(max(20, 2 * nsets)):
                    fn(sets[i % nsets])
                vals = []
                for _ in range(rounds):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    for i in range(iters):
                        fn(sets[i % nsets])
                    e1.record()
                    torch.cuda.synchronize()
                    vals.append(e0.elapsed_time(e1) / iters * 1e3)
                return sorted(vals)[len(vals) // 2]

            def chk(rc):
                if rc:
                    _lib.check(rc, "sweep")

            for kind in ("sym", "asym"):
                fwd = L.fq_sym_fwd_train if kind == "sym" else L.fq_asym_fwd_train
                us_f = t(lambda s: chk(fwd(s["x"].data_ptr(), s["y"].data_ptr(), rows, cols, bits, code, 0, -2.0, 2.0, s["b"].data_ptr(),
                                           s["m"].data_ptr(), s["mb"], st)))
                us_b = t(lambda s: chk(L.fq_ste_bwd_mask(s["g"].data_ptr(), s["gx"].data_ptr(), rows, cols, -2.0, 2.0, s["b"].data_ptr(),
                                                         s["m"].data_ptr(), s["mb"], code, st)))
                us_bx = t(lambda s: chk(L.fq_ste_bwd(s["g"].data_ptr(), s["x"].data_ptr(), s["gx"].data_ptr(), n, -2.0, 2.0, code, st)))
                us_acn = us_acw = None
                if kind == "sym" and esz == 2:   # the reference's arithmetic under torch.autocast (fp32 behind the reciprocal)
                    y32 = torch.empty(rows, cols, device=dev)
                    us_acn = t(lambda s: chk(L.fq_sym_fwd_autocast(s["x"].data_ptr(), s["y"].data_ptr(), rows, cols, bits, code, 1, 0, -2.0, 2.0,
                                                                    s["b"].data_ptr(), s["m"].data_ptr() if s["mb"] else None, s["mb"], None, 0, st)))
                    us_acw = t(lambda s: chk(L.fq_sym_fwd_autocast(s["x"].data_ptr(), y32.data_ptr(), rows, cols, bits, code, 1, 1, -2.0, 2.0,
                                                                    s["b"].data_ptr(), None, 0, None, 0, st)))
                row = dict(shape=label, dtype=str(dtype).split(".")[-1], kind=kind, bits=bits, elems=n, fwd_us=round(us_f, 2),
                           bwd_mask_us=round(us_b, 2), bwd_xread_us=round(us_bx, 2),
                           fwd_autocast_narrow_us=None if us_acn is None else round(us_acn, 2),
                           fwd_autocast_wide_us=None if us_acw is None else round(us_acw, 2),
                           fwd_gbs=round(n * 2 * esz / us_f / 1e3, 1), fwd_bwd_gelems=round(n / (us_f + us_b) / 1e3, 1))
                out.append(row)
                print(f"{label:26s} {row['dtype']:8s} {kind:4s} b{bits}  fwd {us_f:7.2f} us ({row['fwd_gbs']:7.1f} GB/s)  "
                      f"bwd(mask) {us_b:7.2f} us  bwd(x) {us_bx:7.2f} us  fwd+bwd {row['fwd_bwd_gelems']:6.1f} Gelem/s"
                      + (f"  | autocast fwd narrow {us_acn:7.2f} wide {us_acw:7.2f} us" if us_acn is not None else ""), flush=True)
            del sets
            torch.cuda.empty_cache()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "shape_sweep.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
