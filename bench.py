#!/usr/bin/env python3
"""bench.py -- fake-quant fwd+bwd throughput on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload at N=1 (BASELINE.json `metric`: "fake-quant fwd+bwd Gelem/s & achieved HBM GB/s,
4096x11008 bf16 W4A8"; it is the per-tensor hot path of configs[1], LLaMA-7B W4-A8-KV4):
one *step* is one pass of the hot path over one batch of synthetic tensors, i.e.

    W4 leg: SymQuantizer fwd + STE bwd on a weight-style  [4096, 11008] bf16 tensor (down_proj.weight)
    A8 leg: SymQuantizer fwd + STE bwd on an activation-style [4096, 11008] bf16 tensor

through the product's own autograd Functions' kernels (C ABI, current stream).  Inputs are resident
in HBM before the timed region; the step rotates over several buffer sets (> 256 MiB apart) so the
Infinity Cache cannot serve re-reads.  `value` = elements processed per second, whole job.

The op is per-tensor and does not shard ("replicas only"): with --gpus N every rank runs
the same step on its own GPU, no data-path collective; value = N * per-rank elements / max-over-ranks time.
`--gpus N` without a launcher (WORLD_SIZE unset) starts the N ranks itself, as fresh child processes, BEFORE
anything touches a GPU; under `python -m torch.distributed.run` it joins the ranks the launcher started.
A request for N ranks never reports fewer: if a rank is missing the run exits non-zero.

Output (round 3): stdout carries exactly ONE line, a compact headline (< 4 KB: the contract's keys + `roofline`, `roofline_step`,
`cpu_baseline`, `value_product_default`, `self_check`) -- the driver keeps an 8 KB stdout tail, and round 2's single 27 KB line was cut.
Every other family below is printed as its own STDERR line ({"bench_extras": <family>, "data": ...}) and the whole record is
written to bench_extras.json next to this file.
Families (every `frac` is a byte RATE the kernel sustained / 8 TB/s, never > 1):
  roofline      dominant kernel (the step's forward launch): algorithmic bytes / live HIP-event launch time vs 8 TB/s
  kernels_step  both launches of the step;  kernels  the four single-tensor launches
  kernels_model_shapes   the quantizer launches of one LLaMA-7B layer at their real shapes
  sustained     the headline step back to back for --sustain-seconds (default 5 s) after the timed region: the steady-state rate, and
                long enough for an outside utilisation sampler (the driver's gpu_busy, rocm-smi) to see the GPU at work
  autograd_path the same step through SymQuantizer.apply / backward (allocator + Python included)
  cpu_baseline  the reference's CPU path (eager op chain, oracle/eager_chain.py) timed on this box's host
                cores on a bounded sample -- rank 0, N=1 only
  gpu_eager     the reference's eager op chain run on this GPU (9+5 launches): the like-for-like "before"
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ROWS, COLS = 4096, 11008
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
FWD_BYTES_PER_ELEM = 4   # bf16: read x + write y              
BWD_BYTES_PER_ELEM = 6   # bf16: read g + read x + write gx    (the reference's data flow)
BWD_MASK_BYTES_PER_ELEM = 4  # bf16: read g + write gx: what the product's mask backward moves (+ 1 bit/elem of mask
#                              for tensors with clippable rows)


# ----------------------------------------------------------------------------------------------
# timing harness (device-agnostic so the N>1 aggregation is testable with gloo on CPU)
# ----------------------------------------------------------------------------------------------
def timed_region(step, steps, warmup, sync, dist_mod=None):
    """W untimed warm-up steps, then EXACTLY K steps bracketed by barrier+sync on both sides.
    Returns the max-over-ranks wall time in seconds."""
    for i in range(warmup):
        step(i)
    sync()
    if dist_mod is not None:
        dist_mod.barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    sync()
    if dist_mod is not None:
        dist_mod.barrier()
    sync()
    dt = time.perf_counter() - t0
    if dist_mod is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        if dist_mod.get_backend() == "nccl":
            t = t.cuda()
        dist_mod.all_reduce(t, op=dist_mod.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def sustained_region(step, seconds, sync, chunk=1000, clock=time.perf_counter):
    """The same step back to back for about `seconds` of wall time, in chunks of `chunk` steps with one device sync per chunk (the
    queue never drains inside a chunk).  NOT the headline -- K stays exactly --steps -- but it answers two things the 2-25 ms timed
    region cannot: whether the rate holds once clocks and temperatures settle, and it keeps the GPU busy for long enough that an
    outside observer's utilisation sampling (the driver's `gpu_busy`, rocm-smi) sees this process at all.
    Returns (steps_run, seconds_run, [ms_per_step of each chunk])."""
    if seconds <= 0:
        return 0, 0.0, []
    sync()
    chunks, n, t_start = [], 0, clock()
    while True:
        t0 = clock()
        for i in range(chunk):
            step(n + i)
        sync()
        t1 = clock()
        chunks.append((t1 - t0) / chunk * 1e3)
        n += chunk
        if t1 - t_start >= seconds:
            return n, t1 - t_start, chunks


def aggregate_value(elems_per_rank_step, steps, world, seconds):
    """whole-job throughput in Gelem/s: all ranks' elements / max-over-ranks time"""
    return elems_per_rank_step * steps * world / seconds / 1e9


# ----------------------------------------------------------------------------------------------
# GPU workload
# ----------------------------------------------------------------------------------------------
def _act_like(torch, rows, cols, gen, device):
    """activation-style input of BASELINE.md §3: N(0,1) with 0.1 % of the entries x20 (outlier channels)"""
    a = torch.randn(rows, cols, generator=gen, device=device)
    a[torch.rand(rows, cols, generator=gen, device=device) < 1e-3] *= 20.0
    return a.bfloat16()


class Workload:
    def __init__(self, device, rows=ROWS, cols=COLS, nsets=4, seed=1234):
        import torch
        import llm_qat_amd
        from llm_qat_amd import _lib
        _lib.lib()  # no fallback: raises if the HIP library is absent
        self.torch, self.ops = torch, llm_qat_amd.ops
        self.device = device
        self.rows, self.cols, self.nsets = rows, cols, nsets
        self.n = rows * cols
        self.sets = []
        for k in range(nsets):
            # BASELINE.md §3 inputs: W ~ N(0, 0.02^2) seed 1234, g ~ N(0,1)*1e-3 seed 1235, activations ~ N(0,1) with
            # 0.1 % of entries x20 seed 1236 (set 0 uses exactly those seeds; the rotating sets offset them by 1000*k)
            gw_, gg_, ga_ = (torch.Generator(device=device).manual_seed(sd + 1000 * k) for sd in (seed, seed + 1, seed + 2))
            w = (torch.randn(rows, cols, generator=gw_, device=device) * 0.02).bfloat16()
            a = _act_like(torch, rows, cols, ga_, device)
            gw = (torch.randn(rows, cols, generator=gg_, device=device) * 1e-3).bfloat16()
            ga = (torch.randn(rows, cols, generator=gg_, device=device) * 1e-3).bfloat16()
            L = _lib.lib()
            self.mask_bytes = L.fq_ste_mask_bytes(rows, cols, _lib.DTYPE_BF16)
            self.sets.append(dict(w=w, a=a, gw=gw, ga=ga, yw=torch.empty_like(w), ya=torch.empty_like(a),
                                  gxw=torch.empty_like(w), gxa=torch.empty_like(a),
                                  bw=torch.empty(rows, 2, device=device), ba=torch.empty(rows, 2, device=device),
                                  mw=torch.empty(self.mask_bytes, dtype=torch.uint8, device=device),
                                  ma=torch.empty(self.mask_bytes, dtype=torch.uint8, device=device)))
        self.L, self._lib = L, _lib
        self.stream = torch.cuda.current_stream(device).cuda_stream

    # raw C-ABI launches on preallocated buffers (what the autograd Functions do, minus the allocator).
    # Default data flow = the product's default ("mask" mode): fq_sym_fwd_train + fq_ste_bwd_mask.
    def fwd(self, s, leg):
        x, y, b, m, bits = (s["w"], s["yw"], s["bw"], s["mw"], 4) if leg == "w" else (s["a"], s["ya"], s["ba"], s["ma"], 8)
        rc = self.L.fq_sym_fwd_train(x.data_ptr(), y.data_ptr(), self.rows, self.cols, bits, self._lib.DTYPE_BF16,
                                     self._lib.SEM_CPU_EAGER, -2.0, 2.0, b.data_ptr(), m.data_ptr(), self.mask_bytes, self.stream)
        if rc:
            self._lib.check(rc, "fq_sym_fwd_train")

    def bwd(self, s, leg, inplace=False):
        g, gx, b, m = (s["gw"], s["gxw"], s["bw"], s["mw"]) if leg == "w" else (s["ga"], s["gxa"], s["ba"], s["ma"])
        if inplace:
            gx = g
        rc = self.L.fq_ste_bwd_mask(g.data_ptr(), gx.data_ptr(), self.rows, self.cols, -2.0, 2.0, b.data_ptr(), m.data_ptr(),
                                    self.mask_bytes, self._lib.DTYPE_BF16, self.stream)
        if rc:
            self._lib.check(rc, "fq_ste_bwd_mask")

    # the reference's data flow (backward re-reads x), for comparison entries
    def fwd_plain(self, s, leg):
        x, y, bits = (s["w"], s["yw"], 4) if leg == "w" else (s["a"], s["ya"], 8)
        rc = self.L.fq_sym_fwd(x.data_ptr(), y.data_ptr(), self.rows, self.cols, bits, self._lib.DTYPE_BF16,
                               self._lib.SEM_CPU_EAGER, None, None, 0, self.stream)
        if rc:
            self._lib.check(rc, "fq_sym_fwd")

    # SymQuantizer as the reference runs it under torch.autocast("cuda", bf16) (LLM-QAT's training configuration):
    # fp32 arithmetic behind the reciprocal; narrow = rounded once to bf16 (QuantizeLinear operands), wide = fp32 result
    def fwd_autocast(self, s, leg, wide):
        x, y, b, m, bits = (s["w"], s["yw"], s["bw"], s["mw"], 4) if leg == "w" else (s["a"], s["ya"], s["ba"], s["ma"], 8)
        if wide:
            if not hasattr(self, "y32"):
                self.y32 = self.torch.empty(self.rows, self.cols, device=x.device)
            rc = self.L.fq_sym_fwd_autocast(x.data_ptr(), self.y32.data_ptr(), self.rows, self.cols, bits, self._lib.DTYPE_BF16, 1, 1, -2.0, 2.0,
                                            b.data_ptr(), None, 0, None, 0, self.stream)
        else:
            rc = self.L.fq_sym_fwd_autocast(x.data_ptr(), y.data_ptr(), self.rows, self.cols, bits, self._lib.DTYPE_BF16, 1, 0, -2.0, 2.0,
                                            b.data_ptr(), m.data_ptr(), self.mask_bytes, None, 0, self.stream)
        if rc:
            self._lib.check(rc, "fq_sym_fwd_autocast")

    def bwd_xread(self, s, leg):
        g, x, gx = (s["gw"], s["w"], s["gxw"]) if leg == "w" else (s["ga"], s["a"], s["gxa"])
        rc = self.L.fq_ste_bwd(g.data_ptr(), x.data_ptr(), gx.data_ptr(), self.n, -2.0, 2.0, self._lib.DTYPE_BF16, self.stream)
        if rc:
            self._lib.check(rc, "fq_ste_bwd")

    # The product's default data flow for a QuantizeLinear: its weight [out, in] (W4) and its input [tokens, in] (A8)
    # in ONE launch forward, their STE gradients in one launch backward (fq_sym_fwd_pair / fq_ste_bwd_mask_pair).
    def fwd_pair(self, s):
        rc = self.L.fq_sym_fwd_pair(s["w"].data_ptr(), s["yw"].data_ptr(), self.rows, 4, s["bw"].data_ptr(), s["mw"].data_ptr(), self.mask_bytes,
                                    s["a"].data_ptr(), s["ya"].data_ptr(), self.rows, 8, s["ba"].data_ptr(), s["ma"].data_ptr(), self.mask_bytes,
                                    self.cols, self._lib.DTYPE_BF16, self._lib.SEM_CPU_EAGER, 0, -2.0, 2.0, self.stream)
        if rc:
            self._lib.check(rc, "fq_sym_fwd_pair")

    def bwd_pair(self, s, inplace_w=True):
        # the product's default (utils_quant.py, point 6): the WEIGHT's gradient is masked where it stands (gx == g) -- its rows
        # cannot clip, so the kernel touches none of it; the activation's gradient is written to a fresh tensor as always
        gxw = s["gw"] if inplace_w else s["gxw"]
        rc = self.L.fq_ste_bwd_mask_pair(s["gw"].data_ptr(), gxw.data_ptr(), self.rows, s["bw"].data_ptr(), s["mw"].data_ptr(),
                                         s["ga"].data_ptr(), s["gxa"].data_ptr(), self.rows, s["ba"].data_ptr(), s["ma"].data_ptr(),
                                         self.cols, -2.0, 2.0, self._lib.DTYPE_BF16, self.stream)
        if rc:
            self._lib.check(rc, "fq_ste_bwd_mask_pair")

    def step(self, i):
        # forward on set i, backward on the set whose forward ran two steps ago: between the forward
        # and the backward of one tensor > 1.4 GB of other traffic passes, as in a real training step
        sf = self.sets[i % self.nsets]
        sb = self.sets[(i + self.nsets - 2) % self.nsets]
        self.fwd_pair(sf)
        self.bwd_pair(sb)

    def step_out_of_place(self, i):
        # the like-for-like step (the reference's data flow, utils_quant.py:83-87 `grad_output.clone()`): BOTH gradients are
        # written to fresh tensors, as round 1 measured it and as SymQuantizer.apply does for any caller-visible gradient
        sf = self.sets[i % self.nsets]
        sb = self.sets[(i + self.nsets - 2) % self.nsets]
        self.fwd_pair(sf)
        self.bwd_pair(sb, inplace_w=False)

    def step_unpaired(self, i):
        sf = self.sets[i % self.nsets]
        sb = self.sets[(i + self.nsets - 2) % self.nsets]
        self.fwd(sf, "w")
        self.fwd(sf, "a")
        self.bwd(sb, "a")
        self.bwd(sb, "w", inplace=True)

    def prime_bounds(self):
        for s in self.sets:
            self.fwd(s, "w")
            self.fwd(s, "a")
        self.torch.cuda.synchronize()

    def clippable_fraction(self):
        """fraction of the A8 tensor's rows whose recorded bounds reach the clip (they carry a mask: 1 bit/element)"""
        b = self.sets[0]["ba"]
        return float(((b[:, 0] >= 2.0) | (b[:, 1] <= -2.0)).float().mean().item())

    def time_kernel(self, fn, iters, sets=None, name=None, fq=1):
        return time_launches(self.torch, fn, iters, sets if sets is not None else self.sets, name, fq)


PROFILE_MANIFEST = []   # (entry name, fq:: launches it made), in launch order: lets tools/summarize_profile.py split a rocprofv3
#                         trace / PMC pass of this process into one segment per entry by counting fq:: dispatches


MIN_KERNEL_ITERS, MIN_KERNEL_WARMUP = 200, 20   # SURVEY §8d's GPU timing protocol: >= 200 iterations after >= 20 warm-ups


def time_launches(torch, fn, iters, sets, name=None, fq_launches_per_call=1, min_iters=MIN_KERNEL_ITERS):
    """Launch duration of one kernel kind with HIP events on the launch stream (torch's current stream IS the stream the
    C ABI is handed), rotating buffers.  Whatever --steps says, every kernel is timed over >= 200 launches after >= 20 warm-ups
    (round 3 timed 20 launches after 4 warm-ups at the driver's --steps 20: its per-kernel numbers did not reproduce).
    -> (p50 ms of individually bracketed launches, [p10, p50, p90] ms of the same, mean ms over a back-to-back batch)"""
    ns = len(sets)
    iters = max(iters, min_iters)
    warm = max(MIN_KERNEL_WARMUP, ns)   # every buffer set at least once: the first launches that touch fresh memory run 30-50 % slow on small tensors
    PROFILE_MANIFEST.append((name or getattr(fn, "__name__", "?"), (warm + 2 * iters) * fq_launches_per_call))
    for i in range(warm):
        fn(sets[i % ns])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(iters):
        fn(sets[i % ns])
    e1.record()
    torch.cuda.synchronize()
    mean = e0.elapsed_time(e1) / iters
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for i, (a, b) in enumerate(pairs):
        a.record()
        fn(sets[i % ns])
        b.record()
    torch.cuda.synchronize()
    d = sorted(a.elapsed_time(b) for a, b in pairs)
    pct = [d[len(d) // 10], d[len(d) // 2], d[(9 * len(d)) // 10]]
    return Timing(mean, pct, iters, warm)


class Timing(tuple):
    """(mean ms of the back-to-back batch, [p10, p50, p90] ms of bracketed launches) + .iters / .warmup"""

    def __new__(cls, ms, pct, iters, warm):
        t = super().__new__(cls, (ms, pct))
        t.iters, t.warmup = iters, warm
        return t


class ModelShapes:
    """The quantizer launches of ONE LLaMA-7B decoder layer at their real shapes (seq 2048, bs 1; reference call sites
    models/modeling_llama_quant.py:313-327 attention, :235 MLP), through the C ABI on preallocated rotating buffers."""

    def __init__(self, wl, tokens=2048, hidden=4096, inter=11008):
        torch = wl.torch
        self.wl, self.torch, self.L, self._lib = wl, torch, wl.L, wl._lib
        self.tokens, self.hidden, self.inter = tokens, hidden, inter
        self.cache = {}

    def sets_for(self, rows, cols, style, want_y32=False):
        key = (rows, cols, style, want_y32)
        if key in self.cache:
            return self.cache[key]
        torch, dev = self.torch, self.wl.device
        code = self._lib.DTYPE_BF16
        n = rows * cols
        nsets = max(3, min(12, int(700e6 // (n * 2 * 4)) + 1))   # > 256 MiB of distinct buffers where that is affordable
        mb = self.L.fq_ste_mask_bytes(rows, cols, code)
        g = torch.Generator(device=dev).manual_seed(4321 + rows + cols)
        sets = []
        for _ in range(nsets):
            x = (torch.randn(rows, cols, generator=g, device=dev) * 0.02).bfloat16() if style.startswith("w") else _act_like(torch, rows, cols, g, dev)
            d = dict(x=x, y=torch.empty_like(x), g=(torch.randn(rows, cols, generator=g, device=dev) * 1e-3).bfloat16(), gx=torch.empty_like(x),
                     b=torch.empty(rows, 2, device=dev), m=torch.empty(max(mb, 8), dtype=torch.uint8, device=dev), mb=mb)
            if want_y32:
                d["y32"] = torch.empty(rows, cols, device=dev)
                d["g32"] = torch.randn(rows, cols, generator=g, device=dev) * 1e-3
            sets.append(d)
        self.cache[key] = sets
        return sets

    def chk(self, rc, what="model_shapes"):
        if rc:
            self._lib.check(rc, what)

    def single_fwd(self, rows, cols, bits, style):
        L, code, st = self.L, self._lib.DTYPE_BF16, self.wl.stream
        sets = self.sets_for(rows, cols, style)
        return (lambda s: self.chk(L.fq_sym_fwd_train(s["x"].data_ptr(), s["y"].data_ptr(), rows, cols, bits, code, 0, -2.0, 2.0,
                                                      s["b"].data_ptr(), s["m"].data_ptr(), s["mb"], st))), sets

    def single_bwd(self, rows, cols, style):
        L, code, st = self.L, self._lib.DTYPE_BF16, self.wl.stream
        sets = self.sets_for(rows, cols, style)
        return (lambda s: self.chk(L.fq_ste_bwd_mask(s["g"].data_ptr(), s["gx"].data_ptr(), rows, cols, -2.0, 2.0, s["b"].data_ptr(),
                                                     s["m"].data_ptr(), s["mb"], code, st))), sets

    def pair_fwd(self, rows0, style0, bits0, rows1, style1, bits1, cols, autocast=0):
        """tensor 0 + tensor 1 in one launch; autocast 2 = fp32 results (the K / V hooks under autocast)"""
        L, code, st = self.L, self._lib.DTYPE_BF16, self.wl.stream
        wide = autocast == 2
        s0 = self.sets_for(rows0, cols, style0, wide)
        s1 = self.sets_for(rows1, cols, style1 + "2", wide)   # distinct buffers even when both tensors have one shape
        sets = [dict(a=a, b=b) for a, b in zip(s0, s1)]
        yk = "y32" if wide else "y"

        def fn(s):
            a, b = s["a"], s["b"]
            self.chk(L.fq_sym_fwd_pair(a["x"].data_ptr(), a[yk].data_ptr(), rows0, bits0, a["b"].data_ptr(), a["m"].data_ptr(), a["mb"],
                                       b["x"].data_ptr(), b[yk].data_ptr(), rows1, bits1, b["b"].data_ptr(), b["m"].data_ptr(), b["mb"],
                                       cols, code, 1 if autocast else 0, autocast, -2.0, 2.0, st), "fq_sym_fwd_pair")
        return fn, sets

    def pair_bwd(self, rows0, style0, rows1, style1, cols, wide=False, inplace0=False):
        """inplace0: tensor 0 is a weight whose gradient the product masks where it stands (gx == g)"""
        L, code, st = self.L, self._lib.DTYPE_BF16, self.wl.stream
        s0 = self.sets_for(rows0, cols, style0, wide)
        s1 = self.sets_for(rows1, cols, style1 + "2", wide)
        sets = [dict(a=a, b=b) for a, b in zip(s0, s1)]
        gk = "g32" if wide else "g"
        f = L.fq_ste_bwd_mask_wide if wide else L.fq_ste_bwd_mask_pair

        def fn(s):
            a, b = s["a"], s["b"]
            self.chk(f(a[gk].data_ptr(), (a[gk] if inplace0 else a["gx"]).data_ptr(), rows0, a["b"].data_ptr(), a["m"].data_ptr(),
                       b[gk].data_ptr(), b["gx"].data_ptr(), rows1, b["b"].data_ptr(), b["m"].data_ptr(), cols, -2.0, 2.0, code, st), "pair_bwd")
        return fn, sets

    def group_launch(self, specs, cols, backward=False, inplace=()):
        """a sibling group in ONE launch (fq_sym_fwd_multi / fq_ste_bwd_mask_multi): specs = [(rows, style, bits), ...]"""
        L, lib_, st = self.L, self._lib, self.wl.stream
        code = lib_.DTYPE_BF16
        per = [self.sets_for(r, cols, style + f"g{i}") for i, (r, style, _) in enumerate(specs)]
        sets = [dict(t=list(ts)) for ts in zip(*per)]
        n = len(specs)

        def fwd(s):
            arr = (lib_.FwdTensor * n)()
            for i, (d, (r, _, bits)) in enumerate(zip(s["t"], specs)):
                arr[i] = lib_.FwdTensor(d["x"].data_ptr(), d["y"].data_ptr(), r, bits, d["b"].data_ptr(), d["m"].data_ptr(), d["mb"])
            self.chk(L.fq_sym_fwd_multi(n, arr, cols, code, 0, 0, -2.0, 2.0, st), "fq_sym_fwd_multi")

        def bwd(s):
            arr = (lib_.BwdTensor * n)()
            for i, (d, (r, _, _)) in enumerate(zip(s["t"], specs)):
                arr[i] = lib_.BwdTensor(d["g"].data_ptr(), (d["g"] if i in inplace else d["gx"]).data_ptr(), r, d["b"].data_ptr(), d["m"].data_ptr())
            self.chk(L.fq_ste_bwd_mask_multi(n, arr, cols, -2.0, 2.0, code, 0, st), "fq_ste_bwd_mask_multi")

        return (bwd if backward else fwd), sets

    def _spec(self, name, site, fn_sets, elems, fwd, extra_mask_elems=0, bpe_in=2, bpe_out=2):
        """one launch kind of the layer: what it is at the reference call site, the launch, its rotating buffers, and its
        byte counts (algorithmic = SURVEY §8d accounting; moved = what the kernel touches by design, incl. mask bits)"""
        fn, sets = fn_sets
        moved = elems * (bpe_in + bpe_out) + extra_mask_elems // 8
        algo = elems * (FWD_BYTES_PER_ELEM if fwd else BWD_BYTES_PER_ELEM) if (bpe_in, bpe_out) == (2, 2) else moved
        return dict(name=name, site=site, fn=fn, sets=sets, algo=algo, moved=moved)

    def specs_compact(self, tag):
        """the layer's dominant launches only (config 5: LLaMA-13B dimensions)"""
        T, H, I = self.tokens, self.hidden, self.inter
        S = self._spec
        return [
            S(f"{tag} down_proj pair fwd: W4 [{H},{I}] + A8 [{T},{I}]", "utils_quant.py:195-201,:244-248", self.pair_fwd(H, "w", 4, T, "a", 8, I), (H + T) * I, True, T * I),
            S(f"{tag} down_proj pair bwd (weight gradient in place)", "utils_quant.py:77-87 x2", self.pair_bwd(H, "w", T, "a", I, inplace0=True), T * I, False, T * I),
            S(f"{tag} q_proj pair fwd: W4 [{H},{H}] + A8 [{T},{H}]", "modeling_llama_quant.py:313", self.pair_fwd(H, "w", 4, T, "a", 8, H), (H + T) * H, True, T * H),
            S(f"{tag} quantize_kv pair fwd: K4 + V4 [{T},{H}] x2", "modeling_llama_quant.py:320-327", self.pair_fwd(T, "a", 4, T, "a", 4, H), 2 * T * H, True, 2 * T * H),
            S(f"{tag} W4 [{I},{H}] fwd (gate/up weight)", "utils_quant.py:195-201", self.single_fwd(I, H, 4, "w"), I * H, True, 0)]

    def specs(self):
        """one spec per launch kind of the layer, in an order where every backward finds the bounds + masks its forward left"""
        T, H, I = self.tokens, self.hidden, self.inter
        S = self._spec
        qkv = [(H, "w", 4), (T, "a", 8), (H, "w", 4), (H, "w", 4)]
        gu = [(I, "w", 4), (T, "a", 8), (I, "w", 4)]
        return [
            # forward launches (training mode: bounds + STE mask recorded)
            S("down_proj pair fwd: W4 [4096,11008] + A8 [2048,11008]", "utils_quant.py:195-201,:244-248 via modeling_llama_quant.py:235",
              self.pair_fwd(H, "w", 4, T, "a", 8, I), (H + T) * I, True, T * I),
            S("A8 [2048,11008] fwd (down_proj input alone)", "utils_quant.py:244-248", self.single_fwd(T, I, 8, "a"), T * I, True, T * I),
            S("A8 [2048,4096] fwd (shared q/k/v or gate/up input)", "modeling_llama_quant.py:313,317,318,:235", self.single_fwd(T, H, 8, "a"), T * H, True, T * H),
            S("KV4 [2048,4096] fwd (one of K, V)", "modeling_llama_quant.py:320-327", self.single_fwd(T, H, 4, "a"), T * H, True, T * H),
            S("quantize_kv pair fwd: K4 + V4 [2048,4096] x2", "modeling_llama_quant.py:320-327 (one launch)", self.pair_fwd(T, "a", 4, T, "a", 4, H),
              2 * T * H, True, 2 * T * H),
            S("quantize_kv pair fwd under autocast: fp32 results", "modeling_llama_quant.py:320-327 under kd_trainer.py:106 autocast",
              self.pair_fwd(T, "a", 4, T, "a", 4, H, autocast=2), 2 * T * H, True, 2 * T * H, bpe_in=2, bpe_out=4),
            S("q_proj pair fwd: W4 [4096,4096] + A8 [2048,4096]", "modeling_llama_quant.py:313", self.pair_fwd(H, "w", 4, T, "a", 8, H), (H + T) * H, True, T * H),
            S("W4 [11008,4096] fwd (gate/up weight)", "utils_quant.py:195-201", self.single_fwd(I, H, 4, "w"), I * H, True),
            S("q/k/v group fwd in ONE launch: W4 [4096,4096] x3 + their shared A8 input [2048,4096]", "modeling_llama_quant.py:313,317,318 (sibling group)",
              self.group_launch(qkv, H), (3 * H + T) * H, True, T * H),
            S("gate/up group fwd in ONE launch: W4 [11008,4096] x2 + their shared A8 input [2048,4096]", "modeling_llama_quant.py:235 (sibling group)",
              self.group_launch(gu, H), (2 * I + T) * H, True, T * H),
            # backward launches (their forwards above have filled bounds + masks of the same buffers)
            S("down_proj pair bwd (weight gradient in place: product default)", "utils_quant.py:77-87 x2", self.pair_bwd(H, "w", T, "a", I, inplace0=True),
              T * I, False, T * I),
            S("down_proj pair bwd, both gradients to fresh tensors", "utils_quant.py:77-87 x2", self.pair_bwd(H, "w", T, "a", I), (H + T) * I, False, T * I),
            S("q/k/v group bwd in ONE launch (weight gradients in place)", "utils_quant.py:77-87 x4", self.group_launch(qkv, H, backward=True, inplace=(0, 2, 3)),
              T * H, False, T * H),
            S("A8 [2048,4096] bwd", "utils_quant.py:77-87", self.single_bwd(T, H, "a"), T * H, False, T * H),
            S("quantize_kv pair bwd", "utils_quant.py:77-87 x2", self.pair_bwd(T, "a", T, "a", H), 2 * T * H, False, 2 * T * H),
            S("quantize_kv pair bwd under autocast: fp32 grads in, bf16 out", "utils_quant.py:77-87 x2 + the engine's cast",
              self.pair_bwd(T, "a", T, "a", H, wide=True), 2 * T * H, False, 2 * T * H, bpe_in=4, bpe_out=2)]

    def entries_compact(self, iters, tag):
        return time_specs(self.torch, self.specs_compact(tag), iters)

    def entries(self, iters):
        return time_specs(self.torch, self.specs(), iters)


def time_specs(torch, specs, iters, traffic=None, traffic_source=None):
    """roofline entries of a list of launch specs (ModelShapes._spec / export_specs / lowbit_asym_specs)"""
    out = []
    for sp in specs:
        t = time_launches(torch, sp["fn"], iters, sp["sets"], sp["name"], sp.get("fq", 1))
        e = roofline_entry(sp["name"], sp["algo"], t, (traffic or {}).get(sp["name"]), moved_bytes=sp["moved"], traffic_source=traffic_source,
                           method="bracketed_p50")   # extras: one sample for the value and its percentiles (VERDICT r04 "weak" #8)
        if sp.get("site"):
            e["reference_call_site"] = sp["site"]
        out.append(e)
    return out


def baseline_metric_name():
    """BASELINE.json's metric string, verbatim (value is its first quantity, Gelem/s; the second, achieved HBM GB/s,
    is the roofline objects)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:  # noqa: BLE001
        return "fake-quant fwd+bwd Gelem/s & achieved HBM GB/s, 4096×11008 bf16 W4A8"


def roofline_entry(name, algorithmic_bytes, timing, traffic=None, moved_bytes=None, traffic_source=None, method="batch_mean"):
    """One kernel's roofline entry.
      method                     which of the two timings feeds us_per_launch / achieved / frac (it is named in `timing_method`):
                                 "batch_mean" -- HIP-event time of a back-to-back batch / launches (what rocprofv3's per-kernel average
                                 reproduces within 1 %: the headline `roofline` and the step's kernels); "bracketed_p50" -- the median of
                                 individually bracketed launches, the SAME sample the p10 / p50 / p90 come from (every extras family since
                                 round 5, so that us_per_launch can no longer sit below its own p10; a bracketed launch reads 1.5-2 us
                                 longer than the kernel -- the gap between two event records -- so these fractions err low)
      achieved / frac            bytes the kernel MOVES by design (moved_bytes; = the algorithmic bytes unless given) / launch
                                 time: a real byte rate, so frac <= 1 by construction
      achieved_algorithmic /     SURVEY §8d's accounting (4 B/elem forward, 6 B/elem backward, bf16) / launch time -- only
      frac_algorithmic           present where it differs: the mask backward does not re-read x, so against the reference's
                                 6 B/elem data flow it reads > its byte rate ("reference-dataflow equivalent", can exceed 1)
      traffic                    HBM bytes per launch measured with rocprofv3 --pmc; traffic_source says where it was measured"""
    ms, pct = timing if isinstance(timing, tuple) else (timing, None)
    batch_ms = ms
    if method == "bracketed_p50" and pct:
        ms = pct[1]
    moved = algorithmic_bytes if moved_bytes is None else moved_bytes
    ach = moved / (ms * 1e-3) / 1e9
    e = {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "us_per_launch": round(ms * 1e3, 2),
         "bytes_moved_per_launch": moved, "algorithmic_bytes_per_launch": algorithmic_bytes}
    if moved != algorithmic_bytes:
        alg = algorithmic_bytes / (ms * 1e-3) / 1e9
        e["achieved_algorithmic"] = round(alg, 1)
        e["frac_algorithmic"] = round(alg / HBM_PEAK_GBS, 4)
        e["frac_algorithmic_note"] = "reference-dataflow equivalent (SURVEY §8d accounting), not a byte rate"
    if traffic:
        e["traffic_gbs"] = round(traffic / (ms * 1e-3) / 1e9, 1)
        e["traffic_frac"] = round(traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        e["traffic_source"] = traffic_source or "profiles/traffic.json"
    if pct:
        e["us_p10_p50_p90"] = [round(v * 1e3, 2) for v in pct]
        e["timing_method"] = method if method == "batch_mean" or pct else "batch_mean"
        if method == "bracketed_p50":
            e["us_batch_mean"] = round(batch_ms * 1e3, 2)
            e["frac_batch_mean"] = round(moved / (batch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)   # the round-4 way of stating the same entry
    if isinstance(timing, Timing):   # launches behind either number: `iters` after `warmup` warm-ups
        e["launches_timed"], e["warmup_launches"] = timing.iters, timing.warmup
    return e


def load_traffic(name="traffic.json"):
    """HBM bytes per launch from committed rocprofv3 --pmc runs (profiles/traffic.json for the step's kernels,
    profiles/traffic_model_shapes.json for `--model-shapes` entries, keyed by entry name), if present -- NOT measured in this
    run (PMC collection needs the profiler around the process); the JSON line says so in `traffic_source`."""
    p = os.path.join(ROOT, "profiles", name)
    if os.path.exists(p):
        try:
            t = json.load(open(p))
            src = t.pop("_source", None) or (f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier bench.py run, "
                                             "file mtime " + time.strftime("%Y-%m-%d", time.gmtime(os.path.getmtime(p))) + "; not measured in this run)")
            return t, src
        except Exception:
            return {}, None
    return {}, None


def cpu_baseline(budget_s=15.0):
    """The reference's CPU path on this host: eager op chain on the full W4 + A8 step, repeated for ~budget_s."""
    import torch
    from oracle import eager_chain as E
    gw_, gg_, ga_ = (torch.Generator().manual_seed(sd) for sd in (1234, 1235, 1236))   # BASELINE.md §3 seeds
    w = (torch.randn(ROWS, COLS, generator=gw_) * 0.02).bfloat16()
    a = torch.randn(ROWS, COLS, generator=ga_)
    a[torch.rand(ROWS, COLS, generator=ga_) < 1e-3] *= 20.0
    a = a.bfloat16()
    gw = (torch.randn(ROWS, COLS, generator=gg_) * 1e-3).bfloat16()
    clip = torch.tensor([-2.0, 2.0])
    times = []
    t_start = time.perf_counter()
    while True:
        t0 = time.perf_counter()
        E.sym_forward(w, 4)
        E.ste_backward(gw, w, clip)
        E.sym_forward(a, 8)
        E.ste_backward(gw, a, clip)
        times.append(time.perf_counter() - t0)
        if (time.perf_counter() - t_start > budget_s and len(times) >= 2) or len(times) >= 50:
            break
    best = min(times[1:]) if len(times) > 1 else times[0]
    med = sorted(times[1:])[len(times[1:]) // 2] if len(times) > 1 else times[0]
    cores = torch.get_num_threads()
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(2 * ROWS * COLS / best / 1e9, 4), "unit": "Gelem/s", "cores": cores, "kind": "port",
            "sample": f"full step (W4 + A8 fwd+bwd on [4096,11008] bf16), {len(times)} repeats, best of all but the first; "
                      f"eager ATen op chain restating models/utils_quant.py (bit-equal to the reference on CPU)",
            "host_cpu": model, "host_logical_cpus": os.cpu_count(), "seconds_best": round(best, 4), "seconds_median": round(med, 4),
            "value_median": round(2 * ROWS * COLS / med / 1e9, 4)}


def cpu_c_port(rows_sample=256):
    """Second CPU line: the scalar C oracle (oracle/fq_oracle.c), one core, on a row sample of the same step."""
    import numpy as np
    import torch
    from oracle import oracle as O
    g = torch.Generator().manual_seed(1234)
    w = (torch.randn(rows_sample, COLS, generator=g) * 0.02).bfloat16()
    a = torch.randn(rows_sample, COLS, generator=g).bfloat16()
    gw = (torch.randn(rows_sample, COLS, generator=g) * 1e-3).bfloat16()
    tonp = lambda t: t.view(torch.int16).numpy().view(np.uint16)  # noqa: E731
    wn, an, gn = tonp(w), tonp(a), tonp(gw)
    O.lib()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        O.sym_fwd(wn, rows_sample, COLS, 4, "bf16", want_idx=False)
        O.ste_bwd(gn, wn, -2.0, 2.0, "bf16")
        O.sym_fwd(an, rows_sample, COLS, 8, "bf16", want_idx=False)
        O.ste_bwd(gn, an, -2.0, 2.0, "bf16")
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {"value": round(2 * rows_sample * COLS / best / 1e9, 4), "unit": "Gelem/s", "cores": 1, "kind": "port",
            "sample": f"{rows_sample} of 4096 rows of each tensor of the step, scalar C restatement (oracle/fq_oracle.c), best of 3"}


def parity_gate(wl, nrows=48):
    """Reported with the timing: the kernels the step just ran, on a row sample of its own tensors, against
    the CPU oracle -- bins and dequantized values and gradients must be bit-equal.  Part of the cpu_baseline leg (the only
    place bench.py may touch oracle/), never inside a timed region."""
    import numpy as np
    import torch
    import llm_qat_amd
    from oracle import oracle as O
    s = wl.sets[0]
    idxs = torch.linspace(0, wl.rows - 1, nrows).long().to(s["w"].device)
    tonp = lambda t: t.detach().contiguous().cpu().view(torch.int16).numpy().view(np.uint16)  # noqa: E731
    res = {"rows_checked": int(nrows), "bins_bit_exact": True, "values_bit_exact": True, "grad_bit_exact": True}
    for x, g, bits in ((s["w"], s["gw"], 4), (s["a"], s["ga"], 8)):
        xs, gs = x[idxs].contiguous(), g[idxs].contiguous()
        y, idx, _ = llm_qat_amd.ops.sym_quantize_debug(xs, bits, False)
        yo, io, _ = O.sym_fwd(tonp(xs), nrows, wl.cols, bits, "bf16")
        res["bins_bit_exact"] &= bool((idx.cpu().numpy() == io).all())
        res["values_bit_exact"] &= bool((tonp(y) == yo).all())
        tr = llm_qat_amd.ops.quantize_train("sym", xs, bits, False, -2.0, 2.0)
        gx = llm_qat_amd.ops.ste_backward_mask(gs, -2.0, 2.0, tr[1], tr[2], nrows, wl.cols)
        res["values_bit_exact"] &= bool((tonp(tr[0]) == yo).all())
        res["grad_bit_exact"] &= bool((tonp(gx) == O.ste_bwd(tonp(gs), tonp(xs), -2.0, 2.0, "bf16")).all())
    return res


def gpu_eager(wl, iters=5, autocast=False):
    """The reference's eager chain on this GPU: the like-for-like 'before' (14 launches per fwd+bwd).
    autocast=True: inside torch.autocast("cuda", bf16), as LLM-QAT trains (fp32 intermediates behind the reciprocal)."""
    import torch
    clip = torch.tensor([-2.0, 2.0])

    # the op chain of models/utils_quant.py:53-59,:71-72 (forward) and :83-87 (backward), restated here for the timing
    # (nothing from oracle/ runs on the GPU legs of the benchmark)
    def sym_forward(x, bits):
        top = torch.max(torch.abs(x), dim=-1, keepdim=True)[0].expand_as(x)
        s = (2 ** (bits - 1) - 1) / (top + 1e-6)
        return torch.round(x * s).div(s + 1e-6)

    def ste_backward(g, x):
        gx = g.clone()
        gx[x.ge(clip[1])] = 0
        gx[x.le(clip[0])] = 0
        return gx

    def one(k):
        s = wl.sets[k % wl.nsets]
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            yw = sym_forward(s["w"], 4)
            ya = sym_forward(s["a"], 8)
        ste_backward(s["gw"].to(yw.dtype), s["w"])
        ste_backward(s["ga"].to(ya.dtype), s["a"])

    one(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(iters):
        one(k + 1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    return {"value": round(2 * wl.n / dt / 1e9, 2), "unit": "Gelem/s", "ms_per_step": round(dt * 1e3, 3),
            "what": "the reference's eager op chain (restated in bench.py) on the same tensors, same GPU"
                    + (", inside torch.autocast(cuda, bf16) as LLM-QAT trains (fp32 intermediates, fp32 outputs)" if autocast else "")}


def autograd_path(wl, iters=200):
    """The same step through the product's autograd Functions (SymQuantizer.apply + .backward on the step's own tensors):
    allocator, Python, ctypes and autograd-engine cost included -- what a training loop pays per call."""
    import torch
    from llm_qat_amd.utils_quant import SymQuantizer
    clip = torch.tensor([-2.0, 2.0])
    leaves = []
    for s in wl.sets:
        leaves.append((s["w"].detach().requires_grad_(True), s["a"].detach().requires_grad_(True), s["gw"], s["ga"]))

    def one(k):
        w, a, gw, ga = leaves[k % len(leaves)]
        w.grad = a.grad = None
        yw = SymQuantizer.apply(w, clip, 4, False)
        ya = SymQuantizer.apply(a, clip, 8, False)
        torch.autograd.backward([yw, ya], [gw, ga])

    for k in range(20):
        one(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(iters):
        one(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    return {"ms_per_step": round(dt * 1e3, 4), "value": round(2 * wl.n / dt / 1e9, 2), "unit": "Gelem/s",
            "what": "SymQuantizer.apply(W4) + SymQuantizer.apply(A8) + autograd backward on the step's tensors: 4 launches, "
                    "outputs / side buffers / gradients from PyTorch's caching allocator (wall clock incl. host overhead)"}


def self_check(out):
    """Cross-checks a reader can redo from the line alone: a headline that fails one is not a measurement of the named
    metric.  -> {"ok": bool, ...} or None when the record carries no real measurement (stub)."""
    if out.get("stub") or "ms_per_step" not in out or not out.get("config", {}).get("elements_per_step"):
        return None
    ms, elems, world = out["ms_per_step"], out["config"]["elements_per_step"], out.get("n_gpus", 1)
    res = {}
    # (1) SURVEY §8d's accounting on the headline: 10 B/elem x elements / time may not exceed the chip's peak
    eq = (FWD_BYTES_PER_ELEM + BWD_BYTES_PER_ELEM) * elems / (ms * 1e-3) / 1e9
    res["accounting_gbs_10B_per_elem"] = round(eq, 1)
    res["accounting_below_peak"] = bool(eq <= HBM_PEAK_GBS)
    rs = out.get("roofline_step")
    if rs and world == 1:
        # (2) the two separately timed launches of the step cannot take longer than the timed step that contains them (3 % slack)
        res["launch_sum_vs_step"] = round(rs["us_per_step_launches"] / (ms * 1e3), 4)
        res["launch_sum_within_step"] = bool(rs["us_per_step_launches"] <= 1.03 * ms * 1e3)
        res["roofline_step_frac_below_1"] = bool(rs["frac"] <= 1.0)
    r = out.get("roofline")
    if r:
        res["roofline_frac_below_1"] = bool(r["frac"] <= 1.0)
    res["ok"] = all(v for k, v in res.items() if isinstance(v, bool))
    return res


def api_path(wl, iters=200):
    """What the module API delivers on the metric tensors: QuantizeLinear(11008 -> 4096, W4 A8).forward + .backward through the real
    module code (operand pairing: ONE forward launch; _PairNode: one backward launch with the weight's gradient masked in place behind
    its guard; activation cache, autograd, allocator, Python all included) -- with the GEMM taken out: F.linear is replaced, for this
    timing only, by a stand-in that launches nothing (its forward returns an empty output, its backward fresh gradient tensors of the
    right shapes, as F.linear's backward does), so the figure is the cost of the fake-quant path through the API and compares with
    `value`.  `with_gemm` is the same call with the real F.linear (hipBLASLt), for scale."""
    import torch
    import torch.nn.functional as F
    import llm_qat_amd
    from llm_qat_amd.utils_quant import QuantizeLinear
    dev = wl.device
    tokens = wl.rows

    class _NoGemm(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            return torch.empty(x.shape[:-1] + (w.shape[0],), dtype=x.dtype, device=x.device)

        @staticmethod
        def backward(ctx, go):
            x, w = ctx.saved_tensors
            return torch.empty_like(x), torch.empty_like(w)   # fresh tensors nobody else holds, like a GEMM's outputs

    lins = []
    for s in wl.sets:
        lin = QuantizeLinear(wl.cols, wl.rows, w_bits=4, a_bits=8).to(device=dev, dtype=torch.bfloat16)
        lin.weight.data = s["w"]
        lins.append((lin, s["a"].detach().requires_grad_(True)))
    go = torch.empty(tokens, wl.rows, dtype=torch.bfloat16, device=dev)

    class _Floor(torch.nn.Module):   # what PyTorch itself costs for a module of this shape: module call, one autograd Function, two leaves, the engine
        def __init__(self, lin):
            super().__init__()
            self.weight = lin.weight

        def forward(self, x):
            return F.linear(x, self.weight)

    floors = [(_Floor(lin), a) for lin, a in lins]

    last = {}

    def timed(n, mods=lins):
        def one(k):
            m, a = mods[k % len(mods)]
            m.weight.grad = a.grad = None
            m(a).backward(go)
        for k in range(20):
            one(k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for k in range(n):
            one(k)
        e1.record()
        t_host = time.perf_counter() - t0      # everything enqueued
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        last.update(host=t_host / n, gpu=e0.elapsed_time(e1) * 1e-3 / n)
        return dt

    real = F.linear
    import llm_qat_amd.utils_quant as _UQ
    no_gemm = _UQ._cnode.no_gemm_linear if _UQ._cnode is not None else _NoGemm.apply   # (the C++ stand-in: its gradients arrive without a Python wrapper, as a GEMM's do)
    F.linear = torch.nn.functional.linear = lambda x, w, b=None: no_gemm(x, w)
    try:
        dt_floor = timed(iters, floors)
        llm_qat_amd.stats(reset=True)
        dt = timed(iters)
        host_ms, gpu_ms = last["host"] * 1e3, last["gpu"] * 1e3
        st = llm_qat_amd.stats()
    finally:
        F.linear = torch.nn.functional.linear = real
    dt_gemm = timed(max(10, iters // 4))
    flops = 3 * 2.0 * tokens * wl.cols * wl.rows
    return {"ms_per_step": round(dt * 1e3, 4), "value": round(2 * wl.n / dt / 1e9, 2), "unit": "Gelem/s",
            "what": "QuantizeLinear(11008 -> 4096, W4 A8) forward + backward through the module on the step's tensors, F.linear replaced by a "
                    "no-launch stand-in: 1 pair forward launch + 1 pair backward launch (weight gradient in place), wall clock incl. Python / "
                    "autograd / allocator",
            "stats": {k: v for k, v in st.items() if k.startswith(("pair_", "single_", "inplace_", "cpp_"))},
            "host_node": llm_qat_amd.host_node(),   # which autograd node carried the operand pair: the C++ one (csrc/fq_autograd_node.cpp) or utils_quant's Python node
            "host_ms_per_step": round(host_ms, 4), "gpu_span_ms_per_step": round(gpu_ms, 4),
            "pytorch_floor_ms_per_step": round(dt_floor * 1e3, 4),
            "pytorch_floor_what": "the same loop with a plain module whose forward is only the no-launch stand-in (module call + one autograd Function + "
                                  "engine + two AccumulateGrad, no fake-quant): what PyTorch costs on this host before this library does anything",
            "with_gemm": {"ms_per_step": round(dt_gemm * 1e3, 4), "gemm_tflops_equiv": round(flops / dt_gemm / 1e12, 1),
                          "what": "the same call with the real F.linear (forward GEMM + dgrad + wgrad, hipBLASLt)"}}


def ensure_built(local_rank, dist):
    """A fresh checkout has no libllmqat_fakequant.so yet: compile it (one rank per node does, the others wait).
    Building the product is not a fallback -- without the library the benchmark fails loudly."""
    lib = os.path.join(ROOT, "llm-qat_amd", "libllmqat_fakequant.so")
    node = os.path.join(ROOT, "llm-qat_amd", "_fq_node.so")
    if not (os.path.exists(lib) and os.path.exists(node)) and local_rank == 0:
        import importlib.util
        spec = importlib.util.spec_from_file_location("_fq_build", os.path.join(ROOT, "llm-qat_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        print("bench.py: building the HIP library / the C++ autograd node (first run in this checkout)", file=sys.stderr)
        mod.build_extension()
        try:
            mod.build_node()
        except Exception as e:  # noqa: BLE001 -- host code only: without it the Python node carries the module path (api_path.host_node says which)
            print(f"bench.py: the C++ autograd node did not build ({e!r})", file=sys.stderr)
    if dist is not None:
        dist.barrier()


# ----------------------------------------------------------------------------------------------
# launching: N ranks, one process per GPU
# ----------------------------------------------------------------------------------------------
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--sustain-seconds", type=float, default=5.0,
                    help="after the timed region: run the same step back to back for this long and report the rate (0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip per-kernel / eager measurements")
    ap.add_argument("--core-extras", action="store_true",
                    help="per-kernel entries of the step's own kernels only (no model shapes / export / fused GEMM / eager / CPU): what "
                         "tools/profile_bench.sh runs under rocprofv3, so that every fq:: kernel in the trace has ONE launch shape per role")
    ap.add_argument("--model-shapes", action="store_true",
                    help="profiling mode: one fq:: launch shape per role (layer launches, export, W1/W2 one-launch, Asym A8), fixed order "
                         "and call count, manifest on stdout -- what tools/profile_bench.sh runs under rocprofv3 for per-entry PMC traffic")
    ap.add_argument("--no-sidecar", action="store_true", help="do not write bench_extras.json")
    ap.add_argument("--stub", action="store_true",
                    help="harness self-test: the step is a short sleep, no GPU is touched, the process group is gloo. "
                         "The JSON line says data='stub'; it is never a measurement.")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` with no launcher: start N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment, exactly what torch.distributed.run would set) from THIS process, which has not
    imported torch and never touches a GPU (no exec of a GPU-initialised process anywhere).  Rank 0's JSON line goes to
    our stdout.  Any rank failing -> the others are stopped (by PID) and we exit non-zero."""
    if not args.stub:
        ensure_built(0, None)   # hipcc only; no GPU call
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    live = set(range(len(procs)))
    while live:
        for i in list(live):
            code = procs[i].poll()
            if code is None:
                continue
            live.discard(i)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {i} exited with code {code}; stopping the other ranks", file=sys.stderr)
                for j in live:
                    procs[j].terminate()
        time.sleep(0.05)
    return rc


def init_ranks(args):
    """-> (world, rank, local_rank, dist or None, device or None, info).  Refuses (SystemExit != 0) rather than running
    fewer ranks than --gpus asks for."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("BENCH_TEST_KILL_RANK") == str(rank) and world > 1:   # tests/test_bench_harness.py: a rank that dies before joining
        raise SystemExit(3)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to report a line "
                         f"for a job that is not the one asked for")
    import torch
    info = {"ranks_seen": 1}
    device = None
    # the process group's backend is chosen up front, never by a per-rank fallback (ranks that disagree would hang):
    # RCCL ("nccl") as the launch contract says; BENCH_DIST_BACKEND=gloo keeps even the barrier off the GPUs
    backend = os.environ.get("BENCH_DIST_BACKEND", "gloo" if args.stub else "nccl")
    if not args.stub:
        ndev = torch.cuda.device_count()   # does not initialise the GPU on this image
        if ndev < 1:
            raise SystemExit("bench.py: no GPU visible")
        if world > ndev:
            if os.environ.get("BENCH_ALLOW_SHARED_GPU") != "1":
                raise SystemExit(f"bench.py: {world} ranks but only {ndev} GPU(s) visible. One process per GPU is the contract; "
                                 f"a rehearsal that shares GPUs needs BENCH_ALLOW_SHARED_GPU=1 (RCCL refuses two ranks on one device: "
                                 f"the timing barrier then falls back to gloo, or set BENCH_DIST_BACKEND=gloo)")
            info["ranks_share_devices"] = True
        device = torch.device("cuda", local_rank % ndev)
        torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "gloo":
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        tmo = datetime.timedelta(seconds=int(os.environ.get("BENCH_INIT_TIMEOUT_S", "300")))

        def join(bk, init_method=None):
            if init_method:   # the fallback's own rendezvous: rank 0 hosts it, whatever launcher started us
                dist.init_process_group(bk, init_method=init_method, rank=rank, world_size=world, timeout=tmo)
            else:
                dist.init_process_group(bk, timeout=tmo)
            t = torch.ones(1, device=device if bk == "nccl" else "cpu")
            dist.all_reduce(t)   # every rank that joined adds 1 (RCCL builds its communicator here: a failure shows up now)
            return int(t.item())

        try:
            info["ranks_seen"] = join(backend)
        except Exception as e:  # noqa: BLE001
            # The data path has no collective: the group only carries the timing barrier and a max over ranks.  If RCCL cannot
            # start on this node, that is no reason to lose the scaling point -- but the switch must be COLLECTIVE: the first
            # all-reduce fails or succeeds on every rank alike, and the gloo group meets on its own rendezvous (MASTER_PORT + 1),
            # so no rank can end up in a different backend.  An explicitly requested backend is never replaced.
            if backend != "nccl" or "BENCH_DIST_BACKEND" in os.environ:
                raise
            print(f"bench.py: rank {rank}: RCCL process group failed ({e!r}); every rank falls back to gloo for the timing barrier", file=sys.stderr)
            if dist.is_initialized():
                dist.destroy_process_group()
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            backend = "gloo"
            info["dist_backend_fallback"] = f"nccl failed: {e!r}"[:300]
            port = int(os.environ.get("MASTER_PORT", "29500")) + 1
            info["ranks_seen"] = join(backend, f"tcp://{os.environ.get('MASTER_ADDR', '127.0.0.1')}:{port}")
        info["dist_backend"] = backend
        if info["ranks_seen"] != args.gpus:
            raise SystemExit(f"bench.py: {info['ranks_seen']} ranks joined, --gpus {args.gpus} asked for")
    return world, rank, local_rank, dist, device, info


def run_stub(args, world, rank, dist, info):
    """the launch / barrier / max-over-ranks / aggregation path with a sleep for a step (tests/test_bench_harness.py)"""
    elems = 1000
    seconds = timed_region(lambda i: time.sleep(0.002 * (1 + rank)), args.steps, args.warmup, lambda: None, dist)
    out = {"metric": baseline_metric_name(), "value": aggregate_value(elems, args.steps, world, seconds), "unit": "Gelem/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": seconds / args.steps * 1e3, "timed_region_s": round(seconds, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "bf16", "data": "stub", "stub": True,
           "config": {"workload": "STUB (harness self-test: sleep step, no GPU) -- not a measurement", "parallelism": "replicas"}}
    out.update(info)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(compact_headline(out), flush=True)
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args, argv)   # this process stays torch-free

    world, rank, local_rank, dist, device, info = init_ranks(args)
    if args.stub:
        return run_stub(args, world, rank, dist, info)
    import torch

    ensure_built(local_rank, dist)
    wl = Workload(device)
    wl.prime_bounds()
    if args.model_shapes:
        return run_model_shapes(args, wl, rank, dist)
    PROFILE_MANIFEST.append(("prologue: prime_bounds", 2 * wl.nsets))
    # THE HEADLINE STEP writes BOTH gradients to fresh tensors (the reference's grad_output.clone(), utils_quant.py:83-87): the work the
    # metric names -- 4 B/elem forward + 4 B/elem backward actually moved for each of the two tensors, 10 B/elem in SURVEY §8d's
    # accounting.  (Round 3 put the product's default step here, whose W4 backward moves nothing because the weight's gradient is
    # handed on by reference: its 966 Gelem/s implied 9.7 TB/s under the 10 B/elem accounting -- above the chip's peak.  That step is
    # still timed, right after, under the same K / W / barrier / max-over-ranks protocol, as `value_product_default`.)
    seconds = timed_region(wl.step_out_of_place, args.steps, args.warmup, torch.cuda.synchronize, dist)
    PROFILE_MANIFEST.append(("timed region: step (both gradients written)", 2 * (args.steps + args.warmup)))
    PROFILE_MANIFEST.append(("timed region: product-default step (weight gradient in place)", 2 * (args.steps + args.warmup)))
    seconds_pd = timed_region(wl.step, args.steps, args.warmup, torch.cuda.synchronize, dist)
    elems_step = 2 * wl.n
    value = aggregate_value(elems_step, args.steps, world, seconds)
    sus_steps, sus_s, sus_chunks = sustained_region(wl.step_out_of_place, 0.0 if args.no_extras else args.sustain_seconds, torch.cuda.synchronize)
    if sus_steps:
        PROFILE_MANIFEST.append(("sustained: step (both gradients written)", 2 * sus_steps))
    ms_step = seconds / args.steps * 1e3
    ms_step_pd = seconds_pd / args.steps * 1e3
    clip_frac = wl.clippable_fraction()
    mask_bytes_a = int(wl.n * clip_frac) // 8          # 1 bit/element for the A8 tensor's clippable rows; the W4 tensor has none
    algo_bytes_step = elems_step * (FWD_BYTES_PER_ELEM + BWD_BYTES_PER_ELEM)
    # bytes the step's two launches move: forward 4 B/elem (+ mask bits); backward 4 B/elem per tensor (+ mask bits).  Product default:
    # NOTHING for the W4 tensor's backward, whose gradient is handed on by reference (in place, no row can clip)
    moved_bytes_step_pd = elems_step * FWD_BYTES_PER_ELEM + wl.n * BWD_MASK_BYTES_PER_ELEM + 2 * mask_bytes_a
    moved_bytes_step = moved_bytes_step_pd + wl.n * BWD_MASK_BYTES_PER_ELEM

    out = {
        "metric": baseline_metric_name(),
        "value": round(value, 2), "unit": "Gelem/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_step, 4), "timed_region_s": round(seconds, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "SymQuantizer fwd + STE bwd, W4 on weight-style [4096,11008] + A8 on activation-style "
                               "[4096,11008], bf16, clip [-2,2] (LLaMA-7B W4-A8 down_proj shapes, configs[1])",
                   "elements_per_step": elems_step, "buffer_sets": wl.nsets, "parallelism": "replicas" if world > 1 else "1gpu",
                   "semantics": "cpu_eager (passed explicitly through the C ABI: the parity gate's oracle is the CPU restatement; the step's tensors are nowhere near the rows where the two scalar policies differ)",
                   "backward": "mask (forward records row bounds + 1-bit STE mask; backward does not re-read x); BOTH gradients are written "
                               "to fresh tensors (the reference's grad_output.clone()); value_product_default = the same step with the "
                               "weight's gradient masked in place (gx == g: rows that cannot clip -- all of a weight's -- are not touched)",
                   "launches_per_step": "2 (weight + input of a QuantizeLinear share one forward and one backward launch)"},
        # the product's default data flow (utils_quant.py point 6): the weight's gradient handed on by reference.  NOT the headline: its
        # backward touches half the elements, so elements/s on it is not comparable with the 10 B/elem accounting
        "value_product_default": round(aggregate_value(elems_step, args.steps, world, seconds_pd), 2),
        "ms_per_step_product_default": round(ms_step_pd, 4),
        # elements whose gradient the timed backward reads + writes
        "backward_elements_touched": 2 * wl.n, "backward_elements_touched_product_default": wl.n,
        "hbm_gbs_moved": round(moved_bytes_step / (ms_step * 1e-3) / 1e9 * world, 1),
        "hbm_gbs_moved_product_default": round(moved_bytes_step_pd / (ms_step_pd * 1e-3) / 1e9 * world, 1),
        "hbm_gbs_note": "bytes the step's two launches move (fwd 4 + bwd 4 B/elem + mask bits) / ms_per_step: a real byte rate",
    }
    out.update(info)
    if sus_steps:
        cs = sorted(sus_chunks)
        out["sustained"] = {"value": round(elems_step * sus_steps / sus_s / 1e9, 2), "unit": "Gelem/s", "seconds": round(sus_s, 2), "steps": sus_steps,
                            "ms_per_step_min_median_max": [round(cs[0], 4), round(cs[len(cs) // 2], 4), round(cs[-1], 4)],
                            "ms_per_step_first_last_chunk": [round(sus_chunks[0], 4), round(sus_chunks[-1], 4)],
                            "what": "rank 0, the headline step back to back for `seconds` (1000-step chunks, one sync each), after the timed "
                                    "region: steady-state rate per GPU; not the headline"}

    if rank == 0 and not args.no_extras:
        traffic, tsrc = load_traffic()
        it = max(MIN_KERNEL_ITERS, min(args.steps, 1000))   # >= 200 launches per kernel whatever --steps says (time_launches enforces it too)
        nb = wl.n
        fwd_w, fwd_a = nb * FWD_BYTES_PER_ELEM, nb * FWD_BYTES_PER_ELEM + mask_bytes_a
        bwd_w, bwd_a = nb * BWD_MASK_BYTES_PER_ELEM, nb * BWD_MASK_BYTES_PER_ELEM + mask_bytes_a
        # (fn, algorithmic bytes, bytes moved by design)
        ks = {
            "sym_fwd_w4": (lambda s: wl.fwd(s, "w"), nb * FWD_BYTES_PER_ELEM, fwd_w),
            "sym_fwd_a8": (lambda s: wl.fwd(s, "a"), nb * FWD_BYTES_PER_ELEM, fwd_a),
            "ste_bwd_a8": (lambda s: wl.bwd(s, "a"), nb * BWD_BYTES_PER_ELEM, bwd_a),
            "ste_bwd_w4": (lambda s: wl.bwd(s, "w"), nb * BWD_BYTES_PER_ELEM, bwd_w),
        }
        pk = {
            "sym_fwd_pair_w4a8": (lambda s: wl.fwd_pair(s), 2 * nb * FWD_BYTES_PER_ELEM, fwd_w + fwd_a),
            "ste_bwd_pair_w4a8 (weight gradient in place)": (lambda s: wl.bwd_pair(s), 2 * nb * BWD_BYTES_PER_ELEM, bwd_a),
        }

        def pair_traffic(k):
            if k in traffic:
                return traffic[k]
            if "in place" in k:
                return traffic.get("ste_bwd_pair_w4a8_inplace")
            parts = {"sym_fwd_pair_w4a8": ("sym_fwd_w4", "sym_fwd_a8")}.get(k)
            return sum(traffic[p] for p in parts) if parts and all(p in traffic for p in parts) else None

        out["kernels_step"] = [roofline_entry(k, b, wl.time_kernel(fn, it, name=k), pair_traffic(k), moved_bytes=mv, traffic_source=tsrc)
                               for k, (fn, b, mv) in pk.items()]
        out["kernels"] = [roofline_entry(k, b, wl.time_kernel(fn, it, name=k), traffic.get(k), moved_bytes=mv, traffic_source=tsrc)
                          for k, (fn, b, mv) in ks.items()]
        # the reference's data flow on the same kernels' siblings: forward without mask/bounds, backward re-reading x
        alt = {
            "sym_fwd_w4_plain": (lambda s: wl.fwd_plain(s, "w"), nb * FWD_BYTES_PER_ELEM),
            "ste_bwd_a8_xread": (lambda s: wl.bwd_xread(s, "a"), nb * BWD_BYTES_PER_ELEM),
        }
        out["kernels_reference_dataflow"] = [roofline_entry(k, b, wl.time_kernel(fn, it, name=k), traffic.get(k), traffic_source=tsrc) for k, (fn, b) in alt.items()]
        # the backward as a copy (gx != g for the weight too: what round 1 measured) and the weight's in-place launch alone
        out["kernels_step_out_of_place"] = [roofline_entry("ste_bwd_pair_w4a8 (both gradients to fresh tensors)", 2 * nb * BWD_BYTES_PER_ELEM,
                                                           wl.time_kernel(lambda s: wl.bwd_pair(s, inplace_w=False), it, name="ste_bwd_pair_w4a8 (both gradients to fresh tensors)"),
                                                           traffic.get("ste_bwd_pair_w4a8 (both gradients to fresh tensors)", traffic.get("ste_bwd_pair_w4a8")),
                                                           moved_bytes=bwd_w + bwd_a, traffic_source=tsrc)]
        tw = wl.time_kernel(lambda s: wl.bwd(s, "w", inplace=True), it, name="ste_bwd_w4_in_place")
        out["ste_bwd_w4_in_place"] = {"us_per_launch": round(tw[0] * 1e3, 2), "us_p10_p50_p90": [round(v * 1e3, 2) for v in tw[1]],
                                      "what": "fq_ste_bwd_mask with gx == g on the W4 tensor: every row's bounds prove nothing clips; one block per 256 "
                                              "rows reads their bounds (8 B per row) and exits"}
        ac = {
            "sym_fwd_w4_autocast_bf16_out": (lambda s: wl.fwd_autocast(s, "w", False), nb * FWD_BYTES_PER_ELEM),
            "sym_fwd_a8_autocast_bf16_out": (lambda s: wl.fwd_autocast(s, "a", False), nb * FWD_BYTES_PER_ELEM),
            "sym_fwd_a8_autocast_fp32_out": (lambda s: wl.fwd_autocast(s, "a", True), nb * 6),  # read 2 + write 4 B/elem
        }
        out["kernels_autocast_arithmetic"] = [roofline_entry(k, b, wl.time_kernel(fn, it, name=k), traffic.get(k), traffic_source=tsrc) for k, (fn, b) in ac.items()]
        # `roofline`: the forward launch of the step (row_reg_kernel over the W4 weight and the A8 input: reduce -> scale ->
        # round -> dequant): achieved = ALGORITHMIC bytes (4 B/elem x 90.2 M elements) / its launch time.
        fwp = out["kernels_step"][0]
        alg = fwp["algorithmic_bytes_per_launch"] / (fwp["us_per_launch"] * 1e-6) / 1e9
        out["roofline"] = {"bound": "hbm", "achieved": round(alg, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / HBM_PEAK_GBS, 4),
                           "traffic": fwp["traffic"], "us_per_launch": fwp["us_per_launch"],
                           "kernel": "row_reg_kernel (Sym forward of the step: W4 weight + A8 input in one launch)",
                           "algorithmic_bytes_per_launch": fwp["algorithmic_bytes_per_launch"]}
        for k in ("traffic_gbs", "traffic_frac", "traffic_source"):
            if k in fwp:
                out["roofline"][k] = fwp[k]
        tot_us_pd = sum(e["us_per_launch"] for e in out["kernels_step"])
        tot_us = out["kernels_step"][0]["us_per_launch"] + out["kernels_step_out_of_place"][0]["us_per_launch"]
        # the credit figure for the whole step: BYTES MOVED / time (never the 10 B/elem accounting, which can exceed the peak)
        out["roofline_step"] = {"bound": "hbm", "achieved": round(moved_bytes_step / (tot_us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": round(moved_bytes_step / (tot_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                "bytes_moved_per_step": moved_bytes_step, "us_per_step_launches": round(tot_us, 2),
                                "product_default": {"bytes_moved_per_step": moved_bytes_step_pd, "us_per_step_launches": round(tot_us_pd, 2),
                                                    "achieved": round(moved_bytes_step_pd / (tot_us_pd * 1e-6) / 1e9, 1),
                                                    "frac": round(moved_bytes_step_pd / (tot_us_pd * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)},
                                "reference_dataflow_bytes_per_step": algo_bytes_step,
                                "what": "both launches of the headline step (both gradients written): bytes moved / sum of the launches' "
                                        "times; product_default: the step with the weight's gradient in place"}
        out["unpaired_step"] = {"ms_per_step": None if args.core_extras else round(timed_region(wl.step_unpaired, it, 5, torch.cuda.synchronize) / it * 1e3, 4),
                                "what": "the same step as four single-tensor launches (fq_sym_fwd_train x2, fq_ste_bwd_mask x2; weight gradient in place)"}
        if not args.core_extras:
            out["autograd_path"] = autograd_path(wl)
            out["api_path"] = api_path(wl)
        # a live yardstick for "how fast can this device move the same bytes": ATen's device-to-device copy of the W tensor
        # (read 90.2 MB + write 90.2 MB = one single-tensor forward's algorithmic bytes), timed like the kernels above
        cmean, cpct = wl.time_kernel(lambda s: s["yw"].copy_(s["w"]), it, name="copy_reference (ATen)", fq=0)
        if args.core_extras:
            out["profile_manifest"] = [list(m) for m in PROFILE_MANIFEST]   # every fq:: launch of this process, in order
        out["copy_reference"] = {"what": "torch Tensor.copy_ device-to-device over the same 180.4 MB as sym_fwd_w4 (compare kernels[0])",
                                 "us_per_launch": round(cmean * 1e3, 2), "gbs": round(nb * FWD_BYTES_PER_ELEM / (cmean * 1e-3) / 1e9, 1),
                                 "us_p10_p50_p90": [round(v * 1e3, 2) for v in cpct]}
        # the launches of one LLaMA-7B layer at their real shapes
        if not args.core_extras:
            mtraffic, msrc = load_traffic("traffic_model_shapes.json")
            out["kernels_model_shapes"] = time_specs(torch, ModelShapes(wl).specs(), it, mtraffic, msrc)
            out["kernels_model_shapes_13b"] = ModelShapes(wl, hidden=5120, inter=13824).entries_compact(max(10, it // 2), "13B")
            for hook in EXTRA_ENTRIES:   # further kernel families register here (export, fused QuantizeLinear, W1/W2, Asym)
                try:
                    out.update(hook(wl, it, mtraffic, msrc))
                except Exception as e:  # noqa: BLE001  -- an extra entry must never lose the headline line
                    out.setdefault("extras_failed", []).append(f"{getattr(hook, '__name__', hook)}: {e!r}")
        if world == 1 and not args.core_extras:
            out["gpu_eager"] = gpu_eager(wl)
            out["gpu_eager_autocast"] = gpu_eager(wl, autocast=True)
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
                out["cpu_baseline_c_port"] = cpu_c_port()
                out["cpu_baseline"]["parity_gate"] = parity_gate(wl)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit(out, args)
    return 0


# ----------------------------------------------------------------------------------------------
# output: the LAST stdout line is a compact headline the driver can parse from an 8 KB tail; everything else goes before it
# ----------------------------------------------------------------------------------------------
HEADLINE_MAX_BYTES = 4096
_ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "us_per_launch", "kernel", "algorithmic_bytes_per_launch", "traffic_source")
_HEAD_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "timed_region_s", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "value_product_default", "ms_per_step_product_default", "backward_elements_touched", "hbm_gbs_moved", "ranks_seen", "dist_backend",
              "dist_backend_fallback", "ranks_share_devices", "stub", "extras_failed")


def _clip(s, n):
    s = str(s)
    return s if len(s) <= n else s[: n - 1] + "…"


def compact_headline(out, extras_file=None):
    """The one line the driver parses: the contract's keys + roofline + roofline_step + cpu_baseline, hard-capped at
    HEADLINE_MAX_BYTES (the driver keeps an 8 KB stdout tail; round 2's 27 KB line was cut and never parsed)."""
    h = {k: out[k] for k in _HEAD_KEYS if k in out}
    cfg = out.get("config", {})
    h["config"] = {"workload": _clip(cfg.get("workload", ""), 170), "parallelism": cfg.get("parallelism"),
                   "backward": "both gradients written to fresh tensors; value_product_default = weight grad in place, activation grad copied"}
    if "elements_per_step" in cfg:
        h["config"]["elements_per_step"] = cfg["elements_per_step"]
    r = out.get("roofline")
    if r:
        h["roofline"] = {k: (_clip(r[k], 150) if k in ("kernel", "traffic_source") else r[k]) for k in _ROOFLINE_KEYS if k in r}
    rs = out.get("roofline_step")
    if rs:
        h["roofline_step"] = {k: rs[k] for k in ("bound", "achieved", "peak", "unit", "frac", "bytes_moved_per_step", "us_per_step_launches", "product_default") if k in rs}
    c = out.get("cpu_baseline")
    if c:
        h["cpu_baseline"] = {k: (_clip(c[k], 200) if k == "sample" else c[k])
                             for k in ("value", "unit", "cores", "kind", "sample", "host_cpu", "seconds_best", "parity_gate") if k in c}
    ap = out.get("api_path")
    if ap:
        h["api_path_gelem_s"] = ap.get("value")
    su = out.get("sustained")
    if su:
        h["sustained_per_gpu_gelem_s"], h["sustained_seconds"] = su["value"], su["seconds"]
    sc = self_check(out)
    if sc:
        h["self_check"] = sc
    ge = out.get("gpu_eager")
    if ge:
        h["gpu_eager_gelem_s"] = ge.get("value")
        if ge.get("value"):
            h["speedup_vs_gpu_eager"] = round(out["value"] / ge["value"], 2)
    if extras_file:
        h["extras"] = extras_file
    line = json.dumps(h)
    if len(line.encode()) >= HEADLINE_MAX_BYTES:   # cannot happen with the clips above; if it ever does, shed the optional parts, never the contract
        for k in ("gpu_eager_gelem_s", "speedup_vs_gpu_eager", "api_path_gelem_s", "sustained_per_gpu_gelem_s", "sustained_seconds", "extras", "self_check", "roofline_step", "cpu_baseline"):
            h.pop(k, None)
            line = json.dumps(h)
            if len(line.encode()) < HEADLINE_MAX_BYTES:
                break
    return line


def emit(out, args):
    """full record -> bench_extras.json (next to bench.py, and gpurun_out/ when that exists) + one STDERR line per extras family;
    stdout carries exactly ONE line, the compact headline (the launch contract: "rank 0 prints ONE JSON line")."""
    head_keys = set(_HEAD_KEYS) | {"config", "roofline", "roofline_step", "cpu_baseline", "hbm_gbs_note"}
    extras_file = None
    if not getattr(args, "no_sidecar", False):
        for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
            try:
                if os.path.isdir(d):
                    with open(os.path.join(d, "bench_extras.json"), "w") as f:
                        json.dump(out, f, indent=1)
                    extras_file = extras_file or "bench_extras.json (next to bench.py; the same families are printed on stderr)"
            except OSError:
                pass
    for k, v in out.items():
        if k not in head_keys:
            print(json.dumps({"bench_extras": k, "data": v}), file=sys.stderr, flush=True)
    print(compact_headline(out, extras_file), flush=True)


def run_model_shapes(args, wl, rank, dist):
    """Profiling mode (tools/profile_bench.sh): every launch kind of the layer, the export kernels, the one-launch W1/W2 kernel and
    Asym A8, ONE fq:: kernel per call, in a fixed order, each for the same number of calls -- so a rocprofv3 kernel trace / PMC pass
    of this command splits into one segment per entry by counting fq:: dispatches (tools/summarize_profile.py).  Prints a manifest
    (entry -> calls, bytes) with the HIP-event timings; it is not the headline measurement."""
    torch = wl.torch
    iters = max(10, min(args.steps, 50))
    specs = ModelShapes(wl).specs() + export_specs(wl) + lowbit_asym_specs(wl, fused_only=True)
    mtraffic, msrc = load_traffic("traffic_model_shapes.json")
    entries = time_specs(torch, specs, iters, mtraffic, msrc)
    out = {"mode": "model-shapes", "prologue_fq_launches": 2 * wl.nsets,
           "entries": entries, "order": [sp["name"] for sp in specs],
           "profile_manifest": [["prologue: prime_bounds", 2 * wl.nsets]] + [list(m) for m in PROFILE_MANIFEST]}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        for d in (os.path.join(ROOT, "gpurun_out"),):
            if os.path.isdir(d):
                json.dump(out, open(os.path.join(d, "bench_model_shapes.json"), "w"), indent=1)
        print(json.dumps(out), flush=True)
    return 0


def export_specs(wl):
    """packed-bin export and the scale pre-pass on the metric tensor (bf16 [4096,11008])"""
    torch, L, lib_, st = wl.torch, wl.L, wl._lib, wl.stream
    rows, cols, n = wl.rows, wl.cols, wl.n
    code = lib_.DTYPE_BF16
    for s in wl.sets:
        if "bins" not in s:
            s["bins"] = torch.empty(n * 2, dtype=torch.uint8, device=wl.device)
            s["scales"] = torch.empty(rows, 2, device=wl.device)
            s["over"] = torch.empty(rows, dtype=torch.int32, device=wl.device)

    def chk(rc):
        if rc:
            lib_.check(rc, "export")

    def exp(key, bits, cont):
        return lambda s: chk(L.fq_sym_export(s[key].data_ptr(), s["bins"].data_ptr(), s["scales"].data_ptr(), s["over"].data_ptr(), rows, cols, bits,
                                             cont, code, 0, 0, st))

    def scales(key, bits):
        return lambda s: chk(L.fq_sym_row_scales(s[key].data_ptr(), s["scales"].data_ptr(), rows, cols, bits, code, 0, 0, -2.0, 2.0, None, None, 0, st))

    site = "utils_quant.py:71-72 (the bins `torch.round(input * s)`)"
    ks = [("sym_export_w4_int4", exp("w", 4, lib_.BINS_INT4), n * 2 + n // 2),
          ("sym_export_w8_int8", exp("w", 8, lib_.BINS_INT8), n * 3),
          ("sym_export_a8_int8", exp("a", 8, lib_.BINS_INT8), n * 3),
          ("sym_row_scales_w4 (pre-pass: read only)", scales("w", 4), n * 2)]
    return [dict(name=k, site=site, fn=fn, sets=wl.sets, algo=b, moved=b) for k, fn, b in ks]


def export_entries(wl, iters, traffic=None, tsrc=None):
    return {"kernels_export": time_specs(wl.torch, export_specs(wl), iters, traffic, tsrc)}


MFMA_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak, MI355X (/opt/skills/guides/MI355X_MICROARCH.md)


def qlinear_entries(wl, iters, traffic=None, tsrc=None):
    """QuantizeLinear's no-grad forward for down_proj on its named shape, x[2048,11008] . W[4096,11008]^T, W4 A8, as the PRODUCT runs
    it: one fq pair launch (fake-quant at the HBM roofline) + F.linear (hipBLASLt).  MFMA roofline (dense bf16 peak) for the whole
    module forward.  The quantize-on-load GEMM experiment that lost against this lives in tools/qlinear/ with its
    own bench (tools/qlinear/qlinear_bench.py)."""
    import torch.nn.functional as F
    torch, L, lib_, st = wl.torch, wl.L, wl._lib, wl.stream
    m, k, n = 2048, wl.cols, wl.rows
    code = lib_.DTYPE_BF16
    sets = [dict(w=s["w"], x=s["a"][:m], wq=s["yw"], xq=s["ya"][:m]) for s in wl.sets[:3]]

    def pair(s):
        rc = L.fq_sym_fwd_pair(s["w"].data_ptr(), s["wq"].data_ptr(), n, 4, None, None, 0, s["x"].data_ptr(), s["xq"].data_ptr(), m, 8, None, None, 0,
                               k, code, 0, 0, -2.0, 2.0, st)
        if rc:
            lib_.check(rc, "fq_sym_fwd_pair")

    for s in sets:
        pair(s)
    flops = 2.0 * m * k * n
    kinds = [
        ("product: fq pair launch + F.linear (hipBLASLt)", lambda s: (pair(s), F.linear(s["xq"], s["wq"]))),
        ("F.linear alone", lambda s: F.linear(s["xq"], s["wq"])),
    ]
    out = []
    for name, fn in kinds:
        ms, pct = time_launches(torch, fn, max(10, iters // 4), sets)
        tf = flops / (ms * 1e-3) / 1e12
        out.append({"kernel": name, "bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS, 4),
                    "us_per_launch": round(ms * 1e3, 2), "us_p10_p50_p90": [round(v * 1e3, 2) for v in pct], "flops": flops, "traffic": None})
    return {"qlinear_down_proj": {"shape": f"x[{m},{k}] . W[{n},{k}]^T, W4 A8, bf16, no-grad forward", "entries": out}}


def lowbit_asym_specs(wl, fused_only=False):
    """The 1-/2-bit weight branches on [4096,11008] bf16 (utils_quant.py:202-242) -- the default (ATen abs + mean, then
    fq_w12_fwd: 3 launches) and the opt-in one-launch kernel -- and AsymQuantizer A8 on [2048,11008].
    fused_only: leave out the entries that launch ATen kernels too (profiling mode: one fq:: launch per call)."""
    torch, L, lib_, st = wl.torch, wl.L, wl._lib, wl.stream
    rows, cols, n = wl.rows, wl.cols, wl.n
    code = lib_.DTYPE_BF16
    for s in wl.sets:
        s.setdefault("sc16", torch.empty(rows, dtype=torch.bfloat16, device=wl.device))

    def chk(rc):
        if rc:
            lib_.check(rc, "w12/asym")

    def w12_default(bits):
        def fn(s):
            sc = s["w"].abs().mean(dim=1, keepdim=True)
            if bits == 2:
                sc = 2 * sc
            chk(L.fq_w12_fwd(s["w"].data_ptr(), sc.data_ptr(), s["yw"].data_ptr(), rows, cols, bits, 1, code, st))
        return fn

    def w12_fused(bits):
        return lambda s: chk(L.fq_w12_fwd_rows(s["w"].data_ptr(), s["yw"].data_ptr(), s["sc16"].data_ptr(), rows, cols, bits, code, st))

    t2 = 2048

    def asym(s):
        chk(L.fq_asym_fwd_train(s["a"].data_ptr(), s["ya"].data_ptr(), t2, cols, 8, code, 0, -2.0, 2.0, s["ba"].data_ptr(), s["ma"].data_ptr(), wl.mask_bytes, st))

    ks = [("w12 1-bit as three launches: ATen abs+mean + fq_w12_fwd (~10 B/elem moved; the default until round 3)", w12_default(1), n * 4, n * 10),
          ("w12 2-bit as three launches: ATen abs+mean + fq_w12_fwd (~10 B/elem moved; the default until round 3)", w12_default(2), n * 4, n * 10),
          ("w12 1-bit one launch (product default: row mean in ATen's summation order)", w12_fused(1), n * 4, n * 4),
          ("w12 2-bit one launch (product default: row mean in ATen's summation order)", w12_fused(2), n * 4, n * 4),
          ("asym_fwd_a8 bf16 [2048,11008] (training mode)", asym, t2 * cols * 4, t2 * cols * 4 + t2 * cols // 8)]
    if fused_only:
        ks = ks[2:]
    site = "utils_quant.py:202-242 (W1/W2), :110-149 (Asym)"
    return [dict(name=k, site=site, fn=fn, sets=wl.sets, algo=algo, moved=moved) for k, fn, algo, moved in ks]


def lowbit_asym_entries(wl, iters, traffic=None, tsrc=None):
    ents = time_specs(wl.torch, lowbit_asym_specs(wl), iters, traffic, tsrc)
    # does the one-launch 1-bit kernel's "redo with the reference chain" branch (fq_kernels.h w12_row_aten_kernel: a wave
    # whose ballot finds a bf16 quotient |w / sc| < 2^-100 with w != 0) ever fire on the bench tensor?  Counted here on the tensor itself.
    torch = wl.torch
    w = wl.sets[0]["w"].float()
    sc = w.abs().mean(dim=1, keepdim=True)
    redo = int(((w != 0) & ((w / sc).abs() < 2.0 ** -100)).sum())
    for e in ents:
        if e["kernel"].startswith("w12 1-bit one launch"):
            e["redo_branch_elements_on_this_tensor"] = redo
            e["redo_branch_note"] = ("0 => the kernel's redo path cannot run here: the 36.4 vs 32.2 us of round 4's two runs was run-to-run spread of one "
                                     "200-launch batch mean, not a data-dependent path; us_per_launch is the bracketed p50 since round 5")
    return {"kernels_lowbit_asym": ents}


EXTRA_ENTRIES = [export_entries, qlinear_entries, lowbit_asym_entries]


if __name__ == "__main__":
    sys.exit(main())
