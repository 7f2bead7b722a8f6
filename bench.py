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

Extra objects on the JSON line:
  roofline      dominant kernel (STE backward): algorithmic bytes / live HIP-event launch time vs 8 TB/s
  kernels       the same for every kernel of the step
  cpu_baseline  the reference's CPU path (eager op chain, oracle/eager_chain.py) timed on this box's host
                cores on a bounded sample -- rank 0, N=1 only
  gpu_eager     the reference's eager op chain run on this GPU (9+5 launches): the like-for-like "before"
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ROWS, COLS = 4096, 11008
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
FWD_BYTES_PER_ELEM = 4   # bf16: read x + write y              
BWD_BYTES_PER_ELEM = 6   # bf16: read g + read x + write gx


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


def aggregate_value(elems_per_rank_step, steps, world, seconds):
    """whole-job throughput in Gelem/s: all ranks' elements / max-over-ranks time"""
    return elems_per_rank_step * steps * world / seconds / 1e9


# ----------------------------------------------------------------------------------------------
# GPU workload
# ----------------------------------------------------------------------------------------------
class Workload:
    def __init__(self, device, rows=ROWS, cols=COLS, nsets=4, seed=1234):
        import torch
        import llm_qat_amd
        from llm_qat_amd import _lib
        _lib.lib()  # no fallback: raises if the HIP library is absent
        self.torch, self.ops = torch, llm_qat_amd.ops
        self.rows, self.cols, self.nsets = rows, cols, nsets
        self.n = rows * cols
        self.sets = []
        for k in range(nsets):
            # BASELINE.md §3 inputs: W ~ N(0, 0.02^2) seed 1234, g ~ N(0,1)*1e-3 seed 1235, activations ~ N(0,1) with
            # 0.1 % of entries x20 seed 1236 (set 0 uses exactly those seeds; the rotating sets offset them by 1000*k)
            gw_, gg_, ga_ = (torch.Generator(device=device).manual_seed(sd + 1000 * k) for sd in (seed, seed + 1, seed + 2))
            w = (torch.randn(rows, cols, generator=gw_, device=device) * 0.02).bfloat16()
            a = torch.randn(rows, cols, generator=ga_, device=device)
            a[torch.rand(rows, cols, generator=ga_, device=device) < 1e-3] *= 20.0                # outlier channels
            a = a.bfloat16()
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

    def bwd(self, s, leg):
        g, gx, b, m = (s["gw"], s["gxw"], s["bw"], s["mw"]) if leg == "w" else (s["ga"], s["gxa"], s["ba"], s["ma"])
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
            rc = self.L.fq_sym_fwd_autocast(x.data_ptr(), self.y32.data_ptr(), self.rows, self.cols, bits, self._lib.DTYPE_BF16, 1, -2.0, 2.0,
                                            b.data_ptr(), None, 0, None, 0, self.stream)
        else:
            rc = self.L.fq_sym_fwd_autocast(x.data_ptr(), y.data_ptr(), self.rows, self.cols, bits, self._lib.DTYPE_BF16, 0, -2.0, 2.0,
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

    def bwd_pair(self, s):
        rc = self.L.fq_ste_bwd_mask_pair(s["gw"].data_ptr(), s["gxw"].data_ptr(), self.rows, s["bw"].data_ptr(), s["mw"].data_ptr(),
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

    def step_unpaired(self, i):
        sf = self.sets[i % self.nsets]
        sb = self.sets[(i + self.nsets - 2) % self.nsets]
        self.fwd(sf, "w")
        self.fwd(sf, "a")
        self.bwd(sb, "a")
        self.bwd(sb, "w")

    def prime_bounds(self):
        for s in self.sets:
            self.fwd(s, "w")
            self.fwd(s, "a")
        self.torch.cuda.synchronize()

    def time_kernel(self, fn, iters):
        """Launch duration of one kernel kind with HIP events on the launch stream, rotating buffers.
        -> (mean ms over a back-to-back batch, [p10, p50, p90] ms of individually bracketed launches)"""
        torch = self.torch
        for i in range(3):
            fn(self.sets[i % self.nsets])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(iters):
            fn(self.sets[i % self.nsets])
        e1.record()
        torch.cuda.synchronize()
        mean = e0.elapsed_time(e1) / iters
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for i, (a, b) in enumerate(pairs):
            a.record()
            fn(self.sets[i % self.nsets])
            b.record()
        torch.cuda.synchronize()
        d = sorted(a.elapsed_time(b) for a, b in pairs)
        return mean, [d[len(d) // 10], d[len(d) // 2], d[(9 * len(d)) // 10]]


def baseline_metric_name():
    """BASELINE.json's metric string, verbatim (value is its first quantity, Gelem/s; the second, achieved HBM GB/s,
    is `hbm_gbs_algorithmic` and the roofline objects)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:  # noqa: BLE001
        return "fake-quant fwd+bwd Gelem/s & achieved HBM GB/s, 4096\u00d711008 bf16 W4A8"


def roofline_entry(name, bytes_per_launch, timing, traffic=None):
    """achieved = ALGORITHMIC bytes (4 B/elem forward, 6 B/elem backward, bf16) / launch time.
    `traffic` = HBM bytes per launch measured with rocprofv3 --pmc (profiles/traffic.json); where the kernel moves
    fewer bytes than the algorithmic figure (backward that does not re-read x) `frac` exceeds the byte-rate it
    actually sustains, which is reported separately as traffic_gbs / traffic_frac."""
    ms, pct = timing if isinstance(timing, tuple) else (timing, None)
    ach = bytes_per_launch / (ms * 1e-3) / 1e9
    e = {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "us_per_launch": round(ms * 1e3, 2),
         "algorithmic_bytes_per_launch": bytes_per_launch}
    if traffic:
        e["traffic_gbs"] = round(traffic / (ms * 1e-3) / 1e9, 1)
        e["traffic_frac"] = round(traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    if pct:
        e["us_p10_p50_p90"] = [round(v * 1e3, 2) for v in pct]
    return e


def load_traffic():
    """HBM bytes per launch from committed rocprofv3 --pmc runs (profiles/traffic.json), if present."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p))
        except Exception:
            return {}
    return {}


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


def ensure_built(local_rank, dist):
    """A fresh checkout has no libllmqat_fakequant.so yet: compile it (one rank per node does, the others wait).
    Building the product is not a fallback -- without the library the benchmark fails loudly."""
    lib = os.path.join(ROOT, "llm-qat_amd", "libllmqat_fakequant.so")
    if not os.path.exists(lib) and local_rank == 0:
        import importlib.util
        spec = importlib.util.spec_from_file_location("_fq_build", os.path.join(ROOT, "llm-qat_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        print("bench.py: building the HIP library (first run in this checkout)", file=sys.stderr)
        mod.build_extension()
    if dist is not None:
        dist.barrier()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-extras", action="store_true", help="skip per-kernel / eager measurements")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    ndev = max(torch.cuda.device_count(), 1)
    device = torch.device("cuda", (local_rank % ndev) if world > 1 else 0)
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # The data path has no collective (replicas only).  The process group exists for the timing barrier and the
        # max-over-ranks: RCCL ("nccl") as the launch contract says; BENCH_DIST_BACKEND=gloo keeps even that off the GPUs.
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
        try:
            dist.init_process_group(backend)
            t = torch.zeros(1, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(t)
        except Exception as e:  # noqa: BLE001
            if backend != "nccl":
                raise
            print(f"bench.py: RCCL process group failed ({e!r}); using gloo for the timing barrier", file=sys.stderr)
            if dist.is_initialized():
                dist.destroy_process_group()
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            dist.init_process_group("gloo")
    elif args.gpus > 1:
        print(f"bench.py: --gpus {args.gpus} needs the torch.distributed.run launcher (WORLD_SIZE unset); running 1 rank",
              file=sys.stderr)

    ensure_built(local_rank, dist)
    wl = Workload(device)
    wl.prime_bounds()
    seconds = timed_region(wl.step, args.steps, args.warmup, torch.cuda.synchronize, dist)
    elems_step = 2 * wl.n
    value = aggregate_value(elems_step, args.steps, world, seconds)
    ms_step = seconds / args.steps * 1e3
    algo_bytes_step = elems_step * (FWD_BYTES_PER_ELEM + BWD_BYTES_PER_ELEM)

    out = {
        "metric": baseline_metric_name(),
        "value": round(value, 2), "unit": "Gelem/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "SymQuantizer fwd + STE bwd, W4 on weight-style [4096,11008] + A8 on activation-style "
                               "[4096,11008], bf16, clip [-2,2] (LLaMA-7B W4-A8 down_proj shapes, configs[1])",
                   "elements_per_step": elems_step, "buffer_sets": wl.nsets, "parallelism": "replicas" if world > 1 else "1gpu",
                   "semantics": "cpu_eager",
                   "backward": "mask (forward records row bounds + 1-bit STE mask; backward does not re-read x)",
                   "launches_per_step": "2 (weight + input of a QuantizeLinear share one forward and one backward launch)"},
        "hbm_gbs_algorithmic": round(algo_bytes_step / (ms_step * 1e-3) / 1e9 * world, 1),
    }

    if rank == 0 and not args.no_extras:
        traffic = load_traffic()
        it = max(20, min(args.steps, 200))
        nb = wl.n
        ks = {
            "sym_fwd_w4": (lambda s: wl.fwd(s, "w"), nb * FWD_BYTES_PER_ELEM),
            "sym_fwd_a8": (lambda s: wl.fwd(s, "a"), nb * FWD_BYTES_PER_ELEM),
            "ste_bwd_a8": (lambda s: wl.bwd(s, "a"), nb * BWD_BYTES_PER_ELEM),
            "ste_bwd_w4": (lambda s: wl.bwd(s, "w"), nb * BWD_BYTES_PER_ELEM),
        }
        pk = {
            "sym_fwd_pair_w4a8": (lambda s: wl.fwd_pair(s), 2 * nb * FWD_BYTES_PER_ELEM),
            "ste_bwd_pair_w4a8": (lambda s: wl.bwd_pair(s), 2 * nb * BWD_BYTES_PER_ELEM),
        }

        def pair_traffic(k):
            if k in traffic:
                return traffic[k]
            parts = {"sym_fwd_pair_w4a8": ("sym_fwd_w4", "sym_fwd_a8"), "ste_bwd_pair_w4a8": ("ste_bwd_w4", "ste_bwd_a8")}[k]
            return sum(traffic[p] for p in parts) if all(p in traffic for p in parts) else None

        out["kernels_step"] = [roofline_entry(k, b, wl.time_kernel(fn, it), pair_traffic(k)) for k, (fn, b) in pk.items()]
        kernels = [roofline_entry(k, b, wl.time_kernel(fn, it), traffic.get(k)) for k, (fn, b) in ks.items()]
        out["kernels"] = kernels
        # the reference's data flow on the same kernels' siblings: forward without mask/bounds, backward re-reading x
        alt = {
            "sym_fwd_w4_plain": (lambda s: wl.fwd_plain(s, "w"), nb * FWD_BYTES_PER_ELEM),
            "ste_bwd_a8_xread": (lambda s: wl.bwd_xread(s, "a"), nb * BWD_BYTES_PER_ELEM),
        }
        out["kernels_reference_dataflow"] = [roofline_entry(k, b, wl.time_kernel(fn, it), traffic.get(k)) for k, (fn, b) in alt.items()]
        ac = {
            "sym_fwd_w4_autocast_bf16_out": (lambda s: wl.fwd_autocast(s, "w", False), nb * FWD_BYTES_PER_ELEM),
            "sym_fwd_a8_autocast_bf16_out": (lambda s: wl.fwd_autocast(s, "a", False), nb * FWD_BYTES_PER_ELEM),
            "sym_fwd_a8_autocast_fp32_out": (lambda s: wl.fwd_autocast(s, "a", True), nb * 6),  # read 2 + write 4 B/elem
        }
        out["kernels_autocast_arithmetic"] = [roofline_entry(k, b, wl.time_kernel(fn, it), traffic.get(k)) for k, (fn, b) in ac.items()]
        # `roofline`: the forward launch of the step (row_reg_kernel over the W4 weight and the A8 input: reduce -> scale ->
        # round -> dequant).  It moves exactly its algorithmic bytes, so its fraction is a real byte rate; the mask backward
        # moves fewer bytes than its 6 B/elem accounting (see `kernels_step`, frac > 1).
        fwp = out["kernels_step"][0]
        out["roofline"] = {k: fwp[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "us_per_launch") if k in fwp}
        out["roofline"]["kernel"] = "row_reg_kernel (Sym forward of the step: W4 weight + A8 input in one launch)"
        for k in ("traffic_gbs", "traffic_frac"):
            if k in fwp:
                out["roofline"][k] = fwp[k]
        tot_us = sum(e["us_per_launch"] for e in out["kernels_step"])
        out["roofline_step"] = {"bound": "hbm", "achieved": round(algo_bytes_step / (tot_us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": round(algo_bytes_step / (tot_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                "what": "both launches of one step: 10 algorithmic B/elem / sum of launch times"}
        out["unpaired_step"] = {"ms_per_step": round(timed_region(wl.step_unpaired, it, 5, torch.cuda.synchronize) / it * 1e3, 4),
                                "what": "the same step as four single-tensor launches (fq_sym_fwd_train x2, fq_ste_bwd_mask x2)"}
        # a live yardstick for "how fast can this device move the same bytes": ATen's device-to-device copy of the W tensor
        # (read 90.2 MB + write 90.2 MB = one single-tensor forward's algorithmic bytes), timed like the kernels above
        cmean, cpct = wl.time_kernel(lambda s: s["yw"].copy_(s["w"]), it)
        out["copy_reference"] = {"what": "torch Tensor.copy_ device-to-device over the same 180.4 MB as sym_fwd_w4 (compare kernels[0])",
                                 "us_per_launch": round(cmean * 1e3, 2), "gbs": round(nb * FWD_BYTES_PER_ELEM / (cmean * 1e-3) / 1e9, 1),
                                 "us_p10_p50_p90": [round(v * 1e3, 2) for v in cpct]}
        if world == 1:
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
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
