"""GPU tier: the widened rows of SURVEY §8f and the ABI's concurrency / capture claims.
  * 1-/2-bit weight branches (utils_quant.py:202-242) vs golden + oracle + live ATen
  * shared activation quant for sibling projections; weight-quant reuse across checkpoint recompute
  * hipGraph capture of the C-ABI calls; use from several streams and from the autograd thread
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint

from conftest import ROOT, bits_equal, golden, mismatch_report, to_f32
from oracle import oracle as O
from test_gpu_parity import TD, dev_from, np_from

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import llm_qat_amd
    from llm_qat_amd import _lib
    _lib.lib()
    return llm_qat_amd


@pytest.fixture
def no_pairing(pkg):
    """launch-count assertions below are written for the unpaired data flow (weight and input in separate launches)"""
    pkg.pair_operands(False)
    yield
    pkg.pair_operands(True)


# ------------------------------------------------------------------------------------------ W1 / W2
def test_low_bit_weight_golden(pkg):
    G = golden("w12.npz")
    for c in G.cases:
        dt = c["dtype"]
        w = dev_from(G.arr(c, "w"), dt)
        sc = dev_from(G.arr(c, "scale"), dt)
        q = pkg.ops.low_bit_weight(w, sc if not c["layerwise"] else sc.reshape(()), c["w_bits"])
        assert bits_equal(np_from(q), G.arr(c, "wq"), dt), f"{c['name']}: {mismatch_report(np_from(q), G.arr(c, 'wq'), dt)}"


def eager_low_bit(w, w_bits, layerwise):
    """the reference's op chain for the branch (utils_quant.py:203-242), run by ATen on the device"""
    if w_bits == 1:
        sc = torch.mean(abs(w)).detach() if layerwise else torch.mean(abs(w), dim=1, keepdim=True).detach()
        q = sc * torch.sign(w / sc)
    else:
        nb, cv = 2 ** (w_bits - 1), 1 - 1e-2
        sc = 2 * torch.mean(abs(w)).detach() if layerwise else 2 * torch.mean(abs(w), dim=1, keepdim=True).detach()
        q = sc * (torch.round(torch.clamp(w / sc, -cv, cv) * nb - 0.5) + 0.5) / nb
    return q.detach() - w.detach() + w


@pytest.mark.parametrize("dtype", ["bf16", "fp32", "fp16"])
def test_low_bit_module_vs_live_aten(pkg, dtype):
    from llm_qat_amd.utils_quant import QuantizeLinear
    g = torch.Generator(device="cuda").manual_seed(5)
    for (out_f, in_f) in [(64, 256), (33, 100), (128, 11008), (7, 4097)]:
        for w_bits in (1, 2):
            for lw in (False, True):
                lin = QuantizeLinear(in_f, out_f, w_bits=w_bits, a_bits=32, weight_layerwise=lw).cuda().to(TD[dtype])
                with torch.no_grad():
                    lin.weight.copy_(torch.randn(out_f, in_f, generator=g, device="cuda") * 0.05)
                x = torch.randn(3, in_f, generator=g, device="cuda").to(TD[dtype])
                out = lin(x)
                wref = lin.weight.detach().clone().requires_grad_(True)
                ref = F.linear(x, eager_low_bit(wref, w_bits, lw))
                assert torch.equal(out, ref), (dtype, out_f, in_f, w_bits, lw)
                go = torch.randn_like(out)
                out.backward(go)
                ref.backward(go)
                assert torch.equal(lin.weight.grad, wref.grad)       # identity STE for these branches


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp32"])
def test_low_bit_chain_stress_vs_oracle(pkg, dtype):
    """Round 3 rewrote the 1-/2-bit elementwise chain (packed fp32, steps that are provably the identity for an ordinary scale
    removed, odd cases behind wave-uniform branches).  Randomized stress of BOTH kernels that share it against the oracle's op-by-op
    chain (utils_quant.py:203-242), with the scale given (so the summation order plays no part): ordinary, zero, denormal, tiny,
    huge, infinite and NaN scales (the ordinary-scale fast path is guarded by sc in [2^-60, 2^100]); w with +-0, NaN, +-inf,
    denormals, values on and around the clamp bounds, quotients that round to zero; vector-aligned and odd widths."""
    import os
    rng = np.random.default_rng({"bf16": 31, "fp16": 32, "fp32": 33}[dtype])
    tiny = {"bf16": 1e-38, "fp16": 6e-8, "fp32": 1e-42}[dtype]
    odd_scales = [0.0, tiny, 1e-30, 1e-20, 2.0 ** -60, 2.0 ** -61, 2.0 ** 100, 2.0 ** 101, 1e30, float("inf"), float("nan")]
    for trial in range(int(os.environ.get("LLMQAT_STRESS_TRIALS", "120")) // 2):
        rows, cols = int(rng.integers(1, 24)), int(rng.choice([1, 7, 8, 64, 100, 256, 264, 1000, 1024, 4096, 4104, 11008]))
        w_bits = 1 + trial % 2
        row_scale = rng.choice([1e-3, 0.02, 1.0, 40.0], size=(rows, 1)).astype(np.float32)
        w = rng.standard_normal((rows, cols)).astype(np.float32) * row_scale
        sc = (np.abs(w).mean(axis=1) * (2.0 if w_bits == 2 else 1.0)).astype(np.float32)
        for r in range(rows):
            if rng.random() < 0.35:
                with np.errstate(over="ignore"):
                    sc[r] = np.float32(rng.choice(odd_scales))
        sct = torch.from_numpy(sc).to(TD[dtype])                       # the scale is a tensor of the weight's dtype (:205-209)
        scf = sct.float().numpy()
        # adversarial elements: signed zeros, specials, denormals, the clamp bounds, quotients far below 1
        for _ in range(min(12, cols)):
            r, c = int(rng.integers(0, rows)), int(rng.integers(0, cols))
            base = scf[r] if np.isfinite(scf[r]) and scf[r] != 0 else 1.0
            with np.errstate(over="ignore", invalid="ignore"):
                w[r, c] = np.float32(rng.choice([0.0, -0.0, np.nan, np.inf, -np.inf, tiny, -tiny, 0.99 * base, -0.99 * base, 0.9921875 * base, 0.98828125 * base,
                                                 0.25 * base, 0.75 * base, -0.25 * base, base * 2.0 ** -120, -base * 2.0 ** -130, base * 1e-30, 3.0 * base]))
        wt = torch.from_numpy(w).to(TD[dtype]).cuda()
        w_np = np_from(wt)
        want, _ = O.w12_fwd(w_np, rows, cols, w_bits, dtype, scale_in=scf)
        got = pkg.ops.low_bit_weight(wt, sct.cuda(), w_bits)
        tag = f"trial {trial}: {dtype} [{rows},{cols}] w_bits={w_bits}"
        assert bits_equal(np_from(got), want, dtype), f"{tag} fq_w12_fwd: {mismatch_report(np_from(got), want, dtype)}"
        res = pkg.ops.low_bit_weight_fused(wt, w_bits)
        if res is not None:                                             # the one-launch kernel: ITS scale, the same chain
            q, s_used = res
            want2, _ = O.w12_fwd(w_np, rows, cols, w_bits, dtype, scale_in=s_used.float().cpu().numpy())
            assert bits_equal(np_from(q), want2, dtype), f"{tag} fq_w12_fwd_rows: {mismatch_report(np_from(q), want2, dtype)}"


def test_low_bit_fused_row_mean(pkg):
    """the 1-/2-bit branch in ONE launch with the row mean reduced in-kernel IN ATen'S OWN SUMMATION ORDER (restated
    from torch's Reduce.cuh, fq_kernels.h w12_row_aten_kernel) -- the default since round 4.  Bar: ZERO differing rows against live
    ATen `torch.mean(abs(w), dim=1)` (utils_quant.py:205-209 / :219-224), scale and output, bf16 / fp16 / fp32, at the model shapes
    [4096,11008], [11008,4096], [4096,4096] (+ 13B widths) and at shapes around every boundary of ATen's reduce configuration: the
    own-row / shared-row switch at cols 8128 | 8132, partial last groups per thread, rows not a multiple of the 8 per block."""
    g = torch.Generator(device="cuda").manual_seed(8)
    shapes = [(4096, 11008), (11008, 4096), (4096, 4096), (5120, 13824), (13824, 5120), (5120, 5120),
              (8, 256), (9, 260), (17, 1000), (33, 8128), (33, 8132), (16, 8192), (11, 12288), (8, 16384), (9, 32768), (4097, 2052)]
    for dtype in ("bf16", "fp16", "fp32"):
        for rows, cols in shapes:
            scale = 0.02 if rows * cols > 1 << 20 else 0.5
            w = (torch.randn(rows, cols, generator=g, device="cuda") * scale).to(TD[dtype])
            if rows >= 9:
                w[1] = 0.0                      # an all-zero row: scale 0 -> 0/0
                w[2, 3] = float("nan")
                w[3] *= 1e-6
            for w_bits in (1, 2):
                res = pkg.ops.low_bit_weight_fused(w, w_bits)
                if dtype == "fp32" and 4096 < cols <= 8128:
                    assert res is None, (dtype, rows, cols)     # not held in registers: the caller takes ATen's reduction
                    continue
                assert res is not None, (dtype, rows, cols)
                q, sc = res
                ref_sc = torch.mean(abs(w), dim=1) * (1 if w_bits == 1 else 2)
                bad = int((~((sc == ref_sc) | (sc.isnan() & ref_sc.isnan()))).sum())
                assert bad == 0, f"{dtype} [{rows},{cols}] w{w_bits}: {bad} of {rows} row scales differ from ATen's"
                assert bits_equal(np_from(q), np_from(eager_low_bit(w, w_bits, False)), dtype), f"{dtype} [{rows},{cols}] w{w_bits}: output"
    # shapes the kernel does not serve answer None (the module then takes ATen's abs + mean and fq_w12_fwd)
    for shape in ((4, 512), (64, 200), (64, 258), (8, 65536)):
        assert pkg.ops.low_bit_weight_fused(torch.randn(shape, device="cuda").bfloat16(), 1) is None, shape
    # module level: the one launch is the default, the three launches the switchable alternative -- identical bits, identity gradient
    from llm_qat_amd.utils_quant import QuantizeLinear
    lin = QuantizeLinear(1024, 64, w_bits=2, a_bits=32).cuda().bfloat16()
    x = torch.randn(4, 1024, device="cuda").bfloat16()
    with Counter(pkg.ops, ["low_bit_weight_fused", "low_bit_weight"]) as c:
        b = lin(x)
    assert c.n == 1
    b.float().sum().backward()
    assert lin.weight.grad is not None
    pkg.fuse_low_bit_mean(False)
    try:
        a = lin(x)
    finally:
        pkg.fuse_low_bit_mean(True)
    assert torch.equal(a, b)


class Counter:
    """counts calls of the named `ops` functions, and the launches the C++ autograd nodes make of the same entry points"""
    TWINS = {}

    # the C++ node's own launches of the same entry points (a sibling's weight-only launch is one of the three single-tensor forwards in Python)
    CPP = {"pair_forward": "cpp_pair_forward", "pair_backward": "cpp_pair_backward", "train_forward": "cpp_weight_forward",
           "sym_forward_autocast": "cpp_weight_forward", "sym_quantize": "cpp_weight_forward", "train_backward": "cpp_one_backward",
           "train_backward_wide": "cpp_one_backward_wide"}

    def __init__(self, mod, names):
        self.mod, self.n = mod, 0
        self.names = list(names) + [self.TWINS[n] for n in names if n in self.TWINS and hasattr(mod, self.TWINS[n])]
        self.cpp = sorted({self.CPP[n] for n in names if n in self.CPP})

    def _cpp_counts(self):
        from llm_qat_amd import utils_quant as U
        c = U._cnode.counters() if U._cnode is not None else {}
        return sum(c.get(k, 0) for k in self.cpp)

    def __enter__(self):
        self.orig = {n: getattr(self.mod, n) for n in self.names}
        for n, f in self.orig.items():
            setattr(self.mod, n, self._wrap(f))
        self.cpp0 = self._cpp_counts()
        return self

    def _wrap(self, f):
        def g(*a, **k):
            self.n += 1
            return f(*a, **k)
        return g

    def __exit__(self, *exc):
        for n, f in self.orig.items():
            setattr(self.mod, n, f)
        self.n += self._cpp_counts() - self.cpp0


def qkv_loss(mods, x):
    outs = [m(x) for m in mods]
    return sum((o.float() * (i + 1)).square().mean() for i, o in enumerate(outs))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_shared_activation_quant_is_transparent(pkg, dtype, no_pairing):
    from llm_qat_amd.utils_quant import QuantizeLinear
    torch.manual_seed(0)
    mods = [QuantizeLinear(512, 384, w_bits=4, a_bits=8).cuda().to(dtype) for _ in range(3)]
    xs = torch.randn(2, 64, 512, device="cuda").to(dtype) * 1.5
    res = {}
    for share in (True, False):
        pkg.share_activation_quant(share)
        try:
            x = xs.clone().requires_grad_(True)
            for m in mods:
                m.zero_grad(set_to_none=True)
            with Counter(pkg.ops, ["train_forward", "sym_quantize"]) as c:
                loss = qkv_loss(mods, x)
            loss.backward()
            res[share] = (loss.detach().clone(), x.grad.clone(), [m.weight.grad.clone() for m in mods], c.n)
        finally:
            pkg.share_activation_quant(True)
    assert torch.equal(res[True][0], res[False][0])
    assert torch.equal(res[True][1], res[False][1])                  # grad wrt the shared input: bit-identical
    for a, b in zip(res[True][2], res[False][2]):
        assert torch.equal(a, b)
    assert res[False][3] == 6 and res[True][3] == 4                  # 3 weights + 3 activations -> 3 weights + 1 activation


def test_shared_activation_respects_inplace_and_identity(pkg, no_pairing):
    from llm_qat_amd.utils_quant import QuantizeLinear
    m1, m2 = (QuantizeLinear(256, 64, w_bits=32, a_bits=8).cuda() for _ in range(2))
    x = torch.randn(4, 256, device="cuda")
    with torch.no_grad(), Counter(pkg.ops, ["train_forward", "sym_quantize"]) as c:
        m1(x)
        m2(x)                      # same tensor, same version -> shared
        assert c.n == 1
        x.mul_(2.0)                # version bump -> must recompute
        m1(x)
        assert c.n == 2
        m2(x.clone())              # different tensor object -> must recompute
        assert c.n == 3


def test_shared_activation_not_reused_across_training_steps(pkg):
    """the same input OBJECT fed again after a backward must get a fresh autograd node (bounds/plain modes free
    their saved tensors in backward; a stale shared node would raise or, worse, go unnoticed)"""
    from llm_qat_amd.utils_quant import QuantizeLinear
    for mode in ("mask", "bounds", "plain"):
        pkg.set_backward_mode(mode)
        try:
            m1, m2 = (QuantizeLinear(256, 64, w_bits=4, a_bits=8).cuda().bfloat16() for _ in range(2))
            x = (torch.randn(8, 256, device="cuda") * 1.5).bfloat16().requires_grad_(True)
            grads = []
            for step in range(3):
                x.grad = None
                (m1(x).float().square().mean() + m2(x).float().square().mean()).backward()
                grads.append(x.grad.clone())
            assert torch.equal(grads[0], grads[1]) and torch.equal(grads[1], grads[2]), mode
        finally:
            pkg.set_backward_mode("mask")


@pytest.mark.parametrize("autocast", [False, True])
def test_weight_quant_cache_with_checkpoint(pkg, autocast, no_pairing):
    from llm_qat_amd.utils_quant import QuantizeLinear
    torch.manual_seed(1)
    ctx = torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast)
    ctx.__enter__()
    try:
        _weight_quant_cache_body(pkg)
    finally:
        ctx.__exit__(None, None, None)


def _weight_quant_cache_body(pkg):
    from llm_qat_amd.utils_quant import QuantizeLinear
    net = torch.nn.Sequential(QuantizeLinear(256, 512, w_bits=4, a_bits=8), torch.nn.SiLU(), QuantizeLinear(512, 256, w_bits=4, a_bits=8)).cuda().bfloat16()
    with torch.no_grad():
        net[0].weight[3, 5] = 2.5   # beyond the STE clip
    xs = torch.randn(8, 256, device="cuda", dtype=torch.bfloat16)
    res = {}
    for cache in (False, True):
        pkg.enable_weight_quant_cache(cache)
        try:
            net.zero_grad(set_to_none=True)
            x = xs.clone().requires_grad_(True)
            with Counter(pkg.ops, ["train_forward", "sym_quantize", "sym_forward_autocast"]) as c:
                out = checkpoint(net, x, use_reentrant=False)
                out.float().square().mean().backward()
            res[cache] = (out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in net.parameters()], c.n)
        finally:
            pkg.enable_weight_quant_cache(False)
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    for a, b in zip(res[True][2], res[False][2]):
        assert torch.equal(a, b)
    assert res[True][2][0][3, 5] == 0                                 # STE mask still applied through the reuse node
    assert res[False][3] == 8 and res[True][3] == 6                   # 2 x (2 weights + 2 acts) -> weights once
    # persistent mode: gradient-accumulation micro-batches reuse the weight until it changes
    pkg.enable_weight_quant_cache(True, persistent=True)
    try:
        net.zero_grad(set_to_none=True)
        with Counter(pkg.ops, ["train_forward", "sym_quantize", "sym_forward_autocast"]) as c:
            for micro in range(3):
                checkpoint(net, xs.clone().requires_grad_(True), use_reentrant=False).float().square().mean().backward()
        assert c.n == 2 + 3 * 2 * 2                                     # 2 weights once + activations every pass
        g_persist = [p.grad.clone() for p in net.parameters()]
        pkg.enable_weight_quant_cache(False)
        net.zero_grad(set_to_none=True)
        for micro in range(3):
            checkpoint(net, xs.clone().requires_grad_(True), use_reentrant=False).float().square().mean().backward()
        for a, b in zip(g_persist, [p.grad for p in net.parameters()]):
            assert torch.equal(a, b)
    finally:
        pkg.enable_weight_quant_cache(False)
    # an optimizer step changes the weights: the cache must not serve a stale copy
    pkg.enable_weight_quant_cache(True)
    try:
        with torch.no_grad():
            y1 = net(xs)
            net(xs)
            for p in net.parameters():
                p.add_(0.01)
            y2 = net(xs)
        pkg.enable_weight_quant_cache(False)
        with torch.no_grad():
            y3 = net(xs)
        assert torch.equal(y2, y3) and not torch.equal(y1, y2)
    finally:
        pkg.enable_weight_quant_cache(False)


@pytest.mark.parametrize("reentrant", [False, True])
@pytest.mark.parametrize("autocast", [False, True])
def test_weight_quant_cache_keeps_operand_pairing(pkg, autocast, reentrant):
    """the weight cache used to switch operand pairing off.  Now the first use of a weight in a step still shares
    its launch with the input (and fills the cache from it); the recompute launches nothing for the weight.  Same results."""
    from llm_qat_amd.utils_quant import QuantizeLinear
    torch.manual_seed(1)
    net = torch.nn.Sequential(QuantizeLinear(256, 512, w_bits=4, a_bits=8), torch.nn.SiLU(), QuantizeLinear(512, 256, w_bits=4, a_bits=8)).cuda().bfloat16()
    with torch.no_grad():
        net[0].weight[3, 5] = 2.5
    xs = torch.randn(8, 256, device="cuda", dtype=torch.bfloat16)
    res = {}
    names = ["pair_forward", "train_forward", "sym_quantize", "sym_forward_autocast", "quantize_train"]
    for cache in (False, True):
        pkg.enable_weight_quant_cache(cache)
        try:
            net.zero_grad(set_to_none=True)
            x = xs.clone().requires_grad_(True)
            with Counter(pkg.ops, names) as c, torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                out = checkpoint(net, x, use_reentrant=reentrant)
                out.float().square().mean().backward()
            res[cache] = (out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in net.parameters()], c.n)
        finally:
            pkg.enable_weight_quant_cache(False)
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    for a, b in zip(res[True][2], res[False][2]):
        assert torch.equal(a, b)
    assert res[True][2][0][3, 5] == 0
    # without the cache: 2 pair launches per pass x 2 passes = 4; with it: 2 pair launches in the first pass, then only the 2 activations
    assert res[False][3] == 4 and res[True][3] == 4, (res[False][3], res[True][3])


# ------------------------------------------------------------------------------------------ ABI claims
def test_c_abi_calls_are_graph_capturable(pkg):
    """no allocation / sync inside the library: a forward+backward pair captures into a hipGraph and replays"""
    from llm_qat_amd import _lib
    L = _lib.lib()
    rows, cols = 512, 4096
    x = torch.randn(rows, cols, device="cuda").bfloat16()
    g = torch.randn(rows, cols, device="cuda").bfloat16()
    y, gx = torch.empty_like(x), torch.empty_like(x)
    bounds = torch.empty(rows, 2, device="cuda")
    mbytes = L.fq_ste_mask_bytes(rows, cols, _lib.DTYPE_BF16)
    mask = torch.empty(mbytes, dtype=torch.uint8, device="cuda")
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(graph, stream=s):
            st = torch.cuda.current_stream().cuda_stream
            assert L.fq_sym_fwd_train(x.data_ptr(), y.data_ptr(), rows, cols, 8, _lib.DTYPE_BF16, 0, -2.0, 2.0, bounds.data_ptr(),
                                      mask.data_ptr(), mbytes, st) == 0
            assert L.fq_ste_bwd_mask(g.data_ptr(), gx.data_ptr(), rows, cols, -2.0, 2.0, bounds.data_ptr(), mask.data_ptr(), mbytes,
                                     _lib.DTYPE_BF16, st) == 0
    for trial in range(3):
        x.copy_(torch.randn(rows, cols, device="cuda") * (trial + 1))
        graph.replay()
        torch.cuda.synchronize()
        yo, _, _ = O.sym_fwd(np_from(x), rows, cols, 8, "bf16")
        assert bits_equal(np_from(y), yo, "bf16")
        assert bits_equal(np_from(gx), O.ste_bwd(np_from(g), np_from(x), -2.0, 2.0, "bf16"), "bf16")


def test_concurrent_streams_and_threads(pkg):
    """stateless + re-entrant: 4 host threads x own stream, interleaved launches, all results exact"""
    import threading
    from llm_qat_amd.utils_quant import SymQuantizer
    errors = []

    def worker(seed):
        try:
            torch.cuda.set_device(0)
            s = torch.cuda.Stream()
            gen = torch.Generator(device="cuda").manual_seed(seed)
            with torch.cuda.stream(s):
                for it in range(20):
                    x = (torch.randn(256, 1024 + 8 * seed, generator=gen, device="cuda") * (0.02 if it % 2 else 1.5)).bfloat16().requires_grad_(True)
                    y = SymQuantizer.apply(x, torch.tensor([-2.0, 2.0]), 4 + 4 * (it % 2), False)
                    y.backward(torch.ones_like(y))
                    s.synchronize()
                    yo, _, _ = O.sym_fwd(np_from(x), 256, x.shape[1], 4 + 4 * (it % 2), "bf16")
                    if not bits_equal(np_from(y), yo, "bf16"):
                        errors.append((seed, it, "fwd"))
                    want = O.ste_bwd(np_from(torch.ones_like(y)), np_from(x), -2.0, 2.0, "bf16")
                    if not bits_equal(np_from(x.grad), want, "bf16"):
                        errors.append((seed, it, "bwd"))
        except Exception as e:  # noqa: BLE001
            errors.append((seed, repr(e)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:5]


def test_python_level_graph_capture(pkg):
    """the autograd Functions allocate only through PyTorch's (graph-safe) allocator and never synchronise:
    a forward+backward captures into a CUDA/HIP graph and replays with fresh data"""
    from llm_qat_amd.utils_quant import SymQuantizer
    clip = torch.tensor([-2.0, 2.0])
    x = (torch.randn(512, 4096, device="cuda") * 1.5).bfloat16().requires_grad_(True)
    g = torch.randn(512, 4096, device="cuda").bfloat16()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):   # warm the allocator / lazy inits outside the capture
            SymQuantizer.apply(x, clip, 8, False).backward(g)
        x.grad = None
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y = SymQuantizer.apply(x, clip, 8, False)
        y.backward(g)
    for trial in range(2):
        with torch.no_grad():
            x.copy_(torch.randn(512, 4096, device="cuda") * (1.0 + trial))
        graph.replay()
        torch.cuda.synchronize()
        yo, _, _ = O.sym_fwd(np_from(x), 512, 4096, 8, "bf16")
        assert bits_equal(np_from(y), yo, "bf16")
        assert bits_equal(np_from(x.grad), O.ste_bwd(np_from(g), np_from(x), -2.0, 2.0, "bf16"), "bf16")


def test_plain_c_host_links_and_matches_oracle(tmp_path):
    """the boundary is a real C ABI: a C program (no Python, no torch) built with gcc against include/*.h and the .so
    drives both data flows on the GPU and matches the oracle bit for bit"""
    import os
    import shutil
    import subprocess
    from conftest import ROOT
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    exe = str(tmp_path / "abi_smoke")
    cmd = ["gcc", "-std=c11", "-O2", "-fno-fast-math", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(rocm, "include"), os.path.join(ROOT, "tests", "c_host", "abi_smoke.c"), os.path.join(ROOT, "oracle", "fq_oracle.c"),
           "-L", os.path.join(ROOT, "llm-qat_amd"), "-lllmqat_fakequant", "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-lm",
           f"-Wl,-rpath,{os.path.join(ROOT, 'llm-qat_amd')}", f"-Wl,-rpath,{os.path.join(rocm, 'lib')}", "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "c host ok" in r.stdout


def test_mask_mode_backward_needs_no_input_storage(pkg):
    """FSDP full_shard frees the unsharded flat parameter between forward and backward.  In the default (mask) mode the
    backward never touches the forward's input, so it works even after that storage is gone."""
    from llm_qat_amd.utils_quant import SymQuantizer
    assert pkg.get_backward_mode() == "mask"
    flat = (torch.randn(2 * 512 * 1024, device="cuda") * 1.2).bfloat16()
    w = flat[512 * 1024:].view(512, 1024).requires_grad_(True)          # a view into a flat buffer, as FSDP hands out
    g = torch.randn(512, 1024, device="cuda").bfloat16()
    ref_mask = ((w.detach() >= 2) | (w.detach() <= -2)).clone()
    y = SymQuantizer.apply(w, torch.tensor([-2.0, 2.0]), 4, False)
    flat.untyped_storage().resize_(0)                                   # "reshard": the weight's memory is released
    (gw,) = torch.autograd.grad(y, w, g)
    assert torch.equal(gw, torch.where(ref_mask, torch.zeros_like(g), g))


@pytest.mark.parametrize("autocast", [False, True])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_paired_operand_launch_is_transparent(pkg, dtype, autocast):
    """QuantizeLinear quantizes weight and input in ONE launch (and their gradients in one): outputs and gradients are
    bit-identical to the two-launch flow; q/k/v siblings still share the activation (the first pairs, the others do only
    their weight)."""
    from llm_qat_amd.utils_quant import QuantizeLinear
    torch.manual_seed(0)
    mods = [QuantizeLinear(1024, 768, w_bits=4, a_bits=8).cuda().to(dtype) for _ in range(3)]
    with torch.no_grad():
        mods[0].weight[3, 5] = 2.5
    xs = (torch.randn(2, 64, 1024, device="cuda") * 1.5).to(dtype)
    res = {}
    for pair in (True, False):
        pkg.pair_operands(pair)
        try:
            x = xs.clone().requires_grad_(True)
            for m in mods:
                m.zero_grad(set_to_none=True)
            with Counter(pkg.ops, ["pair_forward", "pair_backward", "train_forward", "train_backward", "sym_forward_autocast"]) as c:
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                    loss = qkv_loss(mods, x)
                loss.backward()
            res[pair] = (loss.detach().clone(), x.grad.clone(), [m.weight.grad.clone() for m in mods], c.n)
        finally:
            pkg.pair_operands(True)
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    for a, b in zip(res[True][2], res[False][2]):
        assert torch.equal(a, b)
    assert res[True][2][0][3, 5] == 0
    # unpaired: 3 weights + 1 shared activation forward, 3 weight + 3 activation backwards (round 5: every sibling masks its own input
    # gradient, as in the reference's graph) = 10 calls; paired: (1 pair + 2 weights) forward + 3 two-tensor backwards = 6
    assert res[False][3] == 10 and res[True][3] == 6, (res[False][3], res[True][3])
    # eval / no-grad
    with torch.no_grad():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            a = mods[0](xs)
            pkg.pair_operands(False)
            b = mods[0](xs)
            pkg.pair_operands(True)
    assert torch.equal(a, b)


def test_a_bias_assigned_afterwards_is_added_on_every_path(pkg):
    """The constructor never creates a bias (reference :176), but forward adds one if the attribute has been set since (:251-252):
    the paired launch, the two-call flow and conservative mode all do."""
    from llm_qat_amd.utils_quant import QuantizeLinear
    torch.manual_seed(3)
    lin = QuantizeLinear(512, 256, w_bits=4, a_bits=8).cuda().bfloat16()
    x = torch.randn(4, 32, 512, device="cuda").bfloat16()
    with torch.no_grad():
        plain = lin(x)
        lin.bias = torch.nn.Parameter(torch.randn(256, device="cuda").bfloat16())
        want = plain + lin.bias.view(1, -1).expand_as(plain)
        outs = [lin(x)]
        pkg.pair_operands(False)
        try:
            outs.append(lin(x))
        finally:
            pkg.pair_operands(True)
    assert not torch.equal(want, plain) and all(torch.equal(o, want) for o in outs)


@pytest.mark.parametrize("autocast", [False, True])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_kv_hooks_in_one_launch(pkg, dtype, autocast):
    """quantize_kv(K, V, ...) == the two SymQuantizer.apply calls of modeling_llama_quant.py:320-327, bit for bit in
    values and gradients (fp32 results under autocast, like the reference), with ONE forward launch instead of two (the backward keeps one
    launch per tensor: K and V never share an autograd node -- utils_quant.quantize_kv says why)."""
    from llm_qat_amd.utils_quant import SymQuantizer, quantize_kv
    torch.manual_seed(1)
    clip = torch.tensor([-2.0, 2.0])
    k0 = (torch.randn(2, 96, 1024, device="cuda") * 1.2).to(dtype)
    v0 = (torch.randn(2, 96, 1024, device="cuda") * 0.8).to(dtype)
    v0[1, 3, 7] = 2.0
    gk, gv = torch.randn(2, 96, 1024, device="cuda"), torch.randn(2, 96, 1024, device="cuda")
    out = {}
    for one in (False, True):
        k, v = k0.clone().requires_grad_(True), v0.clone().requires_grad_(True)
        with Counter(pkg.ops, ["pair_forward", "pair_backward", "pair_backward_wide", "train_forward", "train_backward", "train_backward_wide",
                               "sym_forward_autocast"]) as c:
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                if one:
                    kq, vq = quantize_kv(k, v, clip, clip, 4)
                else:
                    kq, vq = SymQuantizer.apply(k, clip, 4, False), SymQuantizer.apply(v, clip, 4, False)
            wide = autocast and dtype == torch.bfloat16
            assert kq.dtype == vq.dtype == (torch.float32 if wide else dtype)
            torch.autograd.backward([kq, vq], [gk.to(kq.dtype), gv.to(vq.dtype)])
        out[one] = (kq.detach(), vq.detach(), k.grad, v.grad, c.n)
    for a, b in zip(out[True][:4], out[False][:4]):
        assert a.dtype == b.dtype and torch.equal(a, b)
    assert out[False][3][1, 3, 7] == 0
    assert out[False][4] == 4 and out[True][4] == 3, (out[False][4], out[True][4])
    # only V needs a gradient; no-grad; different clips fall back to two calls
    k, v = k0.clone(), v0.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        kq, vq = quantize_kv(k, v, clip, clip, 4)
    vq.backward(gv.to(vq.dtype))
    assert torch.equal(v.grad, out[False][3]) and torch.equal(kq, out[False][0])
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        kq, vq = quantize_kv(k0, v0, clip, clip, 4)
        k2, v2 = quantize_kv(k0, v0, clip, torch.tensor([-1.0, 1.0]), 4)
    assert torch.equal(kq, out[False][0]) and torch.equal(vq, out[False][1]) and torch.equal(k2, kq) and torch.equal(v2, vq)


@pytest.mark.parametrize("autocast", [False, True])
def test_unchanged_kv_call_site_is_one_launch(pkg, autocast):
    """VERDICT r02 "missing" item 2: the KV-cache hooks exactly as the reference writes them (modeling_llama_quant.py:317-327 --
    k_proj, v_proj, then two consecutive SymQuantizer.apply calls) cost ONE launch forward (backward: one per tensor, as in the reference -- the
    speculated V must stay outside the graph until it is asked for), transparently:
    values and every gradient bit-identical to pairing off; a changed tensor, other clips / bits or a single hook fall back."""
    from llm_qat_amd.utils_quant import QuantizeLinear, SymQuantizer
    torch.manual_seed(3)
    clip = torch.tensor([-2.0, 2.0])
    kp, vp = (QuantizeLinear(512, 512, w_bits=4, a_bits=8).cuda().bfloat16() for _ in range(2))
    h0 = (torch.randn(2, 48, 512, device="cuda") * 1.1).bfloat16()
    names = ["pair_forward", "pair_backward", "pair_backward_wide", "train_forward", "train_backward", "train_backward_wide", "sym_forward_autocast"]

    def site(h, variant="reference"):
        k = kp(h)
        v = vp(h)
        if variant == "reference":
            k = SymQuantizer.apply(k, clip, 4, False)
            v = SymQuantizer.apply(v, clip, 4, False)
        elif variant == "v_modified":        # V changes between the two hooks: the speculative result must not be served
            k = SymQuantizer.apply(k, clip, 4, False)
            v = v * 2.0
            v = SymQuantizer.apply(v, clip, 4, False)
        elif variant == "other_bits":
            k = SymQuantizer.apply(k, clip, 4, False)
            v = SymQuantizer.apply(v, clip, 8, False)
        elif variant == "k_only":
            k = SymQuantizer.apply(k, clip, 4, False)
        return k, v

    def run(pairing, variant="reference"):
        pkg.pair_kv_hooks(pairing)
        pkg.pair_operands(True)
        pkg.reset_learned_state()   # (a wrong guess switches the speculation off for that call signature: see the next test)
        try:
            for m in (kp, vp):
                m.weight.grad = None
            h = h0.clone().requires_grad_(True)
            with Counter(pkg.ops, names) as c:
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                    k, v = site(h, variant)
                (k.float().square().mean() + 2 * v.float().square().mean()).backward()
            return k.detach(), v.detach(), h.grad.clone(), kp.weight.grad.clone(), vp.weight.grad.clone(), c.n
        finally:
            pkg.pair_kv_hooks(True)

    for variant in ("reference", "v_modified", "other_bits", "k_only"):
        a, b = run(True, variant), run(False, variant)
        for x, y in zip(a[:5], b[:5]):
            assert x.dtype == y.dtype and torch.equal(x, y), variant
        # launches: each QuantizeLinear = 1 pair forward + 1 pair backward (4 in all); the hooks 2 + 2 unpaired, 1 + 2 with the speculation
        # (one forward launch for both; K and V keep their own autograd nodes, so that a V nobody asks for leaves no trace in the graph)
        if variant == "reference":
            assert b[5] == 8 and a[5] == 7, (a[5], b[5])
    wide = autocast
    assert run(True)[0].dtype == (torch.float32 if wide else torch.bfloat16)    # fp32 under autocast, as the reference returns
    # no-grad (eval / the first pass of a reentrant checkpoint): still one launch, same values
    counts = {}
    for pairing in (True, False):
        pkg.pair_kv_hooks(pairing)
        pkg.reset_learned_state()
        try:
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                with Counter(pkg.ops, names + ["sym_quantize"]) as c:
                    k, v = site(h0)
        finally:
            pkg.pair_kv_hooks(True)
        counts[pairing] = c.n
        assert torch.equal(k, run(False)[0]) and torch.equal(v, run(False)[1])
    assert counts[True] == counts[False] - 1, counts


def test_kv_speculation_stops_after_a_wrong_guess_and_says_so(pkg):
    """a call site that quantizes K only (or V with other bits) must not pay for a speculative V forever -- after ONE
    discarded result that call signature stops pairing, the pending result (graph + side buffers) is dropped when the next backward
    starts, and llm_qat_amd.stats() shows all of it."""
    from llm_qat_amd import utils_quant as U
    from llm_qat_amd.utils_quant import QuantizeLinear, SymQuantizer
    torch.manual_seed(4)
    clip = torch.tensor([-2.0, 2.0])
    kp, vp = (QuantizeLinear(256, 256, w_bits=4, a_bits=8).cuda().bfloat16() for _ in range(2))
    h = (torch.randn(2, 16, 256, device="cuda") * 1.1).bfloat16().requires_grad_(True)
    pkg.reset_learned_state()
    pkg.stats(reset=True)

    def k_only_step():
        k, v = kp(h), vp(h)
        k = SymQuantizer.apply(k, clip, 4, False)       # the V hook never comes
        (k.float().square().mean() + v.float().square().mean()).backward()

    with Counter(pkg.ops, ["pair_forward"]) as c:
        k_only_step()
    st = pkg.stats()
    assert st.get("kv_pair_launch") == 1 and st.get("kv_pair_discarded") == 1 and st.get("kv_pair_learned_off") == 1 and not st.get("kv_pair_hit"), st
    assert U._state().kv is None, "the unused V result stayed pinned after the backward"
    n_first = c.n
    with Counter(pkg.ops, ["pair_forward"]) as c:
        k_only_step()
        k_only_step()
    assert c.n == 2 * (n_first - 1), "the site kept speculating after a wrong guess"
    assert pkg.stats().get("kv_pair_launch") == 1
    # the ordinary two-hook site with ANOTHER signature still pairs; after reset_learned_state() so does this one
    pkg.reset_learned_state()
    k, v = kp(h), vp(h)
    k, v = SymQuantizer.apply(k, clip, 4, False), SymQuantizer.apply(v, clip, 4, False)
    st = pkg.stats()
    assert st.get("kv_pair_launch") == 2 and st.get("kv_pair_hit") == 1, st
    (k.float().mean() + v.float().mean()).backward()


def test_forward_under_inference_mode(pkg):
    """inference tensors have no version counter (`t._version` raises); the bookkeeping of activation sharing,
    operand pairing and the K/V hooks must step aside -- the reference works under torch.inference_mode(), so does the drop-in,
    with the values of the no_grad forward."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import tiny_llama as TL
    import llm_qat_amd.utils_quant as UQ
    ids = TL.deterministic_batch().cuda()
    for dt, kv in ((torch.bfloat16, 4), (torch.float32, 8)):
        m = TL.load_deterministic(TL.TinyLlama(UQ, w_bits=4, a_bits=8, kv_bits=kv).to(dt)).cuda()
        with torch.no_grad():
            want = m(ids)[1]
        with torch.inference_mode():
            got = m(ids)[1]
            with torch.autocast("cuda", dtype=torch.bfloat16):
                got_ac = m(ids)[1]
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            want_ac = m(ids)[1]
        assert torch.equal(got, want) and torch.equal(got_ac, want_ac), dt
        # a model BUILT under inference mode (its parameters are inference tensors) too
        with torch.inference_mode():
            mi = TL.load_deterministic(TL.TinyLlama(UQ, w_bits=4, a_bits=8, kv_bits=kv).to(dt)).cuda()
            assert torch.equal(mi(ids)[1], want), dt


def _one_rank_group():
    import socket

    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    try:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
        dist.all_reduce(torch.zeros(1, device="cuda"))
    except Exception as e:  # noqa: BLE001 -- an RCCL that cannot start here says nothing about the quantizers
        if dist.is_initialized():
            dist.destroy_process_group()
        pytest.skip(f"RCCL one-rank process group unavailable: {e!r}")
    return dist


@pytest.mark.parametrize("wcache", [False, True])
def test_fsdp_wrapped_model_step_is_unchanged(pkg, wcache):
    """run_train.sh trains with `--fsdp "full_shard auto_wrap"` around each decoder layer (+ activation checkpointing):
    every weight the quantizers see is a view into FSDP's flat parameter, re-pointed before each forward/backward.
    One rank over RCCL: loss and gradients equal the bare model's, with and without the weight-quant cache."""
    import functools

    from torch.distributed.fsdp import FullyShardedDataParallel as FSDP
    from torch.distributed.fsdp.wrap import transformer_auto_wrap_policy
    import tiny_llama as TL
    import llm_qat_amd.utils_quant as UQ
    dist = _one_rank_group()
    try:
        ids = TL.deterministic_batch().cuda()

        def step(model):
            for _ in range(2):
                model.zero_grad(set_to_none=True)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    loss, logits = model(ids, labels=ids)
                loss.backward()
            return loss.detach(), logits.detach()

        bare = TL.load_deterministic(TL.TinyLlama(UQ, w_bits=4, a_bits=8, kv_bits=4).bfloat16()).cuda()
        inner = TL.load_deterministic(TL.TinyLlama(UQ, w_bits=4, a_bits=8, kv_bits=4).bfloat16()).cuda()
        wrapped = FSDP(inner, auto_wrap_policy=functools.partial(transformer_auto_wrap_policy, transformer_layer_cls={TL.Layer}),
                       device_id=0, use_orig_params=True)
        pkg.enable_weight_quant_cache(wcache)
        try:
            l0, g0 = step(bare)
            l1, g1 = step(wrapped)
        finally:
            pkg.enable_weight_quant_cache(False)
        assert torch.equal(l0, l1) and torch.equal(g0, g1)
        with FSDP.summon_full_params(wrapped, with_grads=True):
            got = {n.replace("_fsdp_wrapped_module.", ""): p.grad.clone() for n, p in wrapped.named_parameters() if p.grad is not None}
        want = dict((n, p.grad) for n, p in bare.named_parameters())
        assert set(got) == set(want), sorted(set(want) ^ set(got))[:5]
        for n in want:
            assert torch.equal(got[n].view_as(want[n]), want[n]), n
    finally:
        dist.destroy_process_group()


def test_two_rank_fsdp_full_shard_with_checkpointing(pkg, tmp_path):
    """the reference's real configuration -- FSDP `full_shard auto_wrap` per decoder layer + NON-reentrant gradient
    checkpointing + bf16 autocast (run_train.sh:17-18,:36,:42-43) -- on TWO ranks (fresh child processes, both on this GPU, gloo
    collectives): flat parameters are really sharded, all-gathered before every layer forward / recompute / backward and freed after.
    Three optimizer steps: every step's loss and every parameter afterwards are bit-identical between the reference's eager op chain
    and the drop-in (default settings, the weight cache, and the opt-in sibling groups, whose activation is version-gated)."""
    import socket
    import subprocess
    import sys
    res = {}
    for impl in ("eager", "ours", "ours_wcache"):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        out = str(tmp_path / f"{impl}.pt")
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "fsdp_worker.py"), impl, out], env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.PIPE, text=True))
        logs = [p.communicate(timeout=420) for p in procs]
        assert all(p.returncode == 0 for p in procs), (impl, [lg[1][-1500:] for lg in logs])
        res[impl] = torch.load(out, weights_only=True)
    ref = res["eager"]
    assert torch.isfinite(ref["losses"]).all() and ref["losses"][0] != ref["losses"][2]
    for impl in ("ours", "ours_wcache"):
        got = res[impl]
        assert torch.equal(got["losses"], ref["losses"]), (impl, got["losses"], ref["losses"])
        assert set(got["params"]) == set(ref["params"])
        for n in ref["params"]:
            assert torch.equal(got["params"][n], ref["params"][n]), (impl, n)


def test_ddp_wrapped_model_step_is_unchanged(pkg):
    """BASELINE config 5 runs the kernels per GPU inside the unchanged outer DDP loop (utils/kd_trainer.py:257-277).
    One rank over RCCL: a DDP-wrapped harness model (bf16 autocast, drop-in quantizers, activation sharing + paired
    launches on) produces the same loss and gradients as the bare model -- the autograd nodes coexist with DDP's
    gradient hooks and buckets."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    import tiny_llama as TL
    import llm_qat_amd.utils_quant as UQ
    dist = _one_rank_group()
    try:
        ids = TL.deterministic_batch().cuda()
        bare = TL.load_deterministic(TL.TinyLlama(UQ, w_bits=4, a_bits=8, kv_bits=4).bfloat16()).cuda()
        wrapped = DDP(TL.load_deterministic(TL.TinyLlama(UQ, w_bits=4, a_bits=8, kv_bits=4).bfloat16()).cuda(), device_ids=[0])
        outs = []
        for m in (bare, wrapped):
            for _ in range(2):   # second step: DDP has rebuilt its buckets
                m.zero_grad(set_to_none=True)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    loss, logits = m(ids, labels=ids)
                loss.backward()
            outs.append((loss.detach(), logits.detach()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        for (n, p), (_, q) in zip(bare.named_parameters(), wrapped.module.named_parameters()):
            assert torch.equal(p.grad, q.grad), n
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kw", [{"gradient_as_bucket_view": True}, {"find_unused_parameters": True}, {"static_graph": True},
                                {"gradient_as_bucket_view": True, "static_graph": True}], ids=lambda k: "+".join(k))
def test_ddp_variants_match_the_bare_eager_model(pkg, kw):
    """DDP's other modes around the drop-in -- gradients as views of the reducer's buckets (AccumulateGrad then writes INTO a bucket: the
    in-place weight-gradient path must hand it a tensor it may take or add), unused-parameter detection, static graph, and a first step
    under no_sync() with accumulation into existing .grad -- three steps, one rank over RCCL: losses and every parameter gradient equal
    the bare model on the live eager chain (utils/kd_trainer.py:257-277 is the reference's outer loop)."""
    from contextlib import nullcontext
    from torch.nn.parallel import DistributedDataParallel as DDP
    import tiny_llama as TL
    import llm_qat_amd.utils_quant as UQ
    dist = _one_rank_group()
    try:
        ids = TL.deterministic_batch().cuda()

        def run(Q, wrap):
            pkg.reset_learned_state()
            m = TL.load_deterministic(TL.TinyLlama(Q, w_bits=4, a_bits=8, kv_bits=4).bfloat16()).cuda()
            if wrap:
                m = DDP(m, device_ids=[0], **kw)
            losses = []
            for step in range(3):
                m.zero_grad(set_to_none=(step != 1))          # step 1 accumulates into step 0's gradients
                with (m.no_sync() if (wrap and step == 0 and "static_graph" not in kw) else nullcontext()):
                    with torch.autocast("cuda", dtype=torch.bfloat16):
                        loss, _ = m(ids, labels=ids)
                    loss.backward()
                losses.append(loss.detach())
            return losses, [p.grad.clone() for p in (m.module if wrap else m).parameters()]

        pkg.set_semantics("device_eager")
        want = run(TL.EagerQuant(), False)
        pkg.stats(reset=True)
        got = run(UQ, True)
        assert all(torch.equal(a, b) for a, b in zip(want[0], got[0]))
        assert all(torch.equal(a, b) for a, b in zip(want[1], got[1]))
        assert pkg.stats().get("inplace_taken") == 3 * 14 and not any(k.startswith("inplace_refused") for k in pkg.stats())
    finally:
        pkg.set_semantics("cpu_eager")
        dist.destroy_process_group()


@pytest.mark.parametrize("autocast", [False, True])
def test_torch_compile_traces_through_the_quantizers(pkg, autocast):
    """HF's `--torch_compile` (utils/kd_trainer.py:281-286): while Dynamo traces, the quantizers are torch.library
    custom ops (llm_qat_amd/compiled.py) -- a QuantizeLinear and a whole harness model compile into ONE graph with no
    graph break, and (backend aot_eager: same ATen kernels around them) give the eager results bit for bit."""
    import torch._dynamo as dynamo
    import tiny_llama as TL
    import llm_qat_amd.utils_quant as UQ
    dynamo.reset()
    torch.manual_seed(0)
    for w_bits, a_bits, sym in ((4, 8, True), (8, 8, False), (2, 8, True), (1, 32, True)):
        lin = UQ.QuantizeLinear(512, 256, symmetric=sym, w_bits=w_bits, a_bits=a_bits).cuda().bfloat16()
        x = (torch.randn(8, 512, device="cuda") * 1.5).bfloat16().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            ref = lin(x)
        ref.float().square().mean().backward()
        gw, gx = lin.weight.grad.clone(), x.grad.clone()
        lin.zero_grad(set_to_none=True)
        x.grad = None
        exp = dynamo.explain(lin)(x)
        assert exp.graph_count == 1 and exp.graph_break_count == 0, (w_bits, a_bits, sym, [str(r.reason)[:120] for r in exp.break_reasons])
        clin = torch.compile(lin, backend="aot_eager", fullgraph=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            out = clin(x)
        out.float().square().mean().backward()
        tag = f"w{w_bits} a{a_bits} sym={sym} autocast={autocast}"
        assert torch.equal(out, ref), tag
        assert torch.equal(lin.weight.grad, gw) and torch.equal(x.grad, gx), tag
    # the whole harness model, K/V hooks included (both call-site forms)
    ids = TL.deterministic_batch().cuda()
    for kv_one in (False, True):
        TL.KV_ONE_LAUNCH = kv_one
        try:
            model = TL.load_deterministic(TL.TinyLlama(UQ, w_bits=4, a_bits=8, kv_bits=4).bfloat16()).cuda()
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                loss, logits = model(ids, labels=ids)
            loss.backward()
            want = [p.grad.clone() for p in model.parameters()]
            model.zero_grad(set_to_none=True)
            dynamo.reset()
            cmodel = torch.compile(model, backend="aot_eager", fullgraph=True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                closs, clogits = cmodel(ids, labels=ids)
            closs.backward()
            assert torch.equal(closs, loss) and torch.equal(clogits, logits), (kv_one, closs.item(), loss.item())
            for (n, p), w in zip(model.named_parameters(), want):
                assert torch.equal(p.grad, w), (kv_one, n)
        finally:
            TL.KV_ONE_LAUNCH = False
    dynamo.reset()


def test_foreign_pending_hip_error_is_neither_cleared_nor_blamed(pkg):
    """rounds 1-3 drained the thread's hipGetLastError() slot before every launch (hiding other libraries'
    failures) and read it afterwards.  Since round 4 the launch status is hipLaunchKernel()'s return value: with somebody else's error
    pending, a launch of ours succeeds, says so, computes the right values -- and the foreign error is still there for its owner."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipGetLastError.restype = hip.hipPeekAtLastError.restype = hip.hipSetDevice.restype = ctypes.c_int
    x = torch.randn(64, 512, device="cuda").bfloat16()
    g = torch.ones_like(x)
    want = pkg.ops.sym_quantize(x, 8)
    torch.cuda.synchronize()
    assert hip.hipGetLastError() == 0
    foreign = hip.hipSetDevice(12345)            # somebody else's failed call: invalid device ordinal
    assert foreign != 0 and hip.hipPeekAtLastError() == foreign
    try:
        got = pkg.ops.sym_quantize(x, 8)         # raises if the library reports a launch failure
        y, side, rows, cols = pkg.ops.train_forward("sym", x, 8, False, -2.0, 2.0)
        gx = pkg.ops.train_backward(g, side, rows, cols, -2.0, 2.0)   # (no torch KERNEL runs in between: torch checks the slot after its own launches and would raise -- that is between torch and the error's owner)
        assert hip.hipPeekAtLastError() == foreign, "the library cleared an error that was not its own"
    finally:
        assert hip.hipGetLastError() == foreign   # the owner collects it
    assert hip.hipGetLastError() == 0
    torch.cuda.synchronize()
    assert torch.equal(got, want) and torch.equal(y, want) and gx.shape == x.shape


def test_double_backward_matches_the_reference(pkg):
    """VERDICT r03 "missing" item 4: the reference's backward is built from differentiable ATen ops (utils_quant.py:83-87), so
    `create_graph=True` works there.  The drop-in's backward launches kernels; asked for a graph it re-expresses the same values as
    `where(keep, g, 0)` with the mask learned from one extra launch on ones -- first and second derivatives equal the eager chain's,
    through SymQuantizer / AsymQuantizer in every backward data flow, through QuantizeLinear (pair node, in-place path never taken),
    and under autocast (fp32 gradient in, 16-bit gradient out)."""
    from llm_qat_amd.utils_quant import AsymQuantizer, QuantizeLinear, SymQuantizer
    from oracle import eager_chain as E
    clip = torch.tensor([-2.0, 2.0])
    gen = torch.Generator(device="cuda").manual_seed(11)
    prev = pkg.get_backward_mode()
    pkg.set_semantics("device_eager")
    try:
        for mode in ("mask", "bounds", "plain"):
            pkg.set_backward_mode(mode)
            for ours, ref in ((SymQuantizer, E.EagerSym), (AsymQuantizer, E.EagerAsym)):
                for dt, ac in ((torch.float32, False), (torch.bfloat16, False), (torch.bfloat16, True)):
                    x0 = (torch.randn(8, 256, generator=gen, device="cuda") * 1.5).to(dt)
                    g0 = torch.randn(8, 256, generator=gen, device="cuda")
                    res = []
                    for q in (ours, ref):
                        x = x0.clone().requires_grad_(True)
                        g = g0.clone().requires_grad_(True)
                        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=ac):
                            y = q.apply(x, clip, 8, False)
                        (gx,) = torch.autograd.grad(y, x, grad_outputs=g.to(y.dtype), create_graph=True)
                        assert gx.requires_grad
                        # a second-order quantity: d/dg of sum(gx^2) = 2 * keep * gx
                        (gg,) = torch.autograd.grad(gx.float().square().sum(), g)
                        res.append((y.detach(), gx.detach(), gg))
                    for a, b in zip(res[0], res[1]):
                        assert a.dtype == b.dtype and torch.equal(a, b), (mode, ours.__name__, dt, ac)
                    assert int((res[0][1] == 0).sum()) > 0
        pkg.set_backward_mode("mask")
        lin = QuantizeLinear(256, 64, w_bits=4, a_bits=8).cuda()
        x = torch.randn(16, 256, device="cuda", requires_grad=True)
        out = lin(x)
        gx, gw = torch.autograd.grad(out.square().sum(), (x, lin.weight), create_graph=True)
        pkg.stats(reset=True)
        (gx.square().sum() + gw.square().sum()).backward()      # flows through F.linear's double backward and the fake-quant masks
        assert lin.weight.grad is not None and x.grad is not None and torch.isfinite(x.grad).all()
    finally:
        pkg.set_backward_mode(prev)
        pkg.set_semantics("cpu_eager")


# ------------------------------------------------------------------------------------------ saved tensors
def test_side_buffers_are_saved_tensors_visible_to_hooks(pkg):
    """The row bounds + STE mask a training-mode forward records are SAVED tensors (ctx.save_for_backward), so
    torch.autograd.graph.saved_tensors_hooks sees them: pack/unpack round-trips them (here through the CPU, as
    save_on_cpu does) and the gradients stay bit-identical."""
    from llm_qat_amd.utils_quant import QuantizeLinear, SymQuantizer, quantize_kv
    torch.manual_seed(3)
    clip = torch.tensor([-2.0, 2.0])
    x0 = (torch.randn(4, 64, 1024, device="cuda") * 1.3).bfloat16()
    lin = QuantizeLinear(1024, 512, w_bits=4, a_bits=8).cuda().bfloat16()

    def run(hooks):
        seen = []

        def pack(t):
            seen.append((t.dtype, t.numel(), t.device.type))
            return t.cpu()

        def unpack(t):
            return t.cuda()

        x = x0.clone().requires_grad_(True)
        lin.zero_grad(set_to_none=True)
        import contextlib
        ctx = torch.autograd.graph.saved_tensors_hooks(pack, unpack) if hooks else contextlib.nullcontext()
        with ctx:
            y = SymQuantizer.apply(x, clip, 8, False)
            k, v = quantize_kv(y, y * 0.5, clip, clip, 4)
            out = lin(k + v)
        out.float().square().mean().backward()
        return x.grad.clone(), lin.weight.grad.clone(), seen

    gx0, gw0, _ = run(False)
    gx1, gw1, seen = run(True)
    assert torch.equal(gx0, gx1) and torch.equal(gw0, gw1)
    side = [s for s in seen if s[0] == torch.uint8]
    # SymQuantizer.apply: 1 side buffer; quantize_kv: 2 (K, V); QuantizeLinear's pair: 2 (weight, input)
    assert len(side) == 5, seen
    rows, cols = 4 * 64, 1024
    assert (torch.uint8, rows * 8 + rows * cols // 8, "cuda") in side       # bounds + 1 bit per element
    # the bf16 quantizer inputs themselves are NOT among the saved tensors of the quantizer nodes (only F.linear / mul save 16-bit tensors)
    n_x = sum(1 for s in seen if s[0] == torch.bfloat16 and s[1] == x0.numel())
    assert n_x <= 3, seen   # (k + v) operand of F.linear, y for `y * 0.5`... never one per quantizer


@pytest.mark.parametrize("reentrant", [False, True])
def test_checkpoint_drops_first_pass_side_buffers(pkg, reentrant):
    """Under activation checkpointing the first forward's side buffers must not survive until the backward: with
    non-reentrant checkpointing the saved-tensor hooks discard them (they are saved tensors now), with reentrant
    checkpointing the first pass runs without grad and records nothing.  Measured: memory held between forward and
    backward of a checkpointed block of 6 quantizers is the block's input + output only."""
    from llm_qat_amd.utils_quant import SymQuantizer
    clip = torch.tensor([-2.0, 2.0])
    rows, cols = 2048, 4096
    pkg.reset_learned_state()      # (what an earlier test's last forward left remembered would be let go of inside the measured region)
    x = (torch.randn(rows, cols, device="cuda") * 1.2).bfloat16().requires_grad_(True)

    def block(t):
        for _ in range(6):
            t = SymQuantizer.apply(t, clip, 8, False) * 1.0009765625
        return t

    def held(fn):
        torch.cuda.synchronize()
        base = torch.cuda.memory_allocated()
        y = fn()
        torch.cuda.synchronize()
        h = torch.cuda.memory_allocated() - base
        y.float().sum().backward()
        g = x.grad.clone()
        x.grad = None
        del y
        return h, g

    side = rows * 8 + rows * cols // 8
    tensor = rows * cols * 2
    h_plain, g_plain = held(lambda: block(x))
    h_ckpt, g_ckpt = held(lambda: checkpoint(block, x, use_reentrant=reentrant))
    assert torch.equal(g_plain, g_ckpt)
    assert h_plain >= 6 * side + tensor                # no checkpointing: every quantizer's side buffer lives until its backward
    assert h_ckpt <= tensor + side // 2, (h_ckpt, tensor, side)   # checkpointed: the output only -- no first-pass side buffer retained


def test_frozen_operand_gets_a_result_that_needs_no_grad(pkg):
    """a frozen weight (or an input without grad) must not make F.linear's backward run a GEMM whose result
    the quantizer node throws away: the paired node marks that output non-differentiable, like SymQuantizer.apply on a
    tensor that needs no grad in the reference."""
    from llm_qat_amd.utils_quant import QuantizeLinear
    from llm_qat_amd import utils_quant as U
    lin = QuantizeLinear(1024, 512, w_bits=4, a_bits=8).cuda().bfloat16()
    x = torch.randn(8, 1024, device="cuda").bfloat16()
    seen = {}
    orig = torch.nn.functional.linear

    def spy(inp, w, b=None):          # what the module hands to F.linear: (quantized weight, quantized input)
        seen["req"] = (w.requires_grad, inp.requires_grad)
        return orig(inp, w, b)

    for node in ("c++", "python"):
        assert pkg.cpp_node(node == "c++") == (node == "c++"), pkg.host_node()
        torch.nn.functional.linear = spy
        try:
            lin.weight.grad = None
            lin.weight.requires_grad_(False)
            xg = x.clone().requires_grad_(True)
            lin(xg).float().sum().backward()
            assert seen["req"] == (False, True) and xg.grad is not None and lin.weight.grad is None, node
            lin.weight.requires_grad_(True)
            lin(x).float().sum().backward()
            assert seen["req"] == (True, False) and lin.weight.grad is not None, node
        finally:
            torch.nn.functional.linear = orig
            pkg.cpp_node(True)


def test_autocast_dtype_differs_from_tensor_dtype(pkg):
    """an fp16 tensor inside autocast(bf16) (or the reverse).  The reference returns fp32 from SymQuantizer and
    F.linear's autocast cast rounds ONCE to the autocast dtype; the drop-in must not round to the tensor dtype in between."""
    from llm_qat_amd.utils_quant import QuantizeLinear
    torch.manual_seed(5)
    for tdt, adt in ((torch.float16, torch.bfloat16), (torch.bfloat16, torch.float16)):
        lin = QuantizeLinear(512, 256, w_bits=4, a_bits=8).cuda().to(tdt)
        x = torch.randn(16, 512, device="cuda").to(tdt)
        with torch.autocast("cuda", dtype=adt):
            got = lin(x)
            # the reference's op chain, live ATen (utils_quant.py:53-59,:71-72,:250)
            def sym(t, bits):
                m = torch.max(torch.abs(t), dim=-1, keepdim=True)[0].expand_as(t).detach()
                s = (2 ** (bits - 1) - 1) / (m + 1e-6)
                return torch.round(t * s).div(s + 1e-6)
            want = F.linear(sym(x, 8), sym(lin.weight, 4))
        assert got.dtype == want.dtype == adt
        assert torch.equal(got, want), (tdt, adt, (got.float() - want.float()).abs().max())


@pytest.mark.parametrize("cols", [256, 11008])
def test_weight_cache_with_autocast_dtype_mismatch_gradients(pkg, cols):
    """weight cache ON + an fp16 weight inside autocast(bf16) (or the reverse) + weight entries beyond the
    STE clip.  Round 2's cache took the fp32-result forward (its own mask layout) and ran the 16-bit-layout backward on it:
    rows that clip read another row's mask bits whenever the two row strides differed (cols = 256: 8 vs 4 words; 11008: 176 vs
    172).  Now the mismatch bypasses the cache (and every mask is the same row bitmap anyway): forward values and BOTH
    gradients bit-identical to the cache-off path, in-place weight gradients on and off, pairing on and off."""
    from llm_qat_amd.utils_quant import QuantizeLinear

    def run(tdt, adt, cache, inplace, pairing):
        torch.manual_seed(11)
        lin = QuantizeLinear(cols, 64, w_bits=4, a_bits=8).cuda().to(tdt)
        with torch.no_grad():
            lin.weight[3, 5], lin.weight[3, cols - 1], lin.weight[40, 0], lin.weight[63, cols // 2] = 2.5, -2.0, 2.0, -7.0
        x = (torch.randn(4, 8, cols, device="cuda") * 1.3).to(tdt).requires_grad_(True)
        pkg.enable_weight_quant_cache(cache)
        pkg.inplace_weight_grad(inplace)
        pkg.pair_operands(pairing)
        try:
            outs = []
            for _ in range(2):        # second pass: the cache (if it engaged) serves the weight
                lin.weight.grad = x.grad = None
                with torch.autocast("cuda", dtype=adt):
                    y = lin(x)
                y.float().square().sum().backward()
                outs.append((y.detach().clone(), x.grad.clone(), lin.weight.grad.clone()))
        finally:
            pkg.enable_weight_quant_cache(False)
            pkg.inplace_weight_grad(True)
            pkg.pair_operands(True)
        return outs

    for tdt, adt in ((torch.float16, torch.bfloat16), (torch.bfloat16, torch.float16), (torch.bfloat16, torch.bfloat16)):
        base = run(tdt, adt, cache=False, inplace=False, pairing=False)
        gw = base[0][2]
        assert gw[3, 5] == 0 and gw[3, cols - 1] == 0 and gw[40, 0] == 0 and gw[63, cols // 2] == 0 and gw[3, 6] != 0   # the STE clip at work
        for cache, inplace, pairing in ((True, True, True), (True, False, False), (True, True, False), (False, True, True)):
            got = run(tdt, adt, cache, inplace, pairing)
            for (ya, xa, wa), (yb, xb, wb) in zip(got, base):
                assert ya.dtype == yb.dtype and torch.equal(ya, yb), (tdt, adt, cache, inplace, pairing)
                assert torch.equal(xa, xb) and torch.equal(wa, wb), (tdt, adt, cache, inplace, pairing)


@pytest.mark.parametrize("node", ["c++", "python"])
def test_inplace_weight_gradient_is_guarded(pkg, node):
    """VERDICT r02 item 10 / ADVICE: handing F.linear's wgrad on by reference (masked in place) deviates from PyTorch's "never
    modify grad_outputs in place" rule; it is taken only when the gradient tensor owns its whole storage (no view, offset 0,
    contiguous) and nobody can observe it: a tensor hook or retain_grad() on the quantized weight, anomaly mode, or a view into a
    larger buffer fall back to the copy.  Gradients are identical either way; observers see the UNMASKED gradient they are entitled to.
    Both autograd nodes (the C++ one of csrc/fq_autograd_node.cpp with its guard on C++ reference counts, the Python `_PairNode`):
    what the guard decided is read from llm_qat_amd.stats()."""
    from llm_qat_amd import utils_quant as U
    from llm_qat_amd.utils_quant import QuantizeLinear
    assert pkg.cpp_node(node == "c++") == (node == "c++"), pkg.host_node()
    try:
        _inplace_guard_body(pkg, U, QuantizeLinear, node)
    finally:
        pkg.cpp_node(True)
        pkg.inplace_weight_grad(True)


def _inplace_guard_body(pkg, U, QuantizeLinear, node):
    torch.manual_seed(13)
    lin = QuantizeLinear(512, 128, w_bits=4, a_bits=8).cuda().bfloat16()
    with torch.no_grad():
        lin.weight[7, 9] = 2.25          # a weight element beyond the clip: its gradient is masked
    x = (torch.randn(16, 512, device="cuda") * 1.2).bfloat16().requires_grad_(True)

    def grads(hook=None, anomaly=False):
        """-> weight grad, input grad, what the hook saw, the guard's decision (True: in place, False: refused, None: not asked)"""
        lin.weight.grad = x.grad = None
        seen = {}
        orig = torch.nn.functional.linear

        def spy(inp, w, b=None):          # catch the quantized weight the module hands to F.linear
            if hook == "clone" and w.requires_grad:
                w.register_hook(lambda g: seen.__setitem__("g", g.clone()))
            if hook == "stash" and w.requires_grad:
                w.register_hook(lambda g: seen.__setitem__("g", g))      # keeps the very tensor the node is about to receive
            return orig(inp, w, b)
        torch.nn.functional.linear = spy
        pkg.stats(reset=True)
        try:
            with torch.autograd.set_detect_anomaly(anomaly, check_nan=False):
                y = lin(x)
                y.float().square().sum().backward()
        finally:
            torch.nn.functional.linear = orig
        st = pkg.stats(reset=True)
        assert (st.get("cpp_pair_backward", 0) == 1) == (node == "c++"), (node, st)
        refused = [k for k in st if k.startswith("inplace_refused")]
        took = True if st.get("inplace_taken") else (False if refused else None)
        return lin.weight.grad.clone(), x.grad.clone(), seen.get("g"), took, refused

    pkg.inplace_weight_grad(False)
    want_w, want_x, _, took, _ = grads()
    assert took is None
    pkg.inplace_weight_grad(True)
    got_w, got_x, _, took, why = grads()
    assert took is True, f"the guarded in-place path never engages in the plain module flow: the guard is miscalibrated on this device ({why})"
    assert torch.equal(got_w, want_w) and torch.equal(got_x, want_x) and got_w[7, 9] == 0
    # a hook that clones: it observes the gradient BEFORE the STE mask (as with the reference's clone): not zero at [7, 9]
    got_w, got_x, seen, took, _ = grads(hook="clone")
    assert torch.equal(got_w, want_w) and torch.equal(got_x, want_x)
    assert seen is not None and seen[7, 9] != 0, "the hook saw a gradient that had already been masked in place"
    # a hook that keeps the tensor itself: somebody else holds the gradient -> the node copies, the stashed tensor stays unmasked
    got_w, got_x, seen, took, why = grads(hook="stash")
    assert took is False and torch.equal(got_w, want_w) and torch.equal(got_x, want_x), why
    assert seen[7, 9] != 0, "a gradient someone else holds was modified in place"
    # anomaly mode keeps gradients for its diagnostics: never in place
    got_w, got_x, _, took, why = grads(anomaly=True)
    assert took is False and why == ["inplace_refused:anomaly"] and torch.equal(got_w, want_w) and torch.equal(got_x, want_x), why
    if node == "c++":
        return
    # the ownership test itself
    g = torch.randn(128, 512, device="cuda").bfloat16()
    assert U._owns_storage(g)
    assert not U._owns_storage(g[1:])                       # a view (storage offset)
    assert not U._owns_storage(g.t())                       # non-contiguous
    v = g.view(64, 1024)
    assert U._owns_storage(v) and not U._inplace_ok(v)      # a view of a tensor somebody holds (here: `g`)
    v2 = torch.randn(128, 512, device="cuda").bfloat16().view(64, 1024)
    assert U._inplace_ok(v2)                                # a view of a temporary nobody else can reach (F.linear's wgrad)
    flat = torch.randn(128 * 512 + 8, device="cuda").bfloat16()
    assert not U._owns_storage(flat[8:].view(128, 512))
    with torch.autograd.detect_anomaly(check_nan=False):
        assert not U._inplace_ok(g)
    pkg.stats(reset=True)
    assert not U._inplace_ok(g) and pkg.stats().get("inplace_refused:storage_refs") == 1   # `v` above still aliases g's storage (round 4: the storage's own holders count)
    del v
    assert U._inplace_ok(g)
    # an alias made WITHOUT view tracking (its own TensorImpl, no _base) shares the storage: the storage's own holder
    # count shows it
    alias = g.data
    pkg.stats(reset=True)
    assert not U._inplace_ok(g) and pkg.stats().get("inplace_refused:storage_refs") == 1, pkg.stats()
    del alias
    assert U._inplace_ok(g)
    # the guarded path engages for EVERY weight node, not only the pair node: _SymQuantizerWeight (operands quantized separately)
    # and _ReuseQuantizedWeight (weight cache), and llm_qat_amd.stats() counts it
    for setup, teardown in (((lambda: pkg.pair_operands(False)), (lambda: pkg.pair_operands(True))),
                            ((lambda: pkg.enable_weight_quant_cache(True)), (lambda: pkg.enable_weight_quant_cache(False)))):
        setup()
        try:
            pkg.stats(reset=True)
            lin.weight.grad = x.grad = None
            lin(x).float().square().sum().backward()
            st = pkg.stats()
            assert st.get("inplace_taken") == 1 and not any(k.startswith("inplace_refused") for k in st), st
            assert torch.equal(lin.weight.grad, want_w) and torch.equal(x.grad, want_x)
        finally:
            teardown()


# ------------------------------------------------------------------------------------------ N-tensor launches
@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp32"])
def test_multi_tensor_launch_matches_separate_calls(pkg, dtype):
    """fq_sym_fwd_multi / fq_ste_bwd_mask_multi on 2, 3 and 4 tensors of one row length (q/k/v weights + their shared
    input): values, side buffers' effect and gradients bit-identical to one call per tensor."""
    ops = pkg.ops
    g = torch.Generator(device="cuda").manual_seed(9)
    cols = 1024
    mk = lambda rows, sc: (torch.randn(rows, cols, generator=g, device="cuda") * sc).to(TD[dtype])  # noqa: E731
    tensors = [mk(384, 0.02), mk(3 * 50, 1.3).view(3, 50, cols), mk(256, 0.02), mk(512, 0.9)]
    tensors[3][5, 7] = 2.5
    bits = [4, 8, 4, 3]
    for n in (2, 3, 4):
        ts, bs = tensors[:n], bits[:n]
        res = ops.multi_forward(ts, bs, [True] * n, -2.0, 2.0)
        assert res is not None
        ys, sides, rows, c = res
        grads = [torch.randn_like(t) for t in ts]
        gxs = ops.multi_backward(grads, sides, rows, c, -2.0, 2.0)
        for t, b, y, gr, gx in zip(ts, bs, ys, grads, gxs):
            y1, s1, r1, c1 = ops.train_forward("sym", t, b, False, -2.0, 2.0)
            assert torch.equal(y.view(torch.uint8), y1.view(torch.uint8))
            assert torch.equal(gx.view(torch.uint8), ops.train_backward(gr, s1, r1, c1, -2.0, 2.0).view(torch.uint8))
        # a subset of the gradients (the others None): only those are produced
        part = ops.multi_backward([grads[0]] + [None] * (n - 1), sides, rows, c, -2.0, 2.0)
        assert part[1] is None and torch.equal(part[0], gxs[0])
    # a tensor that needs no backward gets no side buffer
    ys, sides, rows, c = ops.multi_forward(tensors[:3], bits[:3], [True, False, True], -2.0, 2.0)
    assert sides[1] is None and sides[0] is not None
    # not served: different row lengths / dtypes / too many tensors -> None (the caller falls back to separate calls)
    assert ops.multi_forward([tensors[0], tensors[0][:, :512].contiguous()], [4, 4], [False, False], -2.0, 2.0) is None
    assert ops.multi_forward(tensors + [tensors[0]], bits + [4], [False] * 5, -2.0, 2.0) is None


# ------------------------------------------------------------------------------------------ weight gradients by reference
def test_inplace_ste_backward_kernel(pkg):
    """fq_ste_bwd_mask with gx == g: the gradient is masked where it stands; a tensor whose rows cannot clip is not touched
    (bitwise unchanged), clippable rows get exactly the out-of-place result."""
    ops = pkg.ops
    g0 = torch.Generator(device="cuda").manual_seed(21)
    for dt in (torch.bfloat16, torch.float32, torch.float16):
        w = (torch.randn(300, 2048, generator=g0, device="cuda") * 0.02).to(dt)          # weight-like: no row reaches the clip
        a = (torch.randn(300, 2048, generator=g0, device="cuda") * 1.2).to(dt)           # activation-like: every row does
        mix = w.clone()
        mix[7, 11], mix[7, 12], mix[200, 0] = 2.0, -3.5, 2.5                               # two clippable rows among safe ones
        for x in (w, a, mix):
            y, side, rows, cols = ops.train_forward("sym", x, 4, False, -2.0, 2.0)
            g = torch.randn(300, 2048, generator=g0, device="cuda").to(dt)
            want = ops.train_backward(g.clone(), side, rows, cols, -2.0, 2.0)
            gi = g.clone()
            got = ops.train_backward(gi, side, rows, cols, -2.0, 2.0, inplace=True)
            assert got.data_ptr() == gi.data_ptr()                                         # handed on by reference
            assert torch.equal(got.view(torch.uint8), want.view(torch.uint8))
            ref = torch.where((x >= 2) | (x <= -2), torch.zeros_like(g), g)
            assert torch.equal(got, ref)
    # rows long enough for two chunks per row (grid y = 2), 1 to 600 rows (in-place blocks cover 256 rows each), clippable rows at block edges
    for rows_, cols_ in ((1, 32768), (255, 32768), (257, 24576), (600, 32768)):
        x = (torch.randn(rows_, cols_, generator=g0, device="cuda") * 0.02).bfloat16()
        for r in sorted({0, rows_ // 2, rows_ - 1, min(255, rows_ - 1), min(256, rows_ - 1)}):
            x[r, 5], x[r, cols_ - 3], x[r, cols_ // 2 + 1] = 2.0, -2.5, 3.0
        y, side, rows, cols = ops.train_forward("sym", x, 4, False, -2.0, 2.0)
        g = torch.randn(rows_, cols_, generator=g0, device="cuda").bfloat16()
        ref = torch.where((x >= 2) | (x <= -2), torch.zeros_like(g), g)
        gi = g.clone()
        assert torch.equal(ops.train_backward(gi, side, rows, cols, -2.0, 2.0, inplace=True), ref) and torch.equal(ops.train_backward(g, side, rows, cols, -2.0, 2.0), ref)
    # pair + multi launches: the weight slots in place, the activation slot to a fresh tensor
    xs = (torch.randn(64, 2048, generator=g0, device="cuda") * 1.3).bfloat16()
    w2 = (torch.randn(128, 2048, generator=g0, device="cuda") * 0.02).bfloat16()
    res = ops.multi_forward([mix.bfloat16(), xs, w2], [4, 8, 4], [True] * 3, -2.0, 2.0)
    ys, sides, rows, cols = res
    gs = [torch.randn_like(t) for t in (mix.bfloat16(), xs, w2)]
    want = ops.multi_backward([g.clone() for g in gs], sides, rows, cols, -2.0, 2.0)
    gi = [g.clone() for g in gs]
    got = ops.multi_backward(gi, sides, rows, cols, -2.0, 2.0, inplace=[True, False, True])
    assert got[0].data_ptr() == gi[0].data_ptr() and got[2].data_ptr() == gi[2].data_ptr() and got[1].data_ptr() != gi[1].data_ptr()
    for a_, b_ in zip(got, want):
        assert torch.equal(a_, b_)
    assert torch.equal(gi[1], gs[1])                                                       # the activation's incoming gradient is untouched


@pytest.mark.parametrize("autocast", [False, True])
def test_weight_gradient_by_reference_is_transparent(pkg, autocast):
    """QuantizeLinear hands its weight's gradient on by reference (masked in place, default) -- same gradients, bit for
    bit, as with inplace_weight_grad(False), for paired, grouped and separate-weight flows, incl. a weight that reaches the clip"""
    from llm_qat_amd.utils_quant import QuantizeLinear

    def run(inplace, pairing):
        torch.manual_seed(2)
        mods = [QuantizeLinear(1024, 512, w_bits=4, a_bits=8).cuda().bfloat16() for _ in range(3)]
        with torch.no_grad():
            mods[1].weight[5, 9] = 2.25          # a weight element beyond the clip: its gradient must be zero
        opt = torch.optim.SGD([m.weight for m in mods], lr=0.01)
        pkg.inplace_weight_grad(inplace)
        pkg.pair_operands(pairing)
        out = []
        try:
            for step in range(2):
                x = (torch.randn(2, 32, 1024, device="cuda", generator=torch.Generator(device="cuda").manual_seed(step)) * 1.4).bfloat16().requires_grad_(True)
                opt.zero_grad(set_to_none=True)
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                    loss = qkv_loss(mods, x)
                loss.backward()
                out.append((loss.detach().clone(), x.grad.clone(), [m.weight.grad.clone() for m in mods]))
                opt.step()
        finally:
            pkg.inplace_weight_grad(True)
            pkg.pair_operands(True)
        return out

    for pairing in (True, False):
        a, b = run(True, pairing), run(False, pairing)
        for (la, xa, wa), (lb, xb, wb) in zip(a, b):
            assert torch.equal(la, lb) and torch.equal(xa, xb)
            for p, q in zip(wa, wb):
                assert torch.equal(p, q)
            assert wa[1][5, 9] == 0
