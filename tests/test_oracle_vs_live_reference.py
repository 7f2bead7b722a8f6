"""CPU tier, build container only (skipped where /root/reference is absent, e.g. on the GPU box): a DIFFERENTIAL run of the oracle
against the REAL reference code, live, on inputs no committed fixture holds.

The committed golden vectors (tests/golden/*.npz) are what travels; this test is the same comparison without the detour -- the
reference's own `SymQuantizer` / `AsymQuantizer` (models/utils_quant.py:31-162) executed here on freshly drawn tensors
(3 dtypes, bits 2..16, 1-D .. 4-D, layerwise, rows scaled from 1e-30 to 1e30 for fp32 / bf16 and across fp16's range, signed zeros,
NaN / Inf, all-zero rows, rows at and around the STE clip) against oracle/fq_oracle.c, bit for bit: forward values and the STE gradient.
And the autocast arithmetic: the real `SymQuantizer.apply` under CUDA autocast's cast policy (tests/autocast_policy.py, both scalar
policies) against `fqo_sym_fwd_autocast` / `fqo_ste_bwd_wide`.
"""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, bits_equal, mismatch_report
from oracle import oracle as O

sys.path.insert(0, os.path.join(ROOT, "tests"))
from autocast_policy import cuda_autocast_policy  # noqa: E402

REF = os.environ.get("LLMQAT_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "models", "utils_quant.py")), reason="reference checkout not present")
TD = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}


@pytest.fixture(scope="module")
def ref():
    for name in [m for m in sys.modules if m == "models" or m.startswith("models.")]:
        del sys.modules[name]
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    try:
        import models.utils_quant as R
    finally:
        sys.path.remove(REF)
    yield R
    for name in [m for m in sys.modules if m == "models" or m.startswith("models.")]:
        del sys.modules[name]


def to_np(t):
    t = t.detach().contiguous()
    return t.view(torch.int16).numpy().view(np.uint16).copy() if t.dtype in (torch.bfloat16, torch.float16) else t.numpy().copy()


def draw(rng, dtype, trial):
    """a tensor with adversarial rows; -> torch tensor of `dtype` (CPU)"""
    nd = int(rng.integers(1, 5))
    shape = tuple(int(rng.integers(1, 7)) for _ in range(nd - 1)) + (int(rng.choice([1, 2, 5, 8, 17, 64, 100, 257])),)
    scales = {"fp32": [1e-30, 1e-12, 1e-6, 1e-4, 0.02, 1.0, 2.0, 40.0, 1e6, 1e30], "bf16": [1e-30, 1e-12, 1e-6, 1e-4, 3e-4, 0.02, 1.0, 2.0, 40.0, 1e6, 1e30],
              "fp16": [1e-7, 1e-5, 1.2e-4, 2.4e-4, 1e-3, 0.02, 1.0, 2.0, 40.0, 3e3, 3e4]}[dtype]
    rows = int(np.prod(shape[:-1])) if nd > 1 else 1
    with np.errstate(over="ignore", invalid="ignore"):
        x = rng.standard_normal(shape).astype(np.float32) * rng.choice(scales, size=shape[:-1] + (1,)).astype(np.float32)
        flat = x.reshape(rows, shape[-1])
        for _ in range(int(rng.integers(0, 4))):
            r, c = int(rng.integers(0, rows)), int(rng.integers(0, shape[-1]))
            flat[r, c] = np.float32(rng.choice([0.0, -0.0, 2.0, -2.0, 1.9921875, -2.015625, np.nan, np.inf, -np.inf, 6e4, 1e-40]))
        if rng.random() < 0.2:
            flat[int(rng.integers(0, rows))] = 0.0
    return torch.from_numpy(x).to(TD[dtype])


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp32"])
def test_oracle_equals_the_live_reference_on_fresh_inputs(ref, dtype):
    rng = np.random.default_rng({"bf16": 101, "fp16": 102, "fp32": 103}[dtype])
    n = int(os.environ.get("LLMQAT_LIVE_TRIALS", "150"))
    for trial in range(n):
        x = draw(rng, dtype, trial)
        kind = "sym" if trial % 3 else "asym"
        bits = int(rng.choice([1, 2, 3, 4, 6, 8, 12, 16, 31])) if kind == "sym" else int(rng.choice([1, 2, 4, 8, 16, 24]))
        layerwise = bool(rng.random() < 0.2) and x.dim() <= 3
        lo, hi = [(-2.0, 2.0), (-0.5, 0.75), (-1.0, 1.0)][int(rng.integers(0, 3))]
        clip = torch.tensor([lo, hi])
        quant = ref.SymQuantizer if kind == "sym" else ref.AsymQuantizer
        xr = x.clone().requires_grad_(True)
        with np.errstate(all="ignore"):
            y = quant.apply(xr, clip, bits, layerwise)
            g = (torch.randn(x.shape, generator=torch.Generator().manual_seed(trial)) * 0.1).to(TD[dtype])
            y.backward(g)
        r, c = O.rows_cols(tuple(x.shape), layerwise)
        x_np = to_np(x)
        want = (O.sym_fwd(x_np, r, c, bits, dtype)[0] if kind == "sym" else O.asym_fwd(x_np, r, c, bits, dtype)[0])
        tag = f"trial {trial}: {kind} {dtype} {tuple(x.shape)} b{bits} lw={layerwise} clip=({lo},{hi})"
        assert bits_equal(want, to_np(y), dtype), f"{tag} forward: {mismatch_report(want, to_np(y), dtype)}"
        want_g = O.ste_bwd(to_np(g), x_np, lo, hi, dtype)
        assert bits_equal(want_g, to_np(xr.grad), dtype), f"{tag} gradient: {mismatch_report(want_g, to_np(xr.grad), dtype)}"


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("device_scalars", [False, True])
def test_oracle_equals_the_live_reference_under_the_autocast_policy(ref, dtype, device_scalars):
    """SymQuantizer.apply of the real reference under CUDA autocast's casts (fp32 behind the reciprocal, fp32 result) on fresh tensors"""
    rng = np.random.default_rng({"bf16": 201, "fp16": 202}[dtype] + (10 if device_scalars else 0))
    sem = O.SEM_DEVICE if device_scalars else O.SEM_CPU
    n = int(os.environ.get("LLMQAT_LIVE_TRIALS", "150")) // 2
    for trial in range(n):
        x = draw(rng, dtype, trial)
        if x.dim() > 3 and trial % 2:
            x = x.reshape(-1, x.shape[-1])
        bits = int(rng.choice([1, 2, 3, 4, 8, 12, 16, 31]))
        lo, hi = [(-2.0, 2.0), (-0.5, 0.75)][trial % 2]
        xr = x.clone().requires_grad_(True)
        with cuda_autocast_policy(TD[dtype], device_scalars), np.errstate(all="ignore"):
            y = ref.SymQuantizer.apply(xr, torch.tensor([lo, hi]), bits, False)
        assert y.dtype == torch.float32
        g = torch.randn(x.shape, generator=torch.Generator().manual_seed(trial)) * 0.1
        y.backward(g)      # the engine casts the fp32 gradient to the input's dtype: zeroing commutes with that cast
        r, c = O.rows_cols(tuple(x.shape), False)
        want, _ = O.sym_fwd_autocast(to_np(x), r, c, bits, dtype, wide=True, sem=sem)
        tag = f"trial {trial}: {dtype} {tuple(x.shape)} b{bits} device_scalars={device_scalars}"
        assert bits_equal(want.reshape(y.shape), y.detach().numpy(), "fp32"), f"{tag}: {mismatch_report(want.reshape(y.shape), y.detach().numpy(), 'fp32')}"
        want_g = O.ste_bwd_wide(g.numpy().copy(), to_np(x), lo, hi, dtype)
        assert bits_equal(want_g, to_np(xr.grad), dtype), f"{tag} gradient"


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp32"])
def test_low_bit_weight_branches_equal_the_live_reference(ref, dtype, monkeypatch):
    """QuantizeLinear's 1-/2-bit weight branches (:202-242): the weight the REAL module hands to F.linear, live, against the oracle's
    elementwise chain given the module's own mean-|w| scale (a float sum is order dependent: the scale is ATen's, as in w12.npz)."""
    import torch.nn.functional as F
    rng = np.random.default_rng({"bf16": 301, "fp16": 302, "fp32": 303}[dtype])
    seen = {}
    real = F.linear

    def spy(inp, weight, bias=None):
        seen["w"] = weight.detach().clone()
        return real(inp, weight, bias)

    monkeypatch.setattr(ref.nn.functional, "linear", spy)
    for trial in range(int(os.environ.get("LLMQAT_LIVE_TRIALS", "150")) // 3):
        out_f, in_f = int(rng.integers(1, 12)), int(rng.choice([1, 3, 8, 33, 128]))
        w_bits, layerwise = 1 + trial % 2, bool(rng.random() < 0.3)
        lin = ref.QuantizeLinear(in_f, out_f, w_bits=w_bits, a_bits=32, weight_layerwise=layerwise).to(TD[dtype])
        with torch.no_grad():
            w = rng.standard_normal((out_f, in_f)).astype(np.float32) * rng.choice([1e-3, 0.02, 1.0, 30.0], size=(out_f, 1)).astype(np.float32)
            if rng.random() < 0.5:
                w[int(rng.integers(0, out_f)), int(rng.integers(0, in_f))] = np.float32(rng.choice([0.0, -0.0, 1e-30]))
            lin.weight.copy_(torch.from_numpy(w).to(TD[dtype]))
            lin(torch.zeros(2, in_f, dtype=TD[dtype]))
            absmean = lin.weight.abs().mean() if layerwise else lin.weight.abs().mean(dim=1)
            sc = (absmean if w_bits == 1 else 2 * absmean).float().reshape(-1).numpy()
        w_np = to_np(lin.weight)
        if layerwise:
            want, _ = O.w12_fwd(w_np.reshape(1, -1), 1, out_f * in_f, w_bits, dtype, scale_in=sc)
        else:
            want, _ = O.w12_fwd(w_np, out_f, in_f, w_bits, dtype, scale_in=sc)
        got = to_np(seen["w"])
        assert bits_equal(want.reshape(got.shape), got, dtype), f"trial {trial}: {dtype} [{out_f},{in_f}] w_bits={w_bits} lw={layerwise}: {mismatch_report(want.reshape(got.shape), got, dtype)}"


def test_degenerate_shapes_cpu_tensor_path_vs_the_live_reference(ref):
    """the drop-in's opt-in CPU-tensor path (llm_qat_amd.allow_cpu_tensors) on tensors without elements / with one element: the live
    reference's result, or the live reference's exception type (tests/test_gpu_parity.py checks the same list on the GPU kernels'
    host path against the eager chain)"""
    import llm_qat_amd
    from llm_qat_amd.utils_quant import AsymQuantizer, SymQuantizer
    clip = torch.tensor([-2.0, 2.0])

    def run(q, shape, layerwise):
        x = torch.full(shape, 0.75).requires_grad_(True)
        try:
            y = q.apply(x, clip, 8, layerwise)
            y.sum().backward()
        except Exception as e:  # noqa: BLE001
            return type(e)
        return y.detach(), x.grad

    llm_qat_amd.allow_cpu_tensors(True)
    try:
        for shape in [(0, 8), (3, 0), (0,), (2, 0, 8), (2, 3, 0), (0, 0), (0, 3, 4, 8), (2, 3, 0, 8), (2, 0, 4, 8), (2, 3, 4, 0), (0, 0, 4, 8),
                      (1, 1), (1,), (5, 1), (), (1, 1, 1, 1), (1, 1, 1, 1, 2)]:
            for layerwise in (False, True):
                for rq, q in ((ref.SymQuantizer, SymQuantizer), (ref.AsymQuantizer, AsymQuantizer)):
                    want, got = run(rq, shape, layerwise), run(q, shape, layerwise)
                    tag = f"{q.__name__} {shape} layerwise={layerwise}"
                    if isinstance(want, type):
                        assert got is want, f"{tag}: reference raises {want.__name__}, drop-in gave {got}"
                    else:
                        assert not isinstance(got, type), f"{tag}: drop-in raised {got}"
                        assert all(a.shape == b.shape and torch.equal(a, b) for a, b in zip(got, want)), tag
    finally:
        llm_qat_amd.allow_cpu_tensors(False)
