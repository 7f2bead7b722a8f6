"""GPU tier: QuantizeLinear with every mix of weight dtype x input dtype x autocast dtype (fp32 master weights under bf16 autocast, fp16
tensors inside autocast(bf16), mismatched dtypes without autocast -- where F.linear raises in the reference too), and the module's life
cycle (dtype conversion, deepcopy, state_dict round trip, `.data` swap, Parameter replacement, optimizer steps with and without the
persistent weight cache, freezing a weight later, torch.save of the module) -- against the same module on the live eager chain
(tiny_llama.EagerQuant over oracle/eager_chain.py).  Output, input gradient and weight gradient bit-identical, or the same exception type.
"""
import copy
import io
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tiny_llama as TL  # noqa: E402

DT = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}


def same(a, b):
    return len(a) == len(b) and all((x is None and y is None) or (x is not None and y is not None and x.dtype == y.dtype and x.shape == y.shape
                                                                    and torch.equal(x.nan_to_num(), y.nan_to_num())) for x, y in zip(a, b))


def mk(Q, wdt, i=64, o=48):
    m = Q.QuantizeLinear(i, o, w_bits=4, a_bits=8).cuda().to(wdt)
    with torch.no_grad():
        m.weight.copy_((torch.randn(o, i, generator=torch.Generator().manual_seed(4)) * 0.4).cuda().to(wdt))
        m.weight[1, 2] = 2.5
    return m


def xin(xdt, i=64):
    return (torch.randn(3, 7, i, generator=torch.Generator().manual_seed(5)) * 1.5).cuda().to(xdt).requires_grad_(True)


def both(fn):
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ
    llm_qat_amd.set_semantics("device_eager")
    try:
        res = []
        for Q in (TL.EagerQuant(), UQ):
            llm_qat_amd.reset_learned_state()
            try:
                res.append(fn(Q))
            except Exception as e:  # noqa: BLE001
                res.append(type(e))
        return res
    finally:
        llm_qat_amd.set_semantics("cpu_eager")
        llm_qat_amd.reset_learned_state()


def check(fn, tag):
    want, got = both(fn)
    if isinstance(want, type) or isinstance(got, type):
        assert want is got, f"{tag}: eager chain -> {want}, drop-in -> {got}"
    else:
        assert same(want, got), tag


@pytest.mark.parametrize("xdt", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("wdt", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("acdt", [None, "bf16", "fp16"])
def test_weight_input_autocast_dtype_mixes(acdt, wdt, xdt):
    def f(Q):
        m, x = mk(Q, DT[wdt]), xin(DT[xdt])
        with torch.autocast("cuda", dtype=DT[acdt or "bf16"], enabled=acdt is not None):
            y = m(x)
        y.float().sum().backward()
        return [y.detach(), x.grad, m.weight.grad]
    check(f, f"autocast={acdt} weight={wdt} input={xdt}")


def test_conversion_deepcopy_state_dict():
    def f(Q):
        m, x = mk(Q, torch.float32), xin(torch.bfloat16)
        m = m.bfloat16()
        y = m(x)
        y.float().sum().backward()
        m2 = copy.deepcopy(m)
        m2.zero_grad()
        m3 = Q.QuantizeLinear(64, 48, w_bits=4, a_bits=8).cuda().bfloat16()
        m3.load_state_dict(m.state_dict())
        return [y.detach(), x.grad, m.weight.grad, m2(x.detach()).detach(), m3(x.detach()).detach()]
    check(f, "conversion / deepcopy / state_dict")


@pytest.mark.parametrize("weight_cache", [False, True])
def test_weight_replaced_under_the_module(weight_cache):
    """a `.data` swap (new storage, same Parameter, same version counter) and a replaced Parameter are never served a stale fake-quant"""
    import llm_qat_amd

    def swap(Q):
        m, x = mk(Q, torch.bfloat16), xin(torch.bfloat16)
        y0 = m(x.detach())
        with torch.no_grad():
            m.weight.data = (m.weight.data * 0.5).contiguous()
        y1 = m(x.detach())
        m.weight = torch.nn.Parameter(m.weight.detach() * 2.0)
        y2 = m(x)
        y2.float().sum().backward()
        return [y0.detach(), y1.detach(), y2.detach(), x.grad, m.weight.grad]

    def opt_steps(Q):
        m, x = mk(Q, torch.bfloat16), xin(torch.bfloat16)
        opt = torch.optim.SGD(m.parameters(), lr=0.1)
        outs = []
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            y = m(x)
            y.float().sum().backward()
            opt.step()
            outs.append(y.detach())
        return outs + [m.weight.detach()]

    llm_qat_amd.enable_weight_quant_cache(weight_cache, persistent=weight_cache)
    try:
        check(swap, f"weight swap (weight cache {weight_cache})")
        check(opt_steps, f"optimizer steps (weight cache {weight_cache})")
    finally:
        llm_qat_amd.enable_weight_quant_cache(False)


def test_weight_frozen_after_a_step_and_module_pickle():
    def frozen_later(Q):
        m, x = mk(Q, torch.bfloat16), xin(torch.bfloat16)
        y = m(x)
        y.float().sum().backward()
        m.weight.requires_grad_(False)
        m.weight.grad, x.grad = None, None
        y2 = m(x)
        y2.float().sum().backward()
        return [y.detach(), y2.detach(), x.grad]
    check(frozen_later, "weight frozen after a step")
    import llm_qat_amd.utils_quant as UQ
    m, x = mk(UQ, torch.bfloat16), xin(torch.bfloat16)
    buf = io.BytesIO()
    torch.save(m, buf)          # a file this test wrote itself
    buf.seek(0)
    m2 = torch.load(buf, weights_only=False)
    assert torch.equal(m(x.detach()), m2(x.detach()))


def test_low_bit_weight_branches_across_dtype_and_autocast_mixes():
    """QuantizeLinear with w_bits 1 / 2 (utils_quant.py:202-242) for every weight dtype x input dtype x autocast dtype, row-wise and
    layerwise, with and without activation fake-quant, shapes the one-launch kernel serves and shapes it does not: output, input gradient
    and weight gradient bit-identical to the branch's op chain run live by ATen (432 combinations)"""
    import torch.nn.functional as F
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ
    from oracle import eager_chain as E
    from test_gpu_features import eager_low_bit
    clip = torch.tensor([-2.0, 2.0])
    llm_qat_amd.set_semantics("device_eager")
    bad = []
    try:
        for acdt in (None, torch.bfloat16, torch.float16):
            for wdt in (torch.float32, torch.bfloat16, torch.float16):
                for xdt in (torch.float32, torch.bfloat16):
                    for wb in (1, 2):
                        for lw in (False, True):
                            for (o, i) in ((24, 300), (64, 4096), (3, 8)):
                                for ab in (8, 32):
                                    torch.manual_seed(0)
                                    lin = UQ.QuantizeLinear(i, o, w_bits=wb, a_bits=ab, weight_layerwise=lw).cuda().to(wdt)
                                    with torch.no_grad():
                                        lin.weight.copy_(torch.randn(o, i, device="cuda") * 0.05)
                                    x = torch.randn(5, i, device="cuda").to(xdt).requires_grad_(True)
                                    wref, xr = lin.weight.detach().clone().requires_grad_(True), x.detach().clone().requires_grad_(True)

                                    def ref():
                                        with torch.autocast("cuda", dtype=acdt or torch.bfloat16, enabled=acdt is not None):
                                            y = F.linear(E.EagerSym.apply(xr, clip, ab, False) if ab < 32 else xr, eager_low_bit(wref, wb, lw))
                                        y.float().sum().backward()
                                        return [y.detach(), xr.grad, wref.grad]

                                    def got():
                                        with torch.autocast("cuda", dtype=acdt or torch.bfloat16, enabled=acdt is not None):
                                            y = lin(x)
                                        y.float().sum().backward()
                                        return [y.detach(), x.grad, lin.weight.grad]

                                    res = []
                                    for f in (ref, got):
                                        try:
                                            res.append(f())
                                        except Exception as e:  # noqa: BLE001
                                            res.append(type(e))
                                    a, b = res
                                    ok = (a is b) if isinstance(a, type) or isinstance(b, type) else same(a, b)
                                    if not ok:
                                        bad.append((acdt, wdt, xdt, wb, lw, (o, i), ab))
        assert not bad, bad[:10]
    finally:
        llm_qat_amd.set_semantics("cpu_eager")
        llm_qat_amd.reset_learned_state()
