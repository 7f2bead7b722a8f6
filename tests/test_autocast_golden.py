"""CPU tier: the oracle's autocast arithmetic against tests/golden/autocast.npz -- fixtures the REAL reference produced (its own
SymQuantizer / QuantizeLinear, models/utils_quant.py:31-87,:165-254, run on CPU tensors under CUDA autocast's cast policy,
tests/golden/make_golden_autocast.py + tests/autocast_policy.py).  This is the arithmetic LLM-QAT trains with
(run_train.sh:17-18 --bf16 -> utils/kd_trainer.py:106).

Bar: bit-exact -- fp32 results, bin indices, per-row scales, the result rounded to the operand dtype, gradients.
"""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import bits_equal, golden, mismatch_report
from oracle import eager_chain as E
from oracle import oracle as O

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from autocast_policy import cuda_autocast_policy  # noqa: E402

TD = {"bf16": torch.bfloat16, "fp16": torch.float16}


def sems(case):
    """the oracle `sem` codes a fixture case pins ("both": the two scalar policies agree on these inputs)"""
    return {"cpu": (O.SEM_CPU,), "device": (O.SEM_DEVICE,), "both": (O.SEM_CPU, O.SEM_DEVICE)}[case["scalars"]]


def t16(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).view(TD[dtype])


def n16(t):
    return t.detach().contiguous().view(torch.int16).numpy().view(np.uint16)


def test_fixture_covers_what_it_should():
    G = golden("autocast.npz")
    names = {c["name"] for c in G.cases}
    assert G.meta["policy"]["fp32_cast"] == ["aten.reciprocal"] and "reference" in G.meta
    for dt in ("bf16", "fp16"):
        assert any(c["dtype"] == dt and c["scalars"] == "cpu" for c in G.cases) and any(c["dtype"] == dt and c["scalars"] == "device" for c in G.cases)
    for w in ("2x4096", "1x11008", "1x5120", "1x13824"):   # the model widths of BASELINE configs 1-3 and 5
        assert any(w in n for n in names), w
    lin = json_cases(G)
    assert {(c["dtype"], c["autocast_dtype"]) for c in lin} == {("bf16", "bf16"), ("fp16", "fp16"), ("fp16", "bf16"), ("bf16", "fp16")}


def json_cases(G):
    import json
    return json.loads(bytes(G.z["manifest"]).decode())["linear_cases"]


def test_oracle_autocast_forward_matches_reference_fixture():
    G = golden("autocast.npz")
    bad = []
    for c in G.cases:
        dt, bits = c["dtype"], c["bits"]
        x = G.arr(c, "x")
        rows, cols = O.rows_cols(tuple(c["shape"]), c["layerwise"])
        for sem in sems(c):
            y, idx, s = O.sym_fwd_autocast(x, rows, cols, bits, dt, wide=True, sem=sem, want_scale=True)
            yn, _ = O.sym_fwd_autocast(x, rows, cols, bits, dt, wide=False, sem=sem)
            tag = f"{c['name']} sem={sem}"
            if not (idx == G.arr(c, "idx")).all():
                bad.append(f"{tag}: idx {int((idx != G.arr(c, 'idx')).sum())} differ")
            if not bits_equal(y, G.arr(c, "y"), "fp32"):
                bad.append(f"{tag}: y {mismatch_report(y, G.arr(c, 'y'), 'fp32')}")
            if not bits_equal(yn, G.arr(c, "y_narrow"), dt):
                bad.append(f"{tag}: y_narrow {mismatch_report(yn, G.arr(c, 'y_narrow'), dt)}")
            if not bits_equal(s, G.arr(c, "scale"), "fp32"):
                bad.append(f"{tag}: scale")
    assert not bad, "\n".join(bad[:20])


def test_oracle_autocast_backward_matches_reference_fixture():
    """fp32 grad_output -> the gradient that reaches the 16-bit input (mask in the input's dtype, engine cast)"""
    G = golden("autocast.npz")
    n = 0
    for c in G.cases:
        if not c["grad"]:
            continue
        clip = G.arr(c, "clip")
        gx = O.ste_bwd_wide(G.arr(c, "g"), G.arr(c, "x"), float(clip[0]), float(clip[1]), c["dtype"])
        assert bits_equal(gx, G.arr(c, "gx"), c["dtype"]), f"{c['name']}: {mismatch_report(gx, G.arr(c, 'gx'), c['dtype'])}"
        # the same gradient from the 16-bit STE applied to the gradient rounded first (zeroing commutes with the cast):
        # what fq_ste_bwd_mask on a narrow result's gradient computes
        g16 = n16(torch.from_numpy(G.arr(c, "g")).to(TD[c["dtype"]]))
        assert bits_equal(O.ste_bwd(g16, G.arr(c, "x"), float(clip[0]), float(clip[1]), c["dtype"]), G.arr(c, "gx"), c["dtype"]), c["name"]
        n += 1
    assert n >= 60


def test_oracle_export_autocast_bins_are_the_reference_bins():
    """fqo_export(autocast=1): int16 container holds the reference's torch.round output wherever it fits"""
    G = golden("autocast.npz")
    for c in G.cases:
        if c["layerwise"] or len(c["shape"]) > 3:
            rows, cols = O.rows_cols(tuple(c["shape"]), c["layerwise"])
        else:
            rows, cols = O.rows_cols(tuple(c["shape"]), False)
        want = G.arr(c, "idx").reshape(rows, cols)
        for sem in sems(c):
            bins, scales, over = O.export("sym", G.arr(c, "x"), rows, cols, c["bits"], "int16", c["dtype"], sem=sem, autocast=True)
            got = O.unpack_bins(bins, cols, "int16", signed=True)
            fits = (want >= -32768) & (want <= 32767)
            assert (got[fits] == want[fits]).all(), c["name"]
            assert (over == (~fits).sum(axis=1)).all(), c["name"]
            assert bits_equal(scales[:, 0], G.arr(c, "scale"), "fp32"), c["name"]


def test_eager_chain_under_the_policy_matches_reference_fixture():
    """oracle/eager_chain.py is the op chain the GPU tier runs under the live torch.autocast("cuda"); under the emulated policy
    on CPU it must give what the reference's own code gave (same policy, same inputs)"""
    G = golden("autocast.npz")
    for c in G.cases:
        dt = c["dtype"]
        x = t16(G.arr(c, "x"), dt).reshape(c["shape"])
        for dev in {"cpu": (False,), "device": (True,), "both": (False, True)}[c["scalars"]]:
            xr = x.clone().requires_grad_(True)
            with cuda_autocast_policy(TD[dt], dev):
                y = E.EagerSym.apply(xr, torch.from_numpy(G.arr(c, "clip")), c["bits"], c["layerwise"])
            assert y.dtype == torch.float32 and bits_equal(y.detach().numpy(), G.arr(c, "y"), "fp32"), c["name"]
            y.backward(torch.from_numpy(G.arr(c, "g")))
            assert bits_equal(n16(xr.grad), G.arr(c, "gx"), dt), c["name"]


def test_module_operands_match_the_oracle():
    """QuantizeLinear under the policy: the operands handed to the GEMM (reference: fp32 fake-quant results cast by F.linear's
    autocast) == the oracle's narrow results for SymQuantizer operands"""
    G = golden("autocast.npz")
    checked = 0
    for c in json_cases(G):
        dt, adt = c["dtype"], c["autocast_dtype"]
        w, x = G.z[f"{c['name']}/w"], G.z[f"{c['name']}/x"]
        opw, opx = G.z[f"{c['name']}/opw"], G.z[f"{c['name']}/opx"]
        for sem in sems(c):
            if 3 <= c["w_bits"] < 32:
                lw = c.get("weight_layerwise", False)
                rows, cols = O.rows_cols(w.shape, lw)
                y32, _ = O.sym_fwd_autocast(w, rows, cols, c["w_bits"], dt, wide=True, sem=sem)
                want = n16(torch.from_numpy(y32.reshape(w.shape)).to(TD[adt]))   # ONE rounding, to the autocast dtype
                assert bits_equal(opw, want, adt), f"{c['name']} weight operand"
                if dt == adt:
                    yn, _ = O.sym_fwd_autocast(w, rows, cols, c["w_bits"], dt, wide=False, sem=sem)
                    assert bits_equal(opw, yn.reshape(w.shape), adt), f"{c['name']} weight operand (narrow)"
                checked += 1
            if c["symmetric"] and 2 < c["a_bits"] < 32:
                la = c.get("act_layerwise", False)
                rows, cols = O.rows_cols(x.shape, la)
                y32, _ = O.sym_fwd_autocast(x, rows, cols, c["a_bits"], dt, wide=True, sem=sem)
                want = n16(torch.from_numpy(y32.reshape(x.shape)).to(TD[adt]))
                assert bits_equal(opx, want, adt), f"{c['name']} input operand"
                checked += 1
    assert checked >= 30
