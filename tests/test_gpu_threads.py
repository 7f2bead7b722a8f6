"""GPU tier: four Python threads, each on its own HIP stream, run forward + backward of their own fake-quantized block (seven QuantizeLinear
layers + the KV hooks, default settings, bf16 autocast) concurrently.  Every thread's outputs and gradients are bit-identical to the same
block run alone, and the host logic's counters are exactly what four independent runs give: what one thread remembers between calls (shared
activations, a pending V) is invalidated by backward passes over ITS graphs only -- the backward runs on the autograd engine's threads, so
each node carries the id of the forward thread that built it.  (Until round 4 any thread's backward invalidated every thread's state:
results stayed valid, but whether siblings shared a node -- and with it the association order of a bf16 gradient sum -- depended on timing.)
"""
import os
import sys
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_gpu_graph_block import Block  # noqa: E402

ITERS, THREADS = 60, 4


def test_concurrent_threads_get_the_results_of_running_alone():
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ
    llm_qat_amd.set_semantics("device_eager")
    llm_qat_amd.reset_learned_state()

    def make(seed):
        torch.manual_seed(seed)
        b = Block(UQ, 128, 352).cuda().bfloat16()
        with torch.no_grad():
            for p in b.parameters():
                p.mul_(0.6)
        x = torch.randn(2, 24, 128, device="cuda").bfloat16().requires_grad_(True)
        go = (torch.randn(2, 24, 128, device="cuda") * 1e-2).bfloat16()
        return b, x, go

    def step(b, x, go):
        b.zero_grad(set_to_none=True)
        x.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = b(x)
        out.backward(go)
        return [out.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in b.parameters()]

    try:
        jobs = [make(s) for s in range(THREADS)]
        want = [step(*j) for j in jobs]
        torch.cuda.synchronize()
        errs, results = [], [None] * THREADS

        def worker(i):
            try:
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.default_stream())
                with torch.cuda.stream(s):
                    for _ in range(ITERS):
                        r = step(*jobs[i])
                    s.synchronize()
                results[i] = r
            except Exception as e:  # noqa: BLE001
                errs.append((i, repr(e)))

        llm_qat_amd.stats(reset=True)
        ts = [threading.Thread(target=worker, args=(i,)) for i in range(THREADS)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        torch.cuda.synchronize()
        assert not errs, errs
        for i in range(THREADS):
            assert all(torch.equal(a, b) for a, b in zip(results[i], want[i])), f"thread {i} differs from the same block run alone"
        st, n = llm_qat_amd.stats(), THREADS * ITERS
        # per step: q / o / gate / down pair with their input (4), k / v / up find it shared (3 singles, 3 hits), K+V one launch, 7 in-place gradients
        assert (st.get("pair_launch"), st.get("single_launch"), st.get("act_share_hit"), st.get("kv_pair_launch"), st.get("kv_pair_hit"),
                st.get("inplace_taken")) == (4 * n, 3 * n, 3 * n, n, n, 7 * n), st
        # (act_share_miss: the four modules per step that quantized an activation no sibling had done before them -- every one of them
        # a pair launch here; none of the three hits was lost to another thread's state)
        assert not st.get("kv_pair_discarded") and st.get("act_share_miss") == 4 * n, st
    finally:
        llm_qat_amd.set_semantics("cpu_eager")
        llm_qat_amd.reset_learned_state()
