"""CPU tier: the C++ autograd node's host mechanics (llm-qat_amd/csrc/fq_autograd_node.cpp): the extension builds against this
interpreter's PyTorch and loads without a GPU, its in-place guard calibrates, the per-thread epoch cells do what utils_quant relies on,
and the switches say what they do.  What the node computes is GPU tier (tests/test_gpu_cpp_node.py)."""
import os
import subprocess
import sys
import threading

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_node_is_built_loaded_and_calibrated():
    import llm_qat_amd
    from llm_qat_amd import _lib, _node
    from llm_qat_amd import utils_quant as U
    assert os.path.exists(_node.NODE_PATH), "python llm-qat_amd/build.py builds it (g++, host code only)"
    assert llm_qat_amd.host_node() == "c++", llm_qat_amd.host_node()
    assert U._cnode.abi_version == _lib.ABI_VERSION
    base = U._cnode.baselines()
    # a gradient nobody else holds, as this node's backward sees it: (TensorImpl holders, StorageImpl holders, holders of the view's base)
    assert set(base) == {"plain", "view"} and base["view"][2] >= 1 and base["plain"][2] == 0, base
    assert base["view"][1] == base["plain"][1] + 1, base      # a view of a temporary: the base holds the storage too


def test_switches_report_themselves():
    import llm_qat_amd
    try:
        assert llm_qat_amd.cpp_node(False) is False and llm_qat_amd.host_node() == "python (switched off)"
        assert llm_qat_amd.cpp_node(True) is True and llm_qat_amd.host_node() == "c++"
    finally:
        llm_qat_amd.cpp_node(True)
    out = subprocess.run([sys.executable, "-c", "import llm_qat_amd; print(llm_qat_amd.host_node())"], cwd=ROOT, capture_output=True, text=True,
                         env=dict(os.environ, LLMQAT_AMD_CPP_NODE="0"), timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "python (switched off (LLMQAT_AMD_CPP_NODE=0))", (out.stdout, out.stderr[-500:])


def test_a_bumped_epoch_cell_makes_the_forward_thread_forget():
    """what the node's backward does on the engine's thread is one atomic increment of its forward thread's cell; the thread's next
    `_state()` notices: everything remembered goes, and the K/V state word changes (st.epoch)"""
    from llm_qat_amd import utils_quant as U
    st = U._state()
    assert st.cell and st.cepoch is not None
    st.acts = {"k": "something remembered"}
    e0 = st.epoch
    assert U._state() is st and st.acts          # nothing happened: nothing forgotten
    st.cepoch.value += 1                         # (the C++ side: fetch_add on the same address)
    assert U._state() is st and not st.acts and st.epoch == e0 + 1
    assert U._state().epoch == e0 + 1            # once per bump


def test_cells_are_recycled_with_their_threads():
    from llm_qat_amd import utils_quant as U
    seen = []

    def work():
        seen.append(U._state().cell)

    for _ in range(3):
        t = threading.Thread(target=work)
        t.start()
        t.join()
    assert len(seen) == 3 and len(set(seen)) == 1, seen     # a dead thread's cell went back to the pool and served the next thread


def test_probe_node_passes_gradients_through():
    """the calibration node itself: an identity over both operands whose backward only looks at reference counts"""
    from llm_qat_amd import utils_quant as U
    before = U._cnode.baselines()
    w = torch.randn(3, 4, requires_grad=True)
    x = torch.randn(5, 4, requires_grad=True)
    wq, xq = U._cnode.probe_node(w, x)
    assert "FqPairNode" in wq.grad_fn.name() and wq.grad_fn is xq.grad_fn
    torch.nn.functional.linear(xq, wq).square().sum().backward()
    wr, xr = w.detach().clone().requires_grad_(True), x.detach().clone().requires_grad_(True)
    torch.nn.functional.linear(xr, wr).square().sum().backward()
    assert torch.equal(w.grad, wr.grad) and torch.equal(x.grad, xr.grad)
    assert U._cnode.baselines() == before        # (an unarmed probe records nothing)


def test_a_pending_v_result_raises_the_cells_second_word():
    """the one thing a C++ node's backward takes the GIL for: a V result of the K/V speculation that is still pending when a backward
    starts.  The thread raises the flag while one is pending; `_forget_from_cpp(cell)` -- what the node calls -- drops it and counts it."""
    import llm_qat_amd
    from llm_qat_amd import utils_quant as U
    st = U._state()
    assert st.cpending is not None and st.cpending.value == 0
    llm_qat_amd.stats(reset=True)
    st.kv = (None, 0, None, None, ("a call signature",))
    assert st.cpending.value == 1
    st.acts = {"k": "remembered"}
    U._forget_from_cpp(st.cell)
    assert st.kv is None and st.cpending.value == 0 and not st.acts
    assert llm_qat_amd.stats().get("kv_pair_discarded") == 1
    assert U._state() is st and st.cseen == st.cepoch.value      # nothing left for the lazy path to do
    llm_qat_amd.reset_learned_state()
    U._forget_from_cpp(12345)                                    # a cell no live thread state owns: ignored
