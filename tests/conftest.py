import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_sessionstart(session):
    """Build what is missing or stale (the .so files are git-ignored): the HIP library with hipcc
    (cross-compiles for gfx950 without a GPU, ~70 s), the C++ autograd node with g++ and the CPU oracle with gcc.  Failures surface in
    the tests that need the artefact, not here."""
    import importlib.util
    try:
        spec = importlib.util.spec_from_file_location("_fq_build", os.path.join(ROOT, "llm-qat_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build_extension()
    except Exception as e:  # noqa: BLE001
        print(f"[conftest] HIP library build skipped/failed: {e!r}", file=sys.stderr)
    try:
        mod.build_node()        # the C++ autograd node (g++ against this interpreter's PyTorch, ~30 s): before the package is first imported
    except Exception as e:  # noqa: BLE001
        print(f"[conftest] C++ autograd node build skipped/failed: {e!r} (its tests will say so)", file=sys.stderr)
    try:
        from oracle import oracle as O
        O.build()
    except Exception as e:  # noqa: BLE001
        print(f"[conftest] oracle build failed: {e!r}", file=sys.stderr)


@pytest.fixture(scope="session", autouse=True)
def _fixture_semantics():
    """The product's default arithmetic for CUDA tensors is the DEVICE's since round 5 (ops.set_semantics).  Most fixtures of this suite
    were produced by the reference on CPU tensors, and the tests written against them compare kernels with `sem = cpu_eager`; tests of the
    device arithmetic (test_gpu_device_scalars.py, every comparison with the live ATen chain) switch explicitly and restore this base
    state.  So the session starts from cpu_eager -- a statement about which fixture a test compares with, not about the default, which
    test_gpu_device_scalars.py::test_device_eager_is_the_default_for_cuda_tensors checks in a fresh interpreter."""
    try:
        import llm_qat_amd
        llm_qat_amd.set_semantics("cpu_eager")
    except Exception:  # noqa: BLE001 -- tests that need the package fail on their own terms
        pass
    yield


def experiment_module(*rel):
    """A measurement library under tools/ (tools/qlinear, tools/int8_linear: experiments, not the product): import its Python wrapper,
    build / load its .so, and SKIP the calling test -- never fail it -- when that is not possible.  The product's own tests never
    come through here (tests/test_abi_and_host.py checks that the package does not reach into tools/)."""
    import importlib.util
    path = os.path.join(ROOT, *rel)
    try:
        spec = importlib.util.spec_from_file_location("_experiment_" + rel[-1].replace(".py", ""), path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build()
        if hasattr(mod, "lib"):
            mod.lib()
        return mod
    except Exception as e:  # noqa: BLE001
        pytest.skip(f"experiment library {'/'.join(rel[:-1])} is not available ({e!r}): experiments are not part of the product build")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests are skipped (not failed) where no GPU is visible, e.g. in the build container."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden:
    """One tests/golden/*.npz file: arrays + the JSON manifest written by make_golden.py."""

    def __init__(self, fname):
        self.z = np.load(os.path.join(GOLDEN, fname), allow_pickle=False)
        doc = json.loads(bytes(self.z["manifest"]).decode())
        self.meta = doc["meta"]
        self.cases = doc["cases"]

    def arr(self, case, key):
        name = case["name"] if isinstance(case, dict) else case
        return self.z[f"{name}/{key}"]


_cache = {}


def golden(fname):
    if fname not in _cache:
        _cache[fname] = Golden(fname)
    return _cache[fname]


# ---- bit-level helpers shared by the CPU and GPU tests -------------------------------------
def nan_mask(a, dtype):
    if dtype == "fp32":
        return np.isnan(a)
    if dtype == "bf16":
        return (a & 0x7FFF) > 0x7F80
    return (a & 0x7FFF) > 0x7C00


def bits_equal(a, b, dtype):
    """bit-for-bit equal, except that any NaN equals any NaN (payload/sign of NaN is not part of the contract)"""
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    na, nb = nan_mask(a, dtype), nan_mask(b, dtype)
    if a.dtype == np.float32:
        a, b = a.view(np.uint32), b.view(np.uint32)
    return bool((na == nb).all() and ((a == b) | (na & nb)).all())


def mismatch_report(a, b, dtype, limit=5):
    na, nb = nan_mask(a, dtype), nan_mask(b, dtype)
    av, bv = (a.view(np.uint32), b.view(np.uint32)) if a.dtype == np.float32 else (a, b)
    bad = np.argwhere(~(((av == bv) | (na & nb)) & (na == nb)))
    return f"{len(bad)} mismatches, first at {bad[:limit].tolist()}: got {[a[tuple(i)] for i in bad[:limit]]} want {[b[tuple(i)] for i in bad[:limit]]}"


def to_f32(a, dtype):
    """raw storage (uint16 bits / float32) -> float32 values"""
    if dtype == "fp32":
        return a.astype(np.float32)
    if dtype == "bf16":
        return (a.astype(np.uint32) << 16).view(np.float32)
    return a.view(np.float16).astype(np.float32)
