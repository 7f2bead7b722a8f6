"""Build the gfx950 shared library (`libllmqat_fakequant.so`) in-tree with hipcc.

    python llm-qat_amd/build.py [--force]

hipcc cross-compiles for gfx950 without a GPU.  Flags that matter for parity:
  -ffp-contract=off   no mul+add fusion: every reference op rounds once
  (no -ffast-math; HIP's default correctly-rounded fp32 divide is kept)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libllmqat_fakequant.so")
SOURCES = [os.path.join(CSRC, f) for f in ("fq_api.hip", "fq_bf16.hip", "fq_f32.hip", "fq_f16.hip", "fq_export.hip", "fq_f64.hip")]
DEPS = SOURCES + [os.path.join(CSRC, f) for f in ("fq_kernels.h", "fq_device.h", "fq_launch.h", "fq_dtype_impl.h", "fq_export.h")] + [
    os.path.join(os.path.dirname(HERE), "include", "llmqat_fakequant.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden",
         "-fPIC", "-Wall", "-Wno-unused-variable", "-Wno-unused-but-set-variable"]


def up_to_date():
    return os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in DEPS)


def build_extension(force=False, verbose=False, extra_flags=(), out=None):
    """extra_flags / out: A/B builds of kernel variants for tools/ab_bench.sh (e.g. -DSOME_VARIANT=1 --out=build_tmp/lib_b.so);
    the product build takes neither."""
    if out is None and not force and up_to_date():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    objdir = os.path.join(HERE, "build" if out is None else "build_" + os.path.basename(out))
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in SOURCES:  # one translation unit per element type: compile them side by side
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [hipcc] + FLAGS + list(extra_flags) + ["-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), obj, cmd))
    objs = []
    for p, obj, cmd in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
        objs.append(obj)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out or LIB] + objs
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    return out or LIB


# ---- the C++ autograd node (csrc/fq_autograd_node.cpp -> _fq_node.so): a PyTorch extension, host code only -------------------------------
NODE_SRC = os.path.join(CSRC, "fq_autograd_node.cpp")
NODE_LIB = os.path.join(HERE, "_fq_node.so")
NODE_DEPS = [NODE_SRC, os.path.join(os.path.dirname(HERE), "include", "llmqat_fakequant.h")]


def node_up_to_date():
    if not (os.path.exists(NODE_LIB) and os.path.exists(NODE_LIB + ".built_for")):
        return False
    import torch
    return open(NODE_LIB + ".built_for").read().strip() == torch.__version__ and all(os.path.getmtime(NODE_LIB) >= os.path.getmtime(d) for d in NODE_DEPS)


def build_node(force=False, verbose=False):
    """g++ against the PyTorch of this interpreter (its headers, its C++ ABI flag) and the HIP runtime headers (for the current stream);
    linked against libtorch / libc10_hip only -- the kernels are reached through function pointers of the C ABI at run time."""
    if not force and node_up_to_date():
        return NODE_LIB
    import sysconfig
    import torch
    tdir = os.path.dirname(torch.__file__)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_fq_node", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
           "-I" + os.path.join(tdir, "include"), "-I" + os.path.join(tdir, "include", "torch", "csrc", "api", "include"),
           "-I" + os.path.join(rocm, "include"), "-I" + sysconfig.get_paths()["include"], NODE_SRC, "-o", NODE_LIB,
           "-L" + os.path.join(tdir, "lib"), "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch", "-ltorch_python", "-Wl,-rpath," + os.path.join(tdir, "lib")]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(NODE_LIB + ".built_for", "w") as f:     # the loader refuses the file under another PyTorch instead of dlopen()ing it and hoping
        f.write(torch.__version__ + "\n")
    return NODE_LIB


if __name__ == "__main__":
    extra = [a for a in sys.argv[1:] if a.startswith("-D")]
    outs = [a[len("--out="):] for a in sys.argv[1:] if a.startswith("--out=")]
    print(build_extension(force="--force" in sys.argv, verbose=True, extra_flags=extra, out=outs[0] if outs else None))
    if not outs:
        print(build_node(force="--force" in sys.argv, verbose=True))
