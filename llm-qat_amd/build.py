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
SOURCES = [os.path.join(CSRC, "fq_api.hip")]
DEPS = SOURCES + [os.path.join(CSRC, "fq_kernels.h"), os.path.join(CSRC, "fq_device.h"),
                  os.path.join(os.path.dirname(HERE), "include", "llmqat_fakequant.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
         "-fPIC", "-shared", "-Wall", "-Wno-unused-variable", "-Wno-unused-but-set-variable"]


def up_to_date():
    return os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in DEPS)


def build_extension(force=False, verbose=False):
    if not force and up_to_date():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc] + FLAGS + ["-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_extension(force="--force" in sys.argv, verbose=True))
