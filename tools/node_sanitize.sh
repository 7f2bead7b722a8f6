#!/bin/bash
# The C++ autograd node (llm-qat_amd/csrc/fq_autograd_node.cpp) under AddressSanitizer + UBSan, on the CPU (GPU sanitizers are not available on
# this pool): builds build_tmp/san/_fq_node.so, points the loader at it (LLMQAT_AMD_NODE) and runs the CPU tests that reach the node --
# its calibration probes at import, the epoch cells, the probe node's forward / backward.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
T=$(python -c "import torch,os;print(os.path.dirname(torch.__file__))")
mkdir -p build_tmp/san
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 \
    -DTORCH_EXTENSION_NAME=_fq_node -DTORCH_API_INCLUDE_EXTENSION_H -D_GLIBCXX_USE_CXX11_ABI=1 -I"$T/include" -I"$T/include/torch/csrc/api/include" \
    -I/opt/rocm/include -I"$(python -c "import sysconfig;print(sysconfig.get_paths()['include'])")" llm-qat_amd/csrc/fq_autograd_node.cpp \
    -o build_tmp/san/_fq_node.so -L"$T/lib" -lc10 -lc10_hip -ltorch_cpu -ltorch -ltorch_python -Wl,-rpath,"$T/lib"
python -c "import torch; open('build_tmp/san/_fq_node.so.built_for','w').write(torch.__version__+'\n')"
LLMQAT_AMD_NODE="$ROOT/build_tmp/san/_fq_node.so" LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
    ASAN_OPTIONS=detect_leaks=0:detect_odr_violation=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    python -m pytest tests/test_cpp_node_cpu.py tests/test_host_logic_cpu.py -x -q -p no:cacheprovider
