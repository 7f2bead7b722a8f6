#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 counter passes over the direct-operand GEMM experiment: down_proj [2048,11008].[4096,11008]^T,
hipBLASLt's F.linear and the direct kernel (raw, both MFMA shapes, and the no-MFMA ablation = its load pipeline alone), 12 launches each
on rotating buffers.  tools/qlinear/direct_profile.sh wraps it; summarised by tools/qlinear/direct_profile_summary.py."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import qlinear as QX  # noqa: E402

m, k, n = 2048, 11008, 4096
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(7)
sets = []
for _ in range(3):
    sets.append(((torch.randn(m, k, generator=g, device=dev)).bfloat16(), (torch.randn(n, k, generator=g, device=dev) * 0.02).bfloat16(),
                 torch.empty(m, n, dtype=torch.bfloat16, device=dev)))
st = torch.cuda.current_stream().cuda_stream
L = QX.lib()
for i in range(12):
    x, w, o = sets[i % 3]
    F.linear(x, w)
for mf, abl in ((32, 0), (16, 0), (32, 1)):
    for i in range(12):
        x, w, o = sets[i % 3]
        QX.check(L.fq_qlinear_direct_fwd(x.data_ptr(), w.data_ptr(), None, o.data_ptr(), m, k, n, 64, mf, 0, abl, st), "direct")
torch.cuda.synchronize()
print("done")
