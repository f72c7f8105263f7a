"""Runs the split GEMM alone at one shape (for rocprofv3 --pmc passes: tools/pmc_gemm.sh).
python tools/gemm_one.py M K N [iters=6] [variant=0]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pavenet_amd import native, ops  # noqa: E402

M, K, N = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 6
variant = int(sys.argv[5]) if len(sys.argv) > 5 else 0
a = torch.randn(M, K, device='cuda')
w = torch.randn(N, K, device='cuda') * 0.05
wp = ops.split_weight_bf16x3(w)
native.use_diag_build(variant) if variant else native.load()
for _ in range(iters):
    out = ops.gemm_bf16x3(a, wp)
torch.cuda.synchronize()
print('ok', float(out[0, 0]))
