"""Small-row form (a wave per 32 x 32 tile, no LDS: the shipped selection below 8 192 rows and N <= 512) against the
tile kernels (any diag variant keeps them) on mid-sized shapes -- where does the row threshold belong?
python tools/small_vs_tile.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pavenet_amd import native, ops  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for M, K, N in ((300, 256, 256), (1200, 256, 256), (1200, 1024, 256), (1200, 512, 512), (3150, 2048, 512),
                (3150, 1024, 512), (3150, 512, 512), (3150, 256, 256), (3150, 1536, 256), (4000, 1024, 256), (4200, 1024, 256), (6000, 256, 256), (6000, 1024, 256),
                (6000, 2048, 512), (2100, 2048, 512), (4200, 2048, 512), (8000, 2048, 512)):
    a = torch.randn(M, K, device='cuda').relu_()
    wp = ops.split_weight_bf16x3(torch.randn(N, K, device='cuda') * 0.05)
    b = torch.randn(N, device='cuda')
    fn = lambda: ops.gemm_bf16x3(a, wp, b, relu=True)   # noqa: E731
    res = {}
    for v in (0, 17):          # 17: no half-tail form (irrelevant here) -- only "not the shipped selection"
        with native.diag_build(v):
            res[v] = timed(fn)
    fl = 2 * M * K * N
    print(f'{M:6d} x {K:5d} x {N:4d}: small-row form {res[0]:7.1f} us ({fl / res[0] * 1e-6:5.1f} TF/s)   '
          f'tile kernels {res[17]:7.1f} us ({fl / res[17] * 1e-6:5.1f} TF/s)', flush=True)
