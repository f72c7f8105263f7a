"""3x3 convolutions at one-clip batch sizes (few row tiles): the shipped split-K selection against no split-K
(diag variant 6), one process.   python tools/conv_small_batch.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pavenet_amd import native, ops  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for F, H, W in ((3, 25, 42), (7, 25, 42), (28, 25, 42)):      # the ChannelMapper's extra level: 3x3 / stride 2, 2048 -> 256
    x = torch.randn(F, H, W, 2048, device='cuda').relu_().permute(0, 3, 1, 2)
    wp = ops.split_conv3x3_weight(torch.randn(256, 2048, 3, 3, device='cuda') * 0.02)
    b = torch.randn(256, device='cuda')
    fl = 2 * F * 13 * 21 * 9 * 2048 * 256
    t = {}
    for v in (6, 0, 6, 0):
        with native.diag_build(v):
            t[v] = timed(lambda: ops.conv3x3_split(x, wp, b, stride=2, cout=256))
    print(f'3x3 s2 2048->256 on {F}x{H}x{W}: no split-K {t[6]:7.1f} us ({fl / t[6] * 1e-6:5.1f} TF/s)   shipped {t[0]:7.1f} us '
          f'({fl / t[0] * 1e-6:5.1f} TF/s)', flush=True)
for F, C, H, W in ((3, 256, 50, 84), (3, 512, 25, 42), (6, 256, 50, 84), (3, 128, 100, 168)):
    x = torch.randn(F, H, W, C, device='cuda').relu_().permute(0, 3, 1, 2)
    wp = ops.split_conv3x3_weight(torch.randn(C, C, 3, 3, device='cuda') * 0.05)
    b = torch.randn(C, device='cuda')
    fl = 2 * F * H * W * 9 * C * C
    t = {}
    for v in (6, 0, 6, 0):
        with native.diag_build(v):
            t[v] = timed(lambda: ops.conv3x3_split(x, wp, b, relu=True, cout=C))
    print(f'3x3 {C}->{C} on {F}x{H}x{W}: no split-K {t[6]:7.1f} us ({fl / t[6] * 1e-6:5.1f} TF/s)   shipped {t[0]:7.1f} us '
          f'({fl / t[0] * 1e-6:5.1f} TF/s)', flush=True)

# plain row GEMMs with a split-K plan (a one-clip batch's layer4 1x1 reductions, the neck's C5 lateral)
for M, K, N in ((3150, 2048, 512), (3150, 2048, 256), (7350, 2048, 512), (7350, 2048, 256)):
    a = torch.randn(M, K, device='cuda').relu_()
    wp = ops.split_weight_bf16x3(torch.randn(N, K, device='cuda') * 0.02)
    b = torch.randn(N, device='cuda')
    fl = 2 * M * K * N
    t = {}
    for sk in (False, True, False, True):
        ops.SPLITK_ROWS = sk
        t[sk] = timed(lambda: ops.gemm_bf16x3(a, wp, b, relu=True))
    print(f'{M} x {K} x {N}: one pass {t[False]:7.1f} us ({fl / t[False] * 1e-6:5.1f} TF/s)   split-K {t[True]:7.1f} us '
          f'({fl / t[True] * 1e-6:5.1f} TF/s)', flush=True)
