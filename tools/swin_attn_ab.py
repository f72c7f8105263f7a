"""Swin window attention core alone at Swin-L's four stage shapes (3 frames of 800x1344): the shipped fp32-MFMA form
against the per-lane LDS-broadcast form (diag variant 18), one process.   python tools/swin_attn_ab.py"""
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


for (H, W, heads) in ((200, 336, 6), (100, 168, 12), (50, 84, 24), (25, 42, 48)):
    C = 32 * heads
    qkv = torch.randn(3, H, W, 3 * C, device='cuda')
    bt = torch.randn(heads, 49, 49, device='cuda') * 0.5
    pad = torch.randn(3 * C, device='cuda') * 0.3
    for shift in (0, 3):
        t = {}
        for v in (18, 0, 18, 0):
            with native.diag_build(v):
                t[v] = timed(lambda: ops.swin_window_attn(qkv, bt, pad, heads, 7, shift, 32 ** -0.5))
        print(f'{H:4d} x {W:4d} x {heads:2d} heads, shift {shift}: per-lane form {t[18]:7.1f} us   MFMA form {t[0]:7.1f} us', flush=True)
