"""HRNet-w48's 3x3 convolutions on the fp32-input form (operand split per (pixel, tap) on the VALU) against the
pre-split-planes form (pave_conv3x3_planes_f32: the map as [pixel][3][Cin] bf16, no vector arithmetic in the loop) at
the sizes of a T = 7 x 4-clip step; us per launch behind a busy stream, and what writing the planes costs as a pass of
its own (a producer epilogue would write them instead).   python tools/conv3x3_planes_ab.py [frames=28]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pavenet_amd import ops  # noqa: E402

_BLOCK = None


def timed(fn, iters=12):
    global _BLOCK
    if _BLOCK is None:
        _BLOCK = (torch.randn(8192, 8192, device='cuda'), torch.randn(8192, 8192, device='cuda'))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _BLOCK[0] @ _BLOCK[1]
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


F = int(sys.argv[1]) if len(sys.argv) > 1 else 28
for Cin, Cout, H, W in ((48, 48, 200, 336), (96, 96, 100, 168), (64, 64, 200, 336), (192, 192, 50, 84)):
    x = torch.randn(F, Cin, H, W, device='cuda').relu_().contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, 3, 3, device='cuda') / (3 * Cin ** 0.5)
    b = torch.randn(Cout, device='cuda')
    res = torch.randn(F, Cout, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    wp = ops.split_conv3x3_weight(w, 3)
    rows = x.permute(0, 2, 3, 1)
    planes = ops.split_rows_bf16x3(rows)
    ref = ops.conv3x3_split(x, wp, b, relu=True, residual=res, cout=Cout)
    got = ops.conv3x3_planes(planes, wp, b, relu=True, residual=res, cout=Cout)
    same = bool(torch.equal(ref, got))
    t_f32 = timed(lambda: ops.conv3x3_split(x, wp, b, relu=True, residual=res, cout=Cout))
    t_pl = timed(lambda: ops.conv3x3_planes(planes, wp, b, relu=True, residual=res, cout=Cout))
    t_sp = timed(lambda: ops.split_rows_bf16x3(rows))
    fl = 2 * F * H * W * Cout * 9 * Cin
    print(f'{F} x {H} x {W}, {Cin} -> {Cout} + identity + ReLU: fp32 map {t_f32:7.1f} us ({fl / t_f32 * 1e-6:5.1f} TF/s)   '
          f'pre-split planes {t_pl:7.1f} us ({fl / t_pl * 1e-6:5.1f} TF/s)   split pass alone {t_sp:6.1f} us   '
          f'bit-identical: {same}', flush=True)
