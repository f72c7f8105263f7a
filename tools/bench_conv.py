"""3x3 convolutions of the R-50 trunk at the bench batch (28 frames of 800x1344): MIOpen
(F.conv2d, channels_last, solver search on) vs the hand-written fp32-MFMA implicit GEMM."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pavenet_amd import ops  # noqa: E402

torch.backends.cudnn.benchmark = True


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    shapes = [(n, 200, 336, 64, 64, 1), (n, 200, 336, 128, 128, 2), (n, 100, 168, 128, 128, 1),
              (n, 100, 168, 256, 256, 2), (n, 50, 84, 256, 256, 1), (n, 50, 84, 512, 512, 2),
              (n, 25, 42, 512, 512, 1)]
    for (N, H, W, Cin, Cout, s) in shapes:
        x = torch.randn(N, Cin, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Cout, Cin, 3, 3, device='cuda') / (3 * Cin**0.5)).contiguous(
            memory_format=torch.channels_last)
        b = torch.randn(Cout, device='cuda')
        wt = w.permute(2, 3, 1, 0).contiguous()
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        fl = 2.0 * N * Ho * Wo * Cout * Cin * 9
        t_lib = timeit(lambda: ops.bias_act_rows_(F.conv2d(x, w, None, s, 1), b, None, relu=True))
        wp3 = ops.split_conv3x3_weight(w, 3)
        wp16 = ops.split_conv3x3_weight(w, 16)
        t_s3 = timeit(lambda: ops.conv3x3_split(x, wp3, b, stride=s, relu=True))
        t_s16 = timeit(lambda: ops.conv3x3_split(x, wp16, b, stride=s, relu=True, fp16=True))
        t_own = timeit(lambda: ops.conv3x3_nhwc(x, wt, b, stride=s, relu=True))
        err = (ops.conv3x3_nhwc(x, wt, b, stride=s, relu=True) -
               torch.relu(F.conv2d(x, w, b, s, 1))).abs().max().item()
        print(f'{N}x{H}x{W} {Cin}->{Cout} s{s}: MIOpen+bias/relu pass {t_lib:7.3f} ms ({fl / t_lib / 1e9:6.1f} TF/s)'
              f'   mfma {t_own:7.3f} ms ({fl / t_own / 1e9:6.1f} TF/s)   max|d| {err:.2e}'
              f'   split-bf16x3 {t_s3:7.3f} ms ({fl / t_s3 / 1e9:6.1f} TF/s)   fp16 {t_s16:7.3f} ms')


if __name__ == '__main__':
    main()
