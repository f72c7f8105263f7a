"""The decoders' / heads' few-hundred-row Linears on three forms of the split GEMM: the K-split small-row form
(round 6, the shipped selection up to 4 096 tiles of 32 x 32: the block's four waves walk a quarter of K each),
the one-wave small-row form (diag variant 19: the selection of rounds 4 - 5, N <= 512 only) and the 128-row tile
kernels (variant 8).  Back-to-back launches of ONE shape (us per launch): the time a dependent launch on the
decoder tail takes once the stream has reached it.   python tools/small_gemm_ksplit_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pavenet_amd import native, ops  # noqa: E402


_BLOCK = None


def timed(fn, iters=40):
    """us per launch as the DEVICE sees a dependent chain of them: a ~3 ms blocker GEMM goes first, so the host has
    queued all `iters` launches before the stream reaches the first one (a bare loop measures the ~14 us the host
    needs per launch through the Python wrapper, not the kernels)."""
    global _BLOCK
    if _BLOCK is None:
        _BLOCK = (torch.randn(8192, 8192, device='cuda'), torch.randn(8192, 8192, device='cuda'))
    for _ in range(5):
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


SHAPES = [  # (M, K, N, what)
    (300, 256, 256, 'out_proj, one clip'), (300, 1024, 256, 'FFN2, one clip'), (300, 256, 1024, 'FFN1, one clip'),
    (300, 512, 1536, 'branch MLP layer 2 as one GEMM, T=3'), (300, 512, 512, 'a 512-wide MLP layer, one clip'),
    (3150, 2048, 512, 'layer4 conv1, one clip T=3'), (3150, 1024, 256, 'C4 lateral, one clip'),
    (300, 256, 768, 'q|k|v, one clip'), (300, 256, 4352, 'pose proj T=3 (padded to 128)'),
    (300, 256, 1536, 'branch MLP layer 1, T=3'), (300, 256, 1152, 'joint proj T=3'),
    (1200, 256, 256, 'out_proj, 4 clips'), (1200, 1024, 256, 'FFN2, 4 clips'), (1200, 256, 1024, 'FFN1, 4 clips'),
    (1200, 256, 768, 'q|k|v, 4 clips'), (1200, 256, 10112, 'pose proj T=7 (padded to 128)'),
    (1200, 256, 3584, 'branch MLP layer 1, T=7'), (1200, 256, 2688, 'joint proj T=7'),
    (1200, 512, 512, 'a 512-wide MLP layer'), (4500, 256, 256, '20 poses x 15 joints x 15 frames'),
]
for M, K, N, what in SHAPES:
    a = torch.randn(M, K, device='cuda')
    wp = ops.split_weight_bf16x3(torch.randn(N, K, device='cuda') * 0.05)
    b = torch.randn(N, device='cuda')
    fn = lambda: ops.gemm_bf16x3(a, wp, b, relu=True)   # noqa: E731
    res = {}
    for v in (0, 19, 8):
        with native.diag_build(v):
            res[v] = timed(fn)
            if v == 0:
                y0 = fn().clone()
            elif v == 8:
                y8 = fn().clone()
    err = float((y0 - y8).abs().max() / (y8.abs().max() + 1e-9))
    print(f'{M:5d} x {K:4d} x {N:5d}  {what:34s} K-split {res[0]:6.1f} us   one-wave / tile rule of round 5 {res[19]:6.1f} us'
          f'   tile kernels {res[8]:6.1f} us   (K-split vs tile: {err:.1e} relative)', flush=True)

