"""A/B of the LayerNorm-epilogue GEMM forms (8 waves owning a 128 x 256 block | the wide form with
LayerNorm on the accumulator layout; diag variants 13 / 14 force one or the other), interleaved in one process, on the two shapes
of the encoder layer (out_proj: K = 256, FFN2: K = 1024; M = 625 044).   python tools/ln_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pavenet_amd import native, ops  # noqa: E402


def timed(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    M = 625044
    for K in (256, 1024):
        a = torch.randn(M, K, device='cuda')
        r = torch.randn(M, 256, device='cuda')
        w = torch.randn(256, K, device='cuda') * 0.05
        b, g, be = (torch.randn(256, device='cuda') for _ in range(3))
        wp = ops.split_weight_bf16x3(w)
        fn = lambda: ops.gemm_bf16x3_ln(a, wp, b, r, g, be, 1e-5)   # noqa: E731
        for _ in range(3):
            fn()
        res = {'8-wave': [], 'wide': []}
        for rep in range(4):
            with native.diag_build(13):
                res['8-wave'].append(timed(fn))
            with native.diag_build(14):
                res['wide'].append(timed(fn))
        for k, v in res.items():
            print(f'K={K:5d} {k:7s} ' + ' '.join(f'{x:8.1f}' for x in v) + f'   min {min(v):8.1f} us')
        # the shipped build's own choice, out of place and in place (identity == out: the encoder layer's call)
        r2 = r.clone()
        inplace = lambda: ops.gemm_bf16x3_ln(a, wp, b, r2, g, be, 1e-5, out=r2)   # noqa: E731
        t_out = [timed(fn) for _ in range(3)]
        t_in = [timed(inplace) for _ in range(3)]
        print(f'K={K:5d} shipped, out of place ' + ' '.join(f'{x:8.1f}' for x in t_out))
        print(f'K={K:5d} shipped, in place     ' + ' '.join(f'{x:8.1f}' for x in t_in))


if __name__ == '__main__':
    main()
