"""Where the out_proj + LayerNorm launch (625 044 x 256 x 256) spends its time: the LayerNorm-epilogue
GEMM against the same product without LayerNorm, without the identity, and against pure copies of
the same bytes.   python tools/ln_shape_probe.py [M=625044]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pavenet_amd import ops  # noqa: E402


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


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 625044
    dev = 'cuda'
    for K in (256, 1024):
        a = torch.randn(M, K, device=dev)
        r = torch.randn(M, 256, device=dev)
        w = torch.randn(256, K, device=dev) * 0.05
        b, g, be = (torch.randn(256, device=dev) for _ in range(3))
        wp = ops.split_weight_bf16x3(w)
        flop = 2 * M * K * 256
        rows = [('gemm + bias + identity + LayerNorm (one launch)', lambda: ops.gemm_bf16x3_ln(a, wp, b, r, g, be, 1e-5)),
                ('gemm + bias + LayerNorm, no identity', lambda: ops.gemm_bf16x3_ln(a, wp, b, None, g, be, 1e-5)),
                ('gemm + bias + identity (no LayerNorm)', lambda: ops.gemm_bf16x3(a, wp, b, r)),
                ('gemm + bias', lambda: ops.gemm_bf16x3(a, wp, b, None)),
                ('copy of the identity-sized matrix (read + write 640 MB each)', lambda: r.clone())]
        for name, fn in rows:
            us = timed(fn)
            print(f'K={K:5d} {name:62s} {us:8.1f} us  {flop / us / 1e6:7.1f} TFLOP/s')


if __name__ == '__main__':
    main()
