"""Does the split GEMM's speed depend on the DATA?  Times the FFN2 shape (625 044 x 1024 x 256, + residual)
with the same kernel on random, post-ReLU-like (half zeros), all-zero and constant operands.  On a
power-managed part the clock follows the switching activity of the MFMA operands, not the instruction
stream.    python tools/gemm_data_power.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pavenet_amd  # noqa: E402,F401
import torch  # noqa: E402
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
    return s.elapsed_time(e) / iters


def main():
    M, K, N = 625044, 1024, 256
    g = torch.Generator(device='cuda').manual_seed(0)
    w = torch.randn(N, K, device='cuda', generator=g) * 0.05
    r = torch.randn(M, N, device='cuda', generator=g)
    b = torch.randn(N, device='cuda', generator=g)
    a_rand = torch.randn(M, K, device='cuda', generator=g)
    cases = [('A randn, W randn', a_rand, w),
             ('A relu(randn) (half zeros), W randn', torch.relu(a_rand), w),
             ('A = 1.0 (one bf16 plane non-zero), W randn', torch.ones_like(a_rand), w),
             ('A zeros, W randn', torch.zeros_like(a_rand), w),
             ('A randn, W zeros', a_rand, torch.zeros_like(w)),
             ('A zeros, W zeros', torch.zeros_like(a_rand), torch.zeros_like(w))]
    for label, a, ww in cases:
        wp = ops.split_weight_bf16x3(ww)
        ms = timed(lambda: ops.gemm_bf16x3(a, wp, b, r, relu=False))
        print(f'{label:48s} {ms:.3f} ms  {2.0 * M * K * N / ms / 1e9:.0f} TFLOP/s')


if __name__ == '__main__':
    main()
