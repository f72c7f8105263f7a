"""What the encoder layer's merged N = 640 projection pays for beside its MFMA work: the same
625 044 x 256 activation through the plain forms (N = 128 ... 1024), with the row-periodic
table, with two outputs, and through the merged kernel with the sampler arithmetic in its epilogue.
Interleaved rounds in one process; us per launch and fp32-equivalent TFLOP/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pavenet_amd import ops  # noqa: E402

LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21)]
S = sum(h * w for h, w in LEVELS)


def main():
    dev = 'cuda'
    F = 28
    M, K = F * S, 256
    g = torch.Generator(device=dev).manual_seed(0)
    a = torch.randn(M, K, device=dev, generator=g)
    ref = torch.rand(M, 4, 2, device=dev, generator=g)
    planes = {}
    bias = {}
    for N in (128, 256, 384, 512, 640, 1024):
        w = torch.randn(N, K, device=dev, generator=g) / 16
        planes[N] = ops.split_weight_bf16x3(w)
        bias[N] = torch.randn(N, device=dev, generator=g)
    table = {N: torch.randn(S, N, device=dev, generator=g) for N in (384, 512, 640)}
    full = {N: torch.randn(M, N, device=dev, generator=g) for N in (256,)}
    cases = []
    for N in (128, 256, 384, 512, 640, 1024):
        cases.append((f'plain N={N} (+bias)', N, lambda N=N: ops.gemm_bf16x3(a, planes[N], bias[N])))
    cases.append(('N=640 + periodic table', 640,
                  lambda: ops.gemm_bf16x3_ex(a, planes[640], None, table[640], residual_rows=S)))
    cases.append(('N=640 + periodic table, two outputs 256 | 384', 640,
                  lambda: ops.gemm_bf16x3_ex(a, planes[640], None, table[640], residual_rows=S, n_split=256)))
    cases.append(('N=640 merged encoder projection (table, two outputs, sampler epilogue)', 640,
                  lambda: ops.gemm_bf16x3_encproj(a, planes[640], table[640], ref, LEVELS)))
    cases.append(('N=512 + periodic table, two outputs 256 | 256', 512,
                  lambda: ops.gemm_bf16x3_ex(a, planes[512], None, table[512], residual_rows=S, n_split=256)))
    cases.append(('N=384 + periodic table', 384,
                  lambda: ops.gemm_bf16x3_ex(a, planes[384], None, table[384], residual_rows=S)))
    cases.append(('N=256 + full residual', 256, lambda: ops.gemm_bf16x3(a, planes[256], bias[256], full[256])))
    times = {c[0]: [] for c in cases}
    for rnd in range(4):
        for name, N, fn in cases:
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(8):
                fn()
            e.record()
            torch.cuda.synchronize()
            if rnd:
                times[name].append(s.elapsed_time(e) * 1e3 / 8)
    for name, N, fn in cases:
        us = sorted(times[name])[len(times[name]) // 2]
        print(f'{name:78s} {us:8.1f} us  {2.0 * M * K * N / us / 1e6:6.1f} TFLOP/s   per 128 columns {us * 128 / N:6.1f} us')


if __name__ == '__main__':
    main()
