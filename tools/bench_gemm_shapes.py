"""Times every distinct row-GEMM shape of the T=7 x 4-clip workload (ResNet-50 1x1 convolutions on
the NHWC map, encoder projections / FFN) through the same torch entry points the model uses, and
prints TFLOP/s and the effective HBM rate (read A + write C), to show which are MFMA- and which
are bandwidth-bound.   python tools/bench_gemm_shapes.py [frames=28]"""
import sys

import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pavenet_amd import ops  # noqa: E402


def timed(fn, iters=8):
    for _ in range(2):
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
    dev = 'cuda'
    hw = [(200, 336), (100, 168), (50, 84), (25, 42)]
    shapes = []  # (label, M, K, N, kind)
    cin = 64
    for li, (planes, blocks) in enumerate([(64, 3), (128, 4), (256, 6), (512, 3)]):
        h, w = hw[li]
        hp, wp = hw[max(li - 1, 0)]
        for b in range(blocks):
            m_in = n * (hp * wp if (b == 0) else h * w)
            shapes.append((f'layer{li+1}.{b}.conv1', m_in, cin, planes, 'act'))
            shapes.append((f'layer{li+1}.{b}.conv3', n * h * w, planes, planes * 4,
                           'res' if b else 'add'))
            if b == 0:
                shapes.append((f'layer{li+1}.0.ds', n * h * w, cin, planes * 4, 'add'))
            cin = planes * 4
    S = n * 22323
    shapes += [('enc.value_proj', S, 256, 256, 'add'), ('enc.off+attn', S, 256, 384, 'add'),
               ('enc.out_proj', S, 256, 256, 'res'), ('enc.ffn1', S, 256, 1024, 'act'),
               ('enc.ffn2', S, 1024, 256, 'res')]
    seen = {}
    for label, M, K, N, kind in shapes:
        key = (M, K, N, kind)
        if key in seen:
            seen[key][0].append(label)
            continue
        a = torch.randn(M, K, device=dev)
        wt = torch.randn(N, K, device=dev) * 0.05
        bias = torch.randn(N, device=dev)
        if kind == 'act':
            fn = lambda: torch._addmm_activation(bias, a, wt.t())
        elif kind == 'add':
            fn = lambda: torch.addmm(bias, a, wt.t())
        else:
            c = torch.randn(M, N, device=dev)
            fn = lambda: c.addmm_(a, wt.t())
        ms = timed(fn)
        if N % 128 == 0 and K % 64 == 0:
            wp = ops.split_weight_bf16x3(wt)
            if kind == 'res':
                f3 = lambda: ops.gemm_bf16x3(a, wp, bias, c, relu=False, out=c)
            else:
                f3 = lambda: ops.gemm_bf16x3(a, wp, bias, None, relu=(kind == 'act'))
            ms3 = timed(f3)
            from pavenet_amd import native
            native.use_diag_build(2)
            ms3b = timed(f3)
            ms8 = float('nan')
            if N % 256 == 0:
                native.use_diag_build(3)
                ms8 = timed(f3)
            native.use_diag_build(0)
            print(f'   {label}: library {ms:.3f} ms ({2.0 * M * K * N / ms / 1e9:.0f} TF/s)   '
                  f'bf16x3 split {ms3:.3f} ms ({2.0 * M * K * N / ms3 / 1e9:.0f} TF/s)   '
                  f'256-row tile, 1 wave/SIMD {ms3b:.3f} ms ({2.0 * M * K * N / ms3b / 1e9:.0f} TF/s)   '
                  f'128x256 tile, 8 waves {ms8:.3f} ms ({2.0 * M * K * N / ms8 / 1e9:.0f} TF/s)')
            if N == 256 and kind == 'res':
                gam, bet = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
                sep = timed(lambda: ops.bias_add_layernorm(
                    ops.gemm_bf16x3(a, wp, None, c, relu=False, out=c), bias, None, gam, bet, 1e-5))
                fus = timed(lambda: ops.gemm_bf16x3_ln(a, wp, bias, c, gam, bet, 1e-5, out=c))
                print(f'   {label}: split GEMM + LayerNorm pass {sep:.3f} ms   '
                      f'LayerNorm in the GEMM epilogue {fus:.3f} ms')
        own = float('nan')
        if kind == 'res' and 'conv3' in label:
            wkn = wt.t().contiguous()
            # library path = GEMM + the bias/ReLU pass;  own = one kernel
            ms_lib = timed(lambda: ops.bias_act_rows_(c.addmm_(a, wt.t()), bias, None, relu=True))
            own = timed(lambda: ops.rows_gemm_bias_res_act(a, wkn, bias, c, relu=True, out=c))
            ab = torch.randn(K, device=dev)
            own_ab = timed(lambda: ops.rows_gemm_bias_res_act(a, wkn, bias, c, relu=True, out=c,
                                                              a_bias=ab))
            print(f'   {label}: hipBLASLt addmm_ + bias/relu pass {ms_lib:.3f} ms   '
                  f'fused MFMA kernel {own:.3f} ms   with bn2+relu on A load {own_ab:.3f} ms')
        seen[key] = ([label], ms)
        del a, wt
    tot = 0.0
    for (M, K, N, kind), (labels, ms) in seen.items():
        fl = 2.0 * M * K * N
        by = 4.0 * (M * K + M * N * (2 if kind == 'res' else 1))
        tot += ms * len(labels)
        print(f'{labels[0]:18s} x{len(labels)} M={M:8d} K={K:4d} N={N:4d} {kind:3s} {ms:7.3f} ms '
              f'{fl / ms / 1e9:6.1f} TF/s {by / ms / 1e6:7.0f} GB/s')
    print(f'total {tot:.1f} ms/step')


if __name__ == '__main__':
    main()
