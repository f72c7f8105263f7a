"""Encoder-layer GEMM chain (out_proj+LN -> FFN1 -> FFN2+LN -> merged projections) at the bench
batch: fp32 hand-over vs bf16-plane hand-over.   python tools/bench_planes.py [frames=28]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pavenet_amd import ops  # noqa: E402


def timed(fn, iters=6):
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
    M, C, H = n * 22323, 256, 1024
    dev = 'cuda'
    attn, idt = torch.randn(M, C, device=dev), torch.randn(M, C, device=dev)
    wo, w1, w2, wm = [torch.randn(a, b, device=dev) * 0.05 for a, b in ((C, C), (H, C), (C, H), (640, C))]
    po, p1, p2, pm = [ops.split_weight_bf16x3(w) for w in (wo, w1, w2, wm)]
    bo, b1, b2 = [torch.randn(k, device=dev) for k in (C, H, C)]
    ga, be = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    table = torch.randn(22323, 640, device=dev)
    ln = (ga, be, 1e-5)
    x1 = ops.gemm_bf16x3_ln(attn, po, bo, idt, ga, be, 1e-5)
    h = ops.gemm_bf16x3(x1, p1, b1, relu=True)
    x2 = ops.gemm_bf16x3_ln(h, p2, b2, x1, ga, be, 1e-5)
    _, x1p = ops.gemm_bf16x3_planes(po, a=attn, bias=bo, residual=idt, ln=ln, want_planes=True)
    _, hp = ops.gemm_bf16x3_planes(p1, a_planes=x1p, bias=b1, relu=True, want_fp32=False, want_planes=True)
    _, x2p = ops.gemm_bf16x3_planes(p2, a_planes=hp, bias=b2, residual=x1, ln=ln, want_planes=True)
    rows = [
        ('out_proj + LN', lambda: ops.gemm_bf16x3_ln(attn, po, bo, idt, ga, be, 1e-5),
         lambda: ops.gemm_bf16x3_planes(po, a=attn, bias=bo, residual=idt, ln=ln, want_planes=True)),
        ('FFN1 + ReLU', lambda: ops.gemm_bf16x3(x1, p1, b1, relu=True),
         lambda: ops.gemm_bf16x3_planes(p1, a_planes=x1p, bias=b1, relu=True, want_fp32=False, want_planes=True)),
        ('FFN2 + LN', lambda: ops.gemm_bf16x3_ln(h, p2, b2, x1, ga, be, 1e-5),
         lambda: ops.gemm_bf16x3_planes(p2, a_planes=hp, bias=b2, residual=x1, ln=ln, want_planes=True)),
        ('value|offsets|logits', lambda: ops.gemm_bf16x3_ex(x2, pm, None, table, residual_rows=22323, n_split=256),
         lambda: ops.gemm_bf16x3_ex(x2p, pm, None, table, residual_rows=22323, n_split=256)),
    ]
    tot = [0.0, 0.0]
    for name, f32, fpl in rows:
        a, b = timed(f32), timed(fpl)
        tot[0] += a
        tot[1] += b
        print(f'{name:24s} fp32 hand-over {a:6.3f} ms   plane hand-over {b:6.3f} ms')
    print(f'{"layer GEMM chain":24s} fp32 hand-over {tot[0]:6.3f} ms   plane hand-over {tot[1]:6.3f} ms')
    mixed = timed(lambda: ops.gemm_bf16x3_planes(p1, a=x1, bias=b1, relu=True, want_fp32=False, want_planes=True))
    print(f'FFN1 fp32 in -> planes out {mixed:6.3f} ms;  FFN2+LN planes in, fp32 out only '
          f'{timed(lambda: ops.gemm_bf16x3_planes(p2, a_planes=hp, bias=b2, residual=x1, ln=ln)):6.3f} ms')


if __name__ == '__main__':
    main()
