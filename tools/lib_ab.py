"""A/B of two builds of the C-ABI library in ONE process, interleaved, on the encoder layer's GEMM
launches at the bench size: pavenet_amd/lib/libpave_hip.so against a second build with the same ABI
(default lib/libpave_hip_prev.so: `git archive <commit> pavenet_amd/csrc include`, hipcc as
build_native.py does, linked into that path).   python tools/lib_ab.py [other.so]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pavenet_amd import native, ops  # noqa: E402

LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21)]
S = sum(h * w for h, w in LEVELS)


def timed(fn, iters=8):
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
    other = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(native.LIB_PATH), 'libpave_hip_prev.so')
    cur = native.load()
    prev = native._open(other)
    M = 28 * S
    dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(0)
    a256 = torch.randn(M, 256, device=dev, generator=g)
    a1024 = torch.randn(M, 1024, device=dev, generator=g)
    idt = torch.randn(M, 256, device=dev, generator=g)
    mk = lambda n, k: ops.split_weight_bf16x3(torch.randn(n, k, device=dev, generator=g) * 0.05)   # noqa: E731
    w_out, w_ffn1, w_ffn2, w_enc, w_v = mk(256, 256), mk(1024, 256), mk(256, 1024), mk(640, 256), mk(512, 256)
    b256, gam, bet = (torch.randn(256, device=dev, generator=g) for _ in range(3))
    b1024 = torch.randn(1024, device=dev, generator=g)
    table = torch.randn(S, 640, device=dev, generator=g) * 0.1
    ref = torch.rand(M, 4, 2, device=dev, generator=g)
    o1, o2 = idt.clone(), idt.clone()
    cases = [
        ('out_proj + identity + LayerNorm, K = 256, in place', lambda: ops.gemm_bf16x3_ln(a256, w_out, b256, o1, gam, bet, 1e-5, out=o1)),
        ('FFN2 + identity + LayerNorm, K = 1024, in place', lambda: ops.gemm_bf16x3_ln(a1024, w_ffn2, b256, o2, gam, bet, 1e-5, out=o2)),
        ('FFN1 256 -> 1024 + ReLU', lambda: ops.gemm_bf16x3(a256, w_ffn1, b1024, relu=True)),
        ('merged projection N = 640 + sampler epilogue', lambda: ops.gemm_bf16x3_encproj(a256, w_enc, table, ref, LEVELS, value_bias=b256)),
        ('value projections N = 512, two outputs', lambda: ops.gemm_bf16x3_ex(a256, w_v, None, None, n_split=256)),
    ]
    # ResNet layer1 Bottleneck as one launch (3x3 -> conv3 + identity -> next conv1), in place on the identity
    c1 = torch.randn(28, 200, 336, 64, device=dev, generator=g).relu_().permute(0, 3, 1, 2)
    idm = torch.randn(28, 200, 336, 256, device=dev, generator=g).relu_().permute(0, 3, 1, 2)
    w2 = ops.split_conv3x3_weight(torch.randn(64, 64, 3, 3, device=dev, generator=g) * 0.05)
    w3, w1n = mk(256, 64), mk(64, 256)
    b64, b64n = (torch.randn(64, device=dev, generator=g) for _ in range(2))
    cases.append(('layer1 Bottleneck chain (3x3 | conv3 + identity | next conv1)',
                  lambda: ops.bottleneck_chain(c1, w2, b64, w3, b256, residual=idm, w1n_planes=w1n, b1n=b64n, out=idm)))
    # ResNet stem: 7x7 / stride 2 on the 28-frame batch
    img = torch.randn(28, 3, 800, 1344, device=dev, generator=g)
    wst = ops.split_stem7x7_weight(torch.randn(64, 3, 7, 7, device=dev, generator=g) * 0.05)
    cases.append(('stem 7x7 / 2, 28 x 800 x 1344', lambda: ops.conv7x7s2_nchw_split(img, wst, b64, relu=True)))
    native._lib = prev
    y_other = ops.conv7x7s2_nchw_split(img, wst, b64, relu=True)
    native._lib = cur
    print('stem: same bits in both builds:', torch.equal(y_other, ops.conv7x7s2_nchw_split(img, wst, b64, relu=True)))
    del y_other
    # HRNet stem conv1: 3x3 / stride 2, 3 -> 64 channels
    taps = torch.randn(27, 64, device=dev, generator=g) * 0.1
    cases.append(('HRNet stem conv1 3x3 / 2, 28 x 800 x 1344', lambda: ops.conv3x3s2_c3_nchw(img, taps, b64, relu=True)))
    native._lib = prev
    y_other = ops.conv3x3s2_c3_nchw(img, taps, b64, relu=True)
    native._lib = cur
    print('HRNet stem conv1: same bits in both builds:', torch.equal(y_other, ops.conv3x3s2_c3_nchw(img, taps, b64, relu=True)))
    del y_other
    # the decoders' small Linears: 50 dependent launches each (1 200 rows), cold weights every launch
    xs = torch.randn(1200, 256, device=dev, generator=g)
    x4 = torch.randn(1200, 1024, device=dev, generator=g)
    wsm = [mk(256, 256) for _ in range(50)]
    wsl = [mk(256, 1024) for _ in range(50)]

    def chain_small():
        y = xs
        for w_ in wsm:
            y = ops.gemm_bf16x3(y, w_)
        return y

    def chain_small_k1024():
        for w_ in wsl:
            ops.gemm_bf16x3(x4, w_)
    gsm, bsm = torch.randn(256, device=dev, generator=g), torch.randn(256, device=dev, generator=g)
    rsm = torch.randn(1200, 256, device=dev, generator=g)

    def chain_small_ln():
        y = xs
        for w_ in wsm:
            y = ops.gemm_bf16x3_ln(y, w_, b256, rsm, gsm, bsm, 1e-5)
        return y

    def chain_small_ln_k1024():
        for w_ in wsl:
            ops.gemm_bf16x3_ln(x4, w_, b256, rsm, gsm, bsm, 1e-5)
    cases.append(('50 dependent small Linear + LayerNorm 1200 x 256 x 256 (us per 50)', chain_small_ln))
    cases.append(('50 small Linear + LayerNorm 1200 x 1024 x 256 (us per 50)', chain_small_ln_k1024))
    cases.append(('50 dependent small launches 1200 x 256 x 256 (us per 50)', chain_small))
    cases.append(('50 small launches 1200 x 1024 x 256 (us per 50)', chain_small_k1024))
    res = {(n, w): [] for n, _ in cases for w in ('this', 'other')}
    for rnd in range(4):
        for name, fn in cases:
            for which, lib in (('this', cur), ('other', prev)):
                native._lib = lib
                try:
                    t = timed(fn)
                finally:
                    native._lib = cur
                if rnd:
                    res[(name, which)].append(t)
    for name, _ in cases:
        a, b = sorted(res[(name, 'this')])[1], sorted(res[(name, 'other')])[1]
        print(f'{name:58s} this {a:8.1f} us   other {b:8.1f} us   {100 * (a / b - 1):+5.1f} %')


if __name__ == '__main__':
    main()
