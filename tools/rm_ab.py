"""A/B of the two-row-tiles-per-wave forms (diag variant 16 = wherever the form exists) against one row tile per
wave (15 = never) in ONE process, interleaved, on the launches the form was built for: HRNet-w48's 48- and
96-channel 3x3 convolutions at the bench batch (28 frames of 800 x 1344) and ResNet layer1's Bottleneck launches.
python tools/rm_ab.py [frames=28]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pavenet_amd import native, ops  # noqa: E402


def timed(fn, iters=6):
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
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s, sc=1.0: torch.randn(*s, device=dev, generator=g) * sc   # noqa: E731
    cl = lambda n, h, w, c: rnd(n, h, w, c).relu_().permute(0, 3, 1, 2)      # noqa: E731
    mk = lambda n, k: ops.split_weight_bf16x3(rnd(n, k, sc=0.05), pad=True)   # noqa: E731
    cases = []
    for C, H, W in ((48, 200, 336), (96, 100, 168), (64, 200, 336)):
        x, idn = cl(F, H, W, C), cl(F, H, W, C)
        wp = ops.split_conv3x3_weight(rnd(C, C, 3, 3, sc=0.05))
        b = rnd(C)
        cases.append((f'3x3 {C} -> {C} + identity + ReLU, {F} x {H} x {W}',
                      lambda x=x, wp=wp, b=b, idn=idn, C=C: ops.conv3x3_split(x, wp, b, relu=True, residual=idn, cout=C),
                      2 * F * H * W * 9 * C * C))
    M = F * 200 * 336
    a256 = rnd(M, 256).relu_()
    w1, b64 = mk(64, 256), rnd(64)
    cases.append((f'conv1 256 -> 64 + ReLU, {M} rows', lambda: ops.gemm_bf16x3(a256, w1, b64, relu=True), 2 * M * 256 * 64))
    c1 = cl(F, 200, 336, 64)
    idm = cl(F, 200, 336, 256)
    x64 = cl(F, 200, 336, 64)
    w2 = ops.split_conv3x3_weight(rnd(64, 64, 3, 3, sc=0.05))
    w3, w3d, w1n, w1n128 = mk(256, 64), mk(256, 128), mk(64, 256), mk(128, 256)
    b256, b128 = rnd(256), rnd(128)
    fl = lambda k3, cn: 2 * M * (576 * 64 + k3 * 256 + 256 * cn)   # noqa: E731
    cases.append(('layer1.0 chain (3x3 | conv3 + downsample | next conv1)',
                  lambda: ops.bottleneck_chain(c1, w2, b64, w3d, b256, a2=x64, w1n_planes=w1n, b1n=b64), fl(128, 64)))
    cases.append(('layer1.1 chain (3x3 | conv3 + identity | next conv1), in place',
                  lambda: ops.bottleneck_chain(c1, w2, b64, w3, b256, residual=idm, w1n_planes=w1n, b1n=b64, out=idm), fl(64, 64)))
    cases.append(('layer1.2 chain (3x3 | conv3 + identity | layer2 conv1 128), in place',
                  lambda: ops.bottleneck_chain(c1, w2, b64, w3, b256, residual=idm, w1n_planes=w1n128, b1n=b128, out=idm), fl(64, 128)))
    # the half-tail form (16-column tail on 16x16x32 MFMAs over slab pairs; the shipped selection, variant 0)
    # against the zero-padded 64-column form (variant 17) on HRNet-w48's 48-channel 3x3
    x48, id48 = cl(F, 200, 336, 48), cl(F, 200, 336, 48)
    w48, b48 = ops.split_conv3x3_weight(rnd(48, 48, 3, 3, sc=0.05)), rnd(48)
    ht = lambda: ops.conv3x3_split(x48, w48, b48, relu=True, residual=id48, cout=48)   # noqa: E731
    tt = {17: [], 0: []}
    for rd in range(4):
        for v in (17, 0):
            with native.diag_build(v):
                t = timed(ht)
            if rd:
                tt[v].append(t)
    a, b = sorted(tt[17])[1], sorted(tt[0])[1]
    fl48 = 2 * F * 200 * 336 * 9 * 48 * 48
    print(f'3x3 48 -> 48 + identity + ReLU, {F} x 200 x 336: padded 64-column form {a:8.1f} us ({fl48 / a * 1e-6:5.1f} TF/s)   '
          f'half-tail form {b:8.1f} us ({fl48 / b * 1e-6:5.1f} TF/s)   {100 * (b / a - 1):+5.1f} %', flush=True)
    res = {(n, v): [] for n, _, _ in cases for v in (15, 16)}
    for rd in range(4):
        for name, fn, _ in cases:
            for v in (15, 16):
                with native.diag_build(v):
                    t = timed(fn)
                if rd:
                    res[(name, v)].append(t)
    for name, _, flops in cases:
        a, b = sorted(res[(name, 15)])[1], sorted(res[(name, 16)])[1]
        print(f'{name:68s} one row tile {a:8.1f} us ({flops / a * 1e-6:5.1f} TF/s)   two {b:8.1f} us '
              f'({flops / b * 1e-6:5.1f} TF/s)   {100 * (b / a - 1):+5.1f} %', flush=True)


if __name__ == '__main__':
    main()
