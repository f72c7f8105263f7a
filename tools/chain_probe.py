"""What do the layer1 chain's body-to-body hand-overs cost?  (-DPAVE_DIAG build; timing only -- the probe
launches compute WRONG results on purpose.)

`pave_bottleneck_chain_f32` runs three GEMM bodies per row tile (3x3 | conv3 + identity | next conv1); between
them the tile's rows go through global memory behind a workgroup fence + barrier, and each body starts with an
empty DMA ring.  Probes: the same launches (a) without the fences / barriers between the bodies (`g_diag_stagger`
= -4), (b) with every store dropped (-1: no HBM writes; the bodies read whatever the buffers held), (c) both (-5), (d) with the A operand's three planes taken as the raw fp32 bits (-7: no vector arithmetic for the split);
and the three bodies as separate launches for scale.

    python tools/chain_probe.py [frames=28]
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pavenet_amd import native, ops  # noqa: E402


def timed(fn, iters=10):
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
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    dev = 'cuda'
    lib = native.use_diag_build(0)
    lib.pave_diag_set_stagger.argtypes = [ctypes.c_int]
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s, sc=1.0: torch.randn(*s, device=dev, generator=g) * sc   # noqa: E731
    cl = lambda n, h, w, c: rnd(n, h, w, c).relu_().permute(0, 3, 1, 2)      # noqa: E731
    mk = lambda n, k: ops.split_weight_bf16x3(rnd(n, k, sc=0.05), pad=True)   # noqa: E731
    H, W = 200, 336
    M = F * H * W
    c1, idm, x64 = cl(F, H, W, 64), cl(F, H, W, 256), cl(F, H, W, 64)
    w2 = ops.split_conv3x3_weight(rnd(64, 64, 3, 3, sc=0.05))
    w3, w3d, w1n, w1n128 = mk(256, 64), mk(256, 128), mk(64, 256), mk(128, 256)
    b64, b256, b128 = rnd(64), rnd(256), rnd(128)
    cases = [
        ('layer1.0 chain (3x3 | conv3 + downsample | next conv1)',
         lambda: ops.bottleneck_chain(c1, w2, b64, w3d, b256, a2=x64, w1n_planes=w1n, b1n=b64)),
        ('layer1.1 chain (3x3 | conv3 + identity | next conv1), in place',
         lambda: ops.bottleneck_chain(c1, w2, b64, w3, b256, residual=idm, w1n_planes=w1n, b1n=b64, out=idm)),
        ('layer1.2 chain (3x3 | conv3 + identity | layer2 conv1 128), in place',
         lambda: ops.bottleneck_chain(c1, w2, b64, w3, b256, residual=idm, w1n_planes=w1n128, b1n=b128, out=idm)),
    ]
    for _ in range(10):
        cases[1][1]()
    torch.cuda.synchronize()
    print(f'# {F} frames of {H} x {W}: {M} pixels; us per launch')
    print(f'# {"launch":72s} {"shipped":>9s} {"no syncs":>9s} {"no stores":>10s} {"neither":>9s} {"free split":>10s}')
    for label, fn in cases:
        row = []
        for v in (0, -4, -1, -5, -7):
            lib.pave_diag_set_stagger(v)
            row.append(timed(fn))
        lib.pave_diag_set_stagger(0)
        print(f'  {label:72s} ' + ' '.join(f'{t:9.1f}' for t in row))
    # the bodies as launches of their own
    c2 = torch.empty(F, 64, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    a64 = rnd(M, 64).relu_()
    a256 = rnd(M, 256).relu_()
    idr = rnd(M, 256)
    o256 = torch.empty(M, 256, device=dev)
    a1024 = rnd(F * 50 * 84, 1024).relu_()
    w1024 = mk(256, 1024)
    parts = [
        ('(for scale) layer3 conv1 1024 -> 256 + ReLU, 256-column tiles', lambda: ops.gemm_bf16x3(a1024, w1024, b256, relu=True)),
        ('3x3 64 -> 64 + ReLU', lambda: ops.conv3x3_split(c1, w2, b64, relu=True, cout=64)),
        ('conv3 64 -> 256 + identity + ReLU', lambda: ops.gemm_bf16x3(a64, w3, b256, idr, relu=True, out=o256)),
        ('conv1 256 -> 64 + ReLU', lambda: ops.gemm_bf16x3(a256, w1n, b64, relu=True)),
    ]
    tot = [0.0, 0.0]
    for label, fn in parts:
        row = []
        for v in (0, -1, -7):
            lib.pave_diag_set_stagger(v)
            row.append(timed(fn))
        lib.pave_diag_set_stagger(0)
        if not label.startswith('(for scale)'):
            tot[0] += row[0]
            tot[1] += row[1]
        print(f'  {label:72s} {row[0]:9.1f} {"":>9s} {row[1]:10.1f} {"":>9s} {row[2]:10.1f}')
    print(f'  {"sum of the three bodies as separate launches":72s} {tot[0]:9.1f} {"":>9s} {tot[1]:10.1f}')
    del c2


if __name__ == '__main__':
    main()
