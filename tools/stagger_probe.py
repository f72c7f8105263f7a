"""Are the two workgroups that share a CU in step?  (-DPAVE_DIAG build only.)

The wide GEMM forms run two 4-wave workgroups per CU.  Both start a launch together; tiles take equal time, so
unless something de-phases them their main loops (matrix pipe) and their epilogues (HBM) coincide for the whole
launch, and the launch takes MFMA time + epilogue time instead of the larger of the two.  This probe makes the
odd-slot workgroup of every CU sleep n x 8 128 shader clocks (n x ~4.6 us) before its first tile
(`g_diag_stagger`, pave_gemm_dma.hip) and times the launch for a sweep of n on the shapes whose epilogue is a
large share of the tile: encoder out_proj + LN, FFN1, FFN2 + LN, the merged projection, ResNet conv3.

    python tools/stagger_probe.py [frames=28] [inplace|fresh]
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


INPLACE = True


def main():
    global INPLACE
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    INPLACE = (sys.argv[2] if len(sys.argv) > 2 else 'inplace') == 'inplace'
    dev = 'cuda'
    lib = native.use_diag_build(0)
    lib.pave_diag_set_stagger.argtypes = [ctypes.c_int]
    lib.pave_diag_hwid_read.argtypes = [ctypes.c_void_p]
    S = n * 22323
    shapes = [('enc.out_proj+LN', S, 256, 256, 'ln'), ('enc.ffn1', S, 256, 1024, 'relu'),
              ('enc.ffn2+LN', S, 1024, 256, 'ln'), ('layer2.conv3', n * 100 * 168, 128, 512, 'res'),
              ('layer3.conv3', n * 50 * 84, 256, 1024, 'res'), ('layer3.conv1', n * 50 * 84, 1024, 256, 'relu'),
              ('value_proj', S, 256, 256, 'bias')]
    sweeps = [0] + [m * 1000 + k for m in (0, 1, 2) for k in (2, 4, 6, 9, 12)]
    print('# us per launch; columns: stagger off | mode 0 (wave-id bit) n = 2 4 6 9 12 | mode 1 (tg-id bit) ... | '
          'mode 2 (blockIdx bit: control) ...')
    for label, M, K, N, kind in shapes:
        a = torch.randn(M, K, device=dev)
        wt = torch.randn(N, K, device=dev) * 0.05
        bias = torch.randn(N, device=dev)
        wp = ops.split_weight_bf16x3(wt)
        c = torch.randn(M, N, device=dev) if kind in ('ln', 'res') else None
        # (as the model calls them: the LayerNorm / residual forms write over their identity rows)
        out = c if (c is not None and INPLACE) else torch.empty(M, N, device=dev)
        if kind == 'ln':
            gam, bet = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
            fn = lambda: ops.gemm_bf16x3_ln(a, wp, bias, c, gam, bet, 1e-5, out=out)
        elif kind == 'res':
            fn = lambda: ops.gemm_bf16x3(a, wp, bias, c, relu=True, out=out)
        else:
            fn = lambda: ops.gemm_bf16x3(a, wp, bias, None, relu=(kind == 'relu'), out=out)
        for _ in range(20):   # (the first measurements of a process run below the steady clock state)
            fn()
        row = []
        ref = None
        for v in sweeps:
            lib.pave_diag_set_stagger(v)
            row.append(timed(fn))
            if out is not c:
                if v == 0:
                    ref = out.clone()
                else:
                    assert torch.equal(out, ref), (label, v)
        lib.pave_diag_set_stagger(0)
        print(f'{label:18s} M={M:7d} K={K:4d} N={N:4d}  ' + '  '.join(
            f'{row[0]:7.1f}' if i == 0 else ('| ' if (i - 1) % 5 == 0 else '') + f'{t:7.1f}'
            for i, t in enumerate(row)))
        del a, wt, c, out
    # where the first-round workgroups sat: HW_ID of wave 0 (gfx9 layout: wave 3:0, simd 5:4, cu 11:8, sh 12,
    # se 15:13, tg 19:16)
    a = torch.randn(S, 256, device=dev)
    wp = ops.split_weight_bf16x3(torch.randn(256, 256, device=dev) * 0.05)
    lib.pave_diag_set_stagger(1)
    ops.gemm_bf16x3(a, wp)
    buf = (ctypes.c_uint * 1024)()
    lib.pave_diag_hwid_read(ctypes.cast(buf, ctypes.c_void_p))
    lib.pave_diag_set_stagger(0)
    print('# HW_ID of the first 24 workgroups and of 512..519 (block: wave simd cu sh se tg | raw)')
    for b in list(range(24)) + list(range(512, 520)):
        h = buf[b]
        print(f'  {b:4d}: wave {h & 15} simd {(h >> 4) & 3} cu {(h >> 8) & 15} sh {(h >> 12) & 1} '
              f'se {(h >> 13) & 7} tg {(h >> 16) & 15} | {h:#010x}')
    import collections
    pairs = collections.Counter()
    for b in range(512):
        h = buf[b]
        pairs[(b & 7, (h >> 8) & 15, (h >> 12) & 1, (h >> 13) & 7)] += 1
    print('# first-round workgroups per (xcd = block & 7, cu, sh, se): histogram of counts',
          dict(collections.Counter(pairs.values())))
    bits = collections.Counter((buf[b] & 1, (buf[b] >> 16) & 1) for b in range(512))
    print('# (wave-id bit, tg-id bit) over the first 512 workgroups:', dict(bits))


if __name__ == '__main__':
    main()
