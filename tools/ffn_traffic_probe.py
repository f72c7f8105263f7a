"""What would an encoder FFN cost without the HBM round trip of its 1024-wide hidden rows?  (-DPAVE_DIAG build;
timing only -- the probe launches compute WRONG results on purpose.)

FFN1 (625 044 x 256 x 1024) writes 2.56 GB of hidden rows, FFN2 + LN reads them back.  A fused FFN (hidden tile kept
on the CU) would do the same matrix work without those 5.1 GB.  Upper bound of what that buys: FFN1 with its stores
dropped (out-of-range buffer offsets) + FFN2 with every row tile reading the A rows of tile 0 (L2 hits), against the
two launches as shipped.  Also: each launch with ONE workgroup per CU instead of two (diag variant 20), i.e. with no
co-resident block to overlap a tile's epilogue with.

    python tools/ffn_traffic_probe.py [frames=28]
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    dev = 'cuda'
    lib = native.use_diag_build(0)
    lib.pave_diag_set_stagger.argtypes = [ctypes.c_int]
    S = n * 22323
    x = torch.randn(S, 256, device=dev)
    h = torch.randn(S, 1024, device=dev).relu_()
    w1 = ops.split_weight_bf16x3(torch.randn(1024, 256, device=dev) * 0.05)
    w2 = ops.split_weight_bf16x3(torch.randn(256, 1024, device=dev) * 0.03)
    wv = ops.split_weight_bf16x3(torch.randn(256, 256, device=dev) * 0.05)
    b1, b2 = torch.randn(1024, device=dev), torch.randn(256, device=dev)
    gam, bet = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev)
    idt = torch.randn(S, 256, device=dev)
    hid = torch.empty(S, 1024, device=dev)
    out = torch.empty(S, 256, device=dev)
    launches = [
        ('FFN1  x[S,256] -> relu -> hidden[S,1024]', lambda: ops.gemm_bf16x3(x, w1, b1, None, relu=True, out=hid)),
        ('FFN2 + LN  hidden[S,1024] -> [S,256]', lambda: ops.gemm_bf16x3_ln(h, w2, b2, idt, gam, bet, 1e-5, out=out)),
        ('out_proj + LN  [S,256] -> [S,256]', lambda: ops.gemm_bf16x3_ln(x, wv, b2, idt, gam, bet, 1e-5, out=out)),
        ('value_proj  [S,256] -> [S,256]', lambda: ops.gemm_bf16x3(x, wv, b2, None, relu=False, out=out)),
    ]
    # a few warm launches first: the first measurements of a process run below the steady clock state
    for _ in range(20):
        launches[0][1]()
    torch.cuda.synchronize()
    print(f'# {n} frames, S = {S} rows; us per launch')
    print(f'# {"launch":44s} {"shipped":>9s} {"no stores":>10s} {"A in L2":>9s} {"both":>9s} {"1 blk/CU":>9s} {"nt stores":>10s}')
    for label, fn in launches:
        row = []
        for v in (0, -1, -2, -3):
            lib.pave_diag_set_stagger(v)
            row.append(timed(fn))
        lib.pave_diag_set_stagger(0)
        native.use_diag_build(20)
        row.append(timed(fn))
        native.use_diag_build(0)
        lib.pave_diag_set_stagger(-6)      # (stores with the non-temporal hint: same values)
        row.append(timed(fn))
        lib.pave_diag_set_stagger(0)
        row.append(timed(fn))              # (shipped again: the drift of the box over the row)
        print(f'  {label:44s} ' + ' '.join(f'{t:9.1f}' for t in row))
    # the same arithmetic on all-zero operands (the matrix pipe's data-dependent power)
    x.zero_()
    h.zero_()
    print('# all-zero A operands (same launches, same instruction stream):')
    for label, fn in launches[:2]:
        row = []
        for v in (0, -1, -2, -3):
            lib.pave_diag_set_stagger(v)
            row.append(timed(fn))
        lib.pave_diag_set_stagger(0)
        print(f'  {label:44s} ' + ' '.join(f'{t:9.1f}' for t in row))


if __name__ == '__main__':
    main()
