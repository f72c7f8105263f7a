"""Experiment helper: times the split-GEMM entry points of several builds of libpave_hip.so
(tools/build_variants.py) on the bench workload's shapes, in one process, interleaved, and checks
every variant's output against the first one bit for bit.
    python tools/gemm_ab.py base sched ...        (names under pavenet_amd/lib/variants/)"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pavenet_amd import native  # noqa: E402

vp, ci, ll, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_float


def load(name):
    """`name` = a build under pavenet_amd/lib/variants/, or `main` = the regular library; an
    `@V` suffix calls pave_diag_gemm_variant(V) before every timed call (e.g. main@5)."""
    name = name.split('@')[0]
    path = os.path.join(ROOT, 'pavenet_amd', 'lib', 'libpave_hip_diag.so') if name == 'main' else \
        os.path.join(ROOT, 'pavenet_amd', 'lib', 'variants', f'libpave_hip_{name}.so')
    lib = ctypes.CDLL(path)
    for fn, sig in native.SIGNATURES.items():
        f = getattr(lib, fn)
        f.argtypes, f.restype = sig, ci
    return lib


def timed(fn, iters):
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
    names = sys.argv[1:]
    libs = [load(n) for n in names]
    dev = 'cuda'
    st = torch.cuda.current_stream().cuda_stream
    frames = 28
    S = frames * 22323
    from pavenet_amd import ops   # weight layout helpers only (run on the regular library)
    cases = []   # (label, flops, make(lib) -> (callable, out tensor))
    g = torch.Generator(device=dev).manual_seed(0)

    def gemm_case(label, M, K, N, relu=False, res=False, ln=False):
        a = torch.randn(M, K, device=dev, generator=g)
        w = torch.randn(N, K, device=dev, generator=g) * 0.05
        wp = ops.split_weight_bf16x3(w)
        bias = torch.randn(N, device=dev, generator=g)
        r = torch.randn(M, N, device=dev, generator=g) if res else None
        gam, bet = torch.rand(N, device=dev, generator=g) + 0.5, torch.randn(N, device=dev, generator=g)
        out = torch.empty(M, N, device=dev)

        def make(lib):
            if ln:
                return lambda: lib.pave_gemm_bf16x3_ln_f32(a.data_ptr(), wp.data_ptr(), bias.data_ptr(),
                                                           r.data_ptr() if res else None, gam.data_ptr(),
                                                           bet.data_ptr(), 1e-5, out.data_ptr(), M, K, N, 3, st)
            return lambda: lib.pave_gemm_bf16x3_f32(a.data_ptr(), None, wp.data_ptr(), bias.data_ptr(),
                                                    r.data_ptr() if res else None, out.data_ptr(), M, K, N,
                                                    int(relu), 3, st)
        cases.append((label, 2.0 * M * K * N, make, out))

    def conv_case(label, n, H, W, Cin, Cout, stride=1):
        x = torch.randn(n, H, W, Cin, device=dev, generator=g)
        w = torch.randn(Cout, Cin, 3, 3, device=dev, generator=g) * 0.03
        wp = ops.split_conv3x3_weight(w)
        bias = torch.randn(Cout, device=dev, generator=g)
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        out = torch.empty(n, Ho, Wo, Cout, device=dev)

        def make(lib):
            return lambda: lib.pave_conv3x3_split_f32(x.data_ptr(), wp.data_ptr(), bias.data_ptr(), None,
                                                      out.data_ptr(), n, H, W, Cin, Cout, stride, 1, 3, st)
        cases.append((label, 2.0 * n * Ho * Wo * Cout * 9 * Cin, make, out))

    gemm_case('ffn1   625044x256x1024 relu', S, 256, 1024, relu=True)
    gemm_case('ffn2+ln 625044x1024x256', S, 1024, 256, res=True, ln=True)
    gemm_case('ffn2   625044x1024x256 res', S, 1024, 256, res=True)
    gemm_case('merged 625044x256x640', S, 256, 640)
    gemm_case('out+ln 625044x256x256', S, 256, 256, res=True, ln=True)
    gemm_case('l2.c1  470400x512x128 relu', frames * 100 * 168, 512, 128, relu=True)
    gemm_case('l3.c3  117600x256x1024 res', frames * 50 * 84, 256, 1024, res=True, relu=True)
    gemm_case('l1.c1  1881600x256x64 relu', frames * 200 * 336, 256, 64, relu=True)
    conv_case('l2 3x3 100x168 128->128', frames, 100, 168, 128, 128)
    conv_case('l3 3x3 50x84 256->256', frames, 50, 84, 256, 256)
    conv_case('l1 3x3 200x336 64->64', frames, 200, 336, 64, 64)
    for label, flops, make, out in cases:
        ref = None
        row = []
        for name, lib in zip(names, libs):
            if hasattr(lib, 'pave_diag_gemm_variant'):   # -DPAVE_DIAG builds only
                lib.pave_diag_gemm_variant(int(name.split('@')[1]) if '@' in name else 0)
            fn = make(lib)
            assert fn() == 0, (name, label, lib.pave_last_error())
            torch.cuda.synchronize()
            o = out.clone()
            if ref is None:
                ref = o
            same = bool(torch.equal(o, ref))
            if not same:
                d = (o - ref).abs()
                print(f'      {name} {label}: max |diff| {float(d.max()):.3e} at {int(d.argmax())}, '
                      f'mismatching {int((o != ref).sum())} of {o.numel()}', flush=True)
            best = min(timed(fn, 6) for _ in range(2))
            row.append(f'{name} {best:.3f} ms {flops / best / 1e9:5.0f} TF/s{"" if same else " (DIFFERS)"}')
        print(f'{label:30s} ' + ' | '.join(row), flush=True)


if __name__ == '__main__':
    main()
