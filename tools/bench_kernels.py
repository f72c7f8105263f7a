"""Micro-benchmark of the HIP sampling kernels at BASELINE sizes (800x1344 -> S = 22 323).
Prints one line per kernel: us per launch and algorithmic GB/s (SURVEY 8d byte counts)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pavenet_amd import ops  # noqa: E402

LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21)]


def timeit(fn, iters=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def grid_refs(frames, dev):
    ys = torch.cat([((torch.arange(h * w) // w).float() + 0.5) / h for h, w in LEVELS])
    xs = torch.cat([((torch.arange(h * w) % w).float() + 0.5) / w for h, w in LEVELS])
    r = torch.stack([xs, ys], -1)[None, :, None, :].expand(frames, -1, 4, 2)
    return r.reshape(1, -1, 4, 2).contiguous().to(dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=3)
    ap.add_argument('--sigma', type=float, default=3.0, help='offset std-dev in pixels')
    ap.add_argument('--order', default='none')
    ap.add_argument('--enc-only', action='store_true')
    ap.add_argument('--ablate', action='store_true')
    ap.add_argument('--prepared', action='store_true',
                    help='time the sampler in its prepared mode (pixel coordinates + attention weights, as the '
                         "merged projection GEMM's epilogue leaves them: the product's encoder path)")
    ap.add_argument('--ray', type=float, default=0.0,
                    help='add the reference init pattern to the offsets: head h, point i at (i + 1) '
                         '* ray * dir(h) px (MO:227-240 uses ray = 1)')
    args = ap.parse_args()
    dev = 'cuda'
    shapes = torch.as_tensor(LEVELS, dtype=torch.long, device=dev)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    F = args.frames
    g = torch.Generator(device=dev).manual_seed(0)
    value = torch.randn(F, S, 8, 32, device=dev, generator=g)
    U = F * S
    proj = torch.randn(U, 384, device=dev, generator=g)
    proj[:, :256] *= args.sigma
    ref = grid_refs(F, dev)
    shift = None
    if args.ray:
        import math
        th = torch.arange(8, dtype=torch.float32) * (2.0 * math.pi / 8)
        gi = torch.stack([th.cos(), th.sin()], -1)
        gi = (gi / gi.abs().max(-1, keepdim=True)[0]).view(8, 1, 1, 2).repeat(1, 4, 4, 1)
        for i in range(4):
            gi[:, :, i, :] *= (i + 1) * args.ray
        proj[:, :256] += gi.reshape(1, 256).to(dev)
        shift = ops.enc_tile_window_shift(gi.reshape(-1))
    order = None
    if args.order != 'none':
        from pavenet_amd.locality import encoder_unit_order
        order = encoder_unit_order(LEVELS, F, mode=args.order).to(dev)
    us = timeit(lambda: ops.deform_attn_grid_fused(value, shapes, lsi, proj, ref, T=1, n_clips=F,
                                                   units_per_clip=S, order=order))
    alg = 4 * (F * S * 256 + U * 384 + U * 256)
    print(f'enc_fused   frames={F} order={args.order}: {us:9.1f} us  {us / F:8.1f} us/frame  '
          f'alg {alg / us / 1e3:7.1f} GB/s')
    if args.prepared:
        sizes = torch.tensor([[w, h] for h, w in LEVELS], dtype=torch.float32, device=dev)   # (W, H) per level
        off = proj[:, :256].view(U, 8, 4, 4, 2)
        px = (ref.view(U, 1, 4, 1, 2) + off / sizes[None, None, :, None, :]) * sizes[None, None, :, None, :] - 0.5
        samp = torch.cat([px.reshape(U, 256), proj[:, 256:].view(U, 8, 16).softmax(-1).reshape(U, 128)], 1).contiguous()
        us = timeit(lambda: ops.deform_attn_enc_tile(value, samp, None, levels_hw=LEVELS, prepared=True))
        print(f'enc_tile prepared frames={F} sigma={args.sigma} ray={args.ray}: {us:9.1f} us  {us / F:8.1f} us/frame  '
              f'alg {alg / us / 1e3:7.1f} GB/s = {alg / us / 1e3 / 8000:.3f} of 8 TB/s')
        if shift is not None:
            us = timeit(lambda: ops.deform_attn_enc_tile(value, samp, None, levels_hw=LEVELS, prepared=True,
                                                         window_shift=shift))
            print(f'enc_tile prepared ... with the per-head window shift: {us:9.1f} us  '
                  f'= {alg / us / 1e3 / 8000:.3f} of 8 TB/s')
    for variant in (0, 1):
        us = timeit(lambda: ops.deform_attn_enc_tile(value, proj, ref, levels_hw=LEVELS,
                                                     variant=variant))
        print(f'enc_tile v{variant} frames={F} sigma={args.sigma} ray={args.ray}: {us:9.1f} us  {us / F:8.1f} us/frame  '
              f'alg {alg / us / 1e3:7.1f} GB/s = {alg / us / 1e3 / 8000:.3f} of 8 TB/s')
        if shift is not None:
            us = timeit(lambda: ops.deform_attn_enc_tile(value, proj, ref, levels_hw=LEVELS,
                                                         variant=variant, window_shift=shift))
            print(f'enc_tile v{variant} ... with the per-head window shift: {us:9.1f} us  '
                  f'= {alg / us / 1e3 / 8000:.3f} of 8 TB/s')
    if args.ablate:
        import ctypes
        from pavenet_amd import native
        lib = native.use_diag_build()     # the -DPAVE_DIAG build has the ablation entry point
        fn = lib.pave_diag_enc_tile_ablate
        fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p] + \
            [ctypes.c_int] * 3 + [ctypes.c_void_p]
        hw = (ctypes.c_int * 8)(*[v for hw_ in LEVELS for v in hw_])
        out = torch.empty(U, 256, device=dev)
        for ab in (0, 1, 2, 3):
            us = timeit(lambda: fn(value.data_ptr(), proj.data_ptr(), ref.data_ptr(), out.data_ptr(),
                                   F, S, ctypes.cast(hw, ctypes.c_void_p), 384, 0, ab,
                                   torch.cuda.current_stream().cuda_stream))
            print(f'enc_tile v0 ablate={ab} (1: no staging, 2: no gather): {us:9.1f} us')
    if args.enc_only:
        return
    # the un-fused reference-shaped op on the same work
    off = proj[:, :256].view(F, S, 8, 4, 4, 2)
    norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float()
    loc = (ref.view(F, S, 1, 4, 1, 2) + off / norm[None, None, None, :, None, :]).contiguous()
    aw = proj[:, 256:].view(F, S, 8, 16).softmax(-1).view(F, S, 8, 4, 4).contiguous()
    us = timeit(lambda: ops.ms_deform_attn_forward(value, shapes, lsi, loc, aw, 64))
    alg2 = 4 * (F * S * 256 + U * 8 * 16 * 3 + U * 256)
    print(f'msda_generic frames={F}: {us:9.1f} us  {us / F:8.1f} us/frame  alg {alg2 / us / 1e3:7.1f} GB/s')
    # pose decoder shape: Q=300, K=15, T frames of one clip
    for T in (3, 7):
        if T > F:
            continue
        Q, K = 300, 15
        pp = torch.randn(Q, T * 8 * 4 * K * 3, device=dev, generator=g)
        rp = torch.rand(1, T * Q, 4, 2 * K, device=dev, generator=g) * 0.5 + 0.25
        us = timeit(lambda: ops.deform_attn_pose_fused(value[:T], shapes, lsi, pp, rp, T=T,
                                                       n_clips=1, num_query=Q, num_keypoints=K))
        print(f'pose_fused  T={T} Q=300: {us:9.1f} us')
        N = 20
        jp = torch.randn(N * K, T * 8 * 16 * 3, device=dev, generator=g)
        jp[:, :T * 256] *= args.sigma
        jr = torch.rand(T, N * K, 4, 2, device=dev, generator=g) * 0.8 + 0.1
        uc = torch.zeros(N * K, dtype=torch.int32, device=dev)
        us = timeit(lambda: ops.deform_attn_grid_fused(value[:T], shapes, lsi, jp, jr, T=T,
                                                       n_clips=1, units_per_clip=N * K,
                                                       unit_clip=uc))
        print(f'joint_fused T={T} N=20: {us:9.1f} us')


if __name__ == '__main__':
    main()
