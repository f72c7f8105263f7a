"""does the VALU-bound encoder sampler hide under an MFMA-bound GEMM?  Timing-only probe.
Half batch A (14 frames) runs its sampler launch (`enc_tile_kernel`, prepared mode) on one HIP stream while half
batch B runs its FFN1 GEMM (312 522 x 256 x 1024 + ReLU, `gemm_w_kernel<0>`) on another -- once with the GEMM as it
ships (two blocks per CU: 480 of 512 VGPRs per SIMD, 128 KB of LDS -- no room for a sampler block), once capped at ONE
block per CU (diag variant 20: 40 KiB of unused dynamic LDS), which leaves 272 VGPRs per SIMD and 56 KB of LDS = one
6-wave sampler block beside it.  Prints each launch alone, the pair's wall time, and pair - max / pair - sum.
The two-half-batch encoder schedule is worth building only if a pair runs >= 0.5 ms under the sum of its parts.
    python tools/coresidency_probe.py [frames_per_half=14]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pavenet_amd import native, ops  # noqa: E402

LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21)]
F = int(sys.argv[1]) if len(sys.argv) > 1 else 14
dev = 'cuda'
S = sum(h * w for h, w in LEVELS)
g = torch.Generator(device=dev).manual_seed(0)
# half batch A: the sampler's inputs as the merged projection GEMM leaves them (prepared = pixel coords + weights)
x_a = torch.randn(F * S, 256, device=dev, generator=g)
w_all = torch.randn(640, 256, device=dev, generator=g) * 0.05
w_all[256:512] *= 0.3                                    # offsets of ~1 px
table = torch.randn(S, 640, device=dev, generator=g) * 0.1
ys = torch.cat([((torch.arange(h * w) // w).float() + 0.5) / h for h, w in LEVELS])
xs = torch.cat([((torch.arange(h * w) % w).float() + 0.5) / w for h, w in LEVELS])
ref = torch.stack([xs, ys], -1)[None, :, None, :].expand(F, -1, 4, 2).reshape(F * S, 4, 2).contiguous().to(dev)
value, samp = ops.gemm_bf16x3_encproj(x_a, ops.split_weight_bf16x3(w_all), table, ref, LEVELS)
# half batch B: FFN1
x_b = torch.randn(F * S, 256, device=dev, generator=g)
w1 = ops.split_weight_bf16x3(torch.randn(1024, 256, device=dev, generator=g) * 0.06)
b1 = torch.randn(1024, device=dev, generator=g)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def sampler():
    return ops.deform_attn_enc_tile(value.view(F, S, 8, 32), samp, None, levels_hw=LEVELS, prepared=True)


def gemm():
    return ops.gemm_bf16x3(x_b, w1, b1, relu=True)


def alone(fn, iters=10):
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


def pair(iters=10):
    """the two launches of a layer pair issued together on two streams, `iters` pairs back to back; wall per pair"""
    def once():
        with torch.cuda.stream(sa):
            sampler()
        with torch.cuda.stream(sb):
            gemm()
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    s.record()
    sa.wait_stream(cur)
    sb.wait_stream(cur)
    for _ in range(iters):
        once()
        sa.wait_stream(sb)          # the next pair starts when both launches of this one are done
        sb.wait_stream(sa)
    cur.wait_stream(sa)
    cur.wait_stream(sb)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


print(f'{F} frames per half batch: sampler launch = {F} x {S} tokens, GEMM = {F * S} x 256 x 1024 + ReLU (us)')
for v, what in ((0, 'GEMM as shipped (2 blocks per CU)'), (20, 'GEMM capped at 1 block per CU (diag variant 20)')):
    with native.diag_build(v):
        t_s, t_g = alone(sampler), alone(gemm)
        t_p = pair()
    print(f'{what:48s}: sampler alone {t_s:7.1f}   GEMM alone {t_g:7.1f}   sum {t_s + t_g:7.1f}   pair {t_p:7.1f}   '
          f'pair - sum {t_p - t_s - t_g:+7.1f}   pair - max {t_p - max(t_s, t_g):+7.1f}', flush=True)
