"""Which vendor ops are run-to-run non-deterministic at the test sizes?"""
import torch, torch.nn.functional as F
def rep(name, fn, n=6):
    ref = fn().clone(); bad = 0
    for _ in range(n):
        bad += int(not torch.equal(fn(), ref))
    print(f'{name:40s} non-identical reruns: {bad}/{n}')
for cl in (True, False):
    mf = torch.channels_last if cl else torch.contiguous_format
    x = torch.randn(4, 3, 128, 160, device='cuda').contiguous(memory_format=mf)
    w = torch.randn(64, 3, 7, 7, device='cuda').contiguous(memory_format=mf)
    rep(f'conv7x7 s2 cl={cl}', lambda: F.conv2d(x, w, None, 2, 3))
    for c, h in ((64, 32), (128, 16), (256, 8), (512, 4)):
        x3 = torch.randn(4, c, h, h * 5 // 4, device='cuda').contiguous(memory_format=mf)
        w3 = torch.randn(c, c, 3, 3, device='cuda').contiguous(memory_format=mf)
        rep(f'conv3x3 c={c} cl={cl}', lambda: F.conv2d(x3, w3, None, 1, 1))
        rep(f'conv3x3 s2 c={c} cl={cl}', lambda: F.conv2d(x3, w3, None, 2, 1))
        w1 = torch.randn(4 * c, c, 1, 1, device='cuda').contiguous(memory_format=mf)
        rep(f'conv1x1 c={c} cl={cl}', lambda: F.conv2d(x3, w1))
for M, K, N in ((5120, 64, 256), (5120, 256, 64), (1280, 512, 128), (320, 1024, 256), (80, 2048, 512), (1704, 256, 1024), (1704, 1024, 256), (1704, 256, 384)):
    a = torch.randn(M, K, device='cuda'); wt = torch.randn(N, K, device='cuda'); b = torch.randn(N, device='cuda')
    rep(f'addmm {M}x{K}x{N}', lambda: torch.addmm(b, a, wt.t()))
    rep(f'_addmm_activation {M}x{K}x{N}', lambda: torch._addmm_activation(b, a, wt.t()))
g = torch.nn.GroupNorm(32, 256).cuda(); xg = torch.randn(4, 256, 16, 20, device='cuda')
rep('groupnorm', lambda: g(xg))
