"""Where the merged encoder projection's epilogue output differs from the plain GEMM's + torch arithmetic."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pavenet_amd import ops  # noqa: E402

levels = [(100, 168), (50, 84), (25, 42), (13, 21)]
F = 2
S = sum(h * w for h, w in levels)
M, K = F * S, 256
g = torch.Generator(device='cuda').manual_seed(S + F)
a = torch.randn(M, K, device='cuda', generator=g)
w = torch.randn(640, K, device='cuda', generator=g) * 0.05
table = torch.randn(S, 640, device='cuda', generator=g) * 0.1
wp = ops.split_weight_bf16x3(w)
ref = torch.rand(M, 4, 2, device='cuda', generator=g)
v0, proj = ops.gemm_bf16x3_ex(a, wp, None, table, residual_rows=S, n_split=256)
v1, samp = ops.gemm_bf16x3_encproj(a, wp, table, ref, levels)
full = ops.gemm_bf16x3(a, wp) + table.repeat(F, 1)
print('value equal', torch.equal(v0, v1), 'ex vs plain+table: value', float((v0 - full[:, :256]).abs().max()),
      'proj', float((proj - full[:, 256:]).abs().max()))
sizes = torch.tensor([[w_, h] for h, w_ in levels], dtype=torch.float32, device='cuda')
off = proj[:, :256].view(M, 8, 4, 4, 2)
px = (ref.view(M, 1, 4, 1, 2) + off / sizes[None, None, :, None, :]) * sizes[None, None, :, None, :] - 0.5
aw = proj[:, 256:].view(M, 8, 16).softmax(-1)
d1 = (samp[:, :256] - px.reshape(M, 256)).abs()
d2 = (samp[:, 256:] - aw.reshape(M, 128)).abs()
print('coords max diff', float(d1.max()), 'weights max diff', float(d2.max()))
for name, d in (('coords', d1), ('weights', d2)):
    bad = (d > 1e-3).nonzero()
    print(name, 'bad entries', bad.shape[0])
    if bad.shape[0]:
        rows = bad[:, 0].unique()
        cols = bad[:, 1].unique()
        print('  rows', rows[:20].tolist(), '... n =', rows.numel(), ' rows % 128:', (rows % 128).unique()[:40].tolist())
        print('  cols', cols[:40].tolist(), '... n =', cols.numel())

print('--- pieces')
def cmp(name, got, exp):
    d = (got - exp).abs()
    bad = (d > 1e-4).nonzero()
    print(f'{name}: max diff {float(d.max()):.4f}, bad {bad.shape[0]}', end='')
    if bad.shape[0]:
        print('  rows%32', (bad[:, 0] % 32).unique().tolist(), ' cols%32', (bad[:, 1] % 32).unique().tolist(),
              ' col tiles', (bad[:, 1] // 32).unique().tolist())
    else:
        print()
plain = ops.gemm_bf16x3(a, wp)
for rep in range(2):
    t_only, _ = ops.gemm_bf16x3_ex(a, wp, None, table, residual_rows=S)
    cmp('N=640 table, one output', t_only, plain + table.repeat(F, 1))
    s0, s1 = ops.gemm_bf16x3_ex(a, wp, None, None, n_split=256)
    cmp('N=640 split only: out', s0, plain[:, :256])
    cmp('N=640 split only: out2', s1, plain[:, 256:])
    fullres = table.repeat(F, 1).contiguous()
    r_full = ops.gemm_bf16x3(a, wp, None, fullres)
    cmp('N=640 full residual', r_full, plain + fullres)
w512 = w[:512].contiguous()
wp512 = ops.split_weight_bf16x3(w512)
p512 = ops.gemm_bf16x3(a, wp512)
cmp('N=512 plain vs N=640 plain', p512, plain[:, :512])
t512 = table[:, :512].contiguous()
x0, x1 = ops.gemm_bf16x3_ex(a, wp512, None, t512, residual_rows=S, n_split=256)
cmp('N=512 table + split: out', x0, p512[:, :256] + t512.repeat(F, 1)[:, :256])
cmp('N=512 table + split: out2', x1, p512[:, 256:] + t512.repeat(F, 1)[:, 256:])
import torch.nn.functional as Fn
ref64 = (a.double() @ w.double().t()).float()
cmp('plain vs fp64 (tol 1e-4)', plain, ref64)
