"""The decoders' small Linears under the kernel trace (a dependent chain through the Python wrappers is
CPU-bound at ~12 us per launch, so wall time cannot tell two kernels apart):
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/sg -- python3 $GRAFT_REPO_ROOT/tools/small_gemm_trace.py [other.so]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pavenet_amd import native, ops  # noqa: E402

if len(sys.argv) > 1:
    native._lib = native._open(sys.argv[1])
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
mk = lambda n, k: ops.split_weight_bf16x3(torch.randn(n, k, device=dev, generator=g) * 0.05)   # noqa: E731
xs, x4, x5 = (torch.randn(1200, k, device=dev, generator=g) for k in (256, 1024, 512))
res = torch.randn(1200, 256, device=dev, generator=g)
b, gam, bet = (torch.randn(256, device=dev, generator=g) for _ in range(3))
w256 = [mk(256, 256) for _ in range(20)]
w1024 = [mk(256, 1024) for _ in range(20)]
w512 = [mk(512, 512) for _ in range(20)]
w32 = [mk(64, 512) for _ in range(20)]
for rep in range(3):
    for w in w256:
        ops.gemm_bf16x3(xs, w, b)
    for w in w1024:
        ops.gemm_bf16x3(x4, w, b)
    for w in w512:
        ops.gemm_bf16x3(x5, w, None, relu=True)
    for w in w32:
        ops.gemm_bf16x3(x5, w, None, n_out=32)
    for w in w256:
        ops.gemm_bf16x3_ln(xs, w, b, res, gam, bet, 1e-5)
    for w in w1024:
        ops.gemm_bf16x3_ln(x4, w, b, res, gam, bet, 1e-5)
torch.cuda.synchronize()
