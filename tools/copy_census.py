"""Where do the small layout copies / adds / clamps of a step come from?  torch.profiler with
Python stacks over one bench step; prints the product source lines that issue the most
aten::copy_ / aten::add / aten::clamp_min device launches.   python tools/copy_census.py"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pavenet_amd  # noqa: E402,F401
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402
from pavenet_amd.bricks import set_gemm_mode  # noqa: E402
from pavenet_amd.models import build_model, videopose_r50_cfg  # noqa: E402
from pavenet_amd.weights import init_random_weights  # noqa: E402

T, B = 7, 4
m = init_random_weights(build_model(videopose_r50_cfg(num_frames=T, max_per_img=20)), seed=0).cuda().eval()
set_gemm_mode('bf16x3')
img = torch.randn(B, T, 3, 800, 1344, device='cuda')
metas = [dict(batch_input_shape=(800, 1344), img_shape=(800, 1344, 3), scale_factor=(1., 1., 1., 1.))] * B
with torch.no_grad():
    for _ in range(3):
        m.forward_device(img, metas)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
        m.forward_device(img, metas)
        torch.cuda.synchronize()
want = ('aten::copy_', 'aten::add', 'aten::add_', 'aten::clamp_min', 'aten::relu', 'aten::clamp_min_',
        'aten::cat', 'aten::mul', 'aten::sigmoid', 'aten::contiguous', 'aten::clone')
cnt = collections.Counter()
for ka in prof.key_averages(group_by_stack_n=12):
    if ka.key in want:
        src = next((f for f in ka.stack if 'pavenet_amd' in f), None)
        if src is None:
            src = next((f for f in ka.stack if 'torch/nn/' in f), '?')
        cnt[(ka.key, src.strip().split('pavenet_amd/')[-1][:100])] += ka.count
if not cnt:
    print('no stacks recorded; totals:', {ka.key: ka.count for ka in prof.key_averages() if ka.key in want})
for (name, src), n in cnt.most_common(45):
    print(f'{n:4d}  {name:18s} {src}')
