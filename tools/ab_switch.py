"""A/B of one model switch on the bench workload inside ONE process (boxes of the pool differ by
~2 %): alternates the two settings, 3 rounds of N steps each.
python tools/ab_switch.py <attr path under model, e.g. bbox_head.transformer.overlap_value_proj> [steps=10] [value A] [value B]
(values default to True / False; given, they are strings: backbone.chain_mode 10 tail full)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pavenet_amd  # noqa: E402,F401
import torch  # noqa: E402
from pavenet_amd.bricks import set_gemm_mode  # noqa: E402
from pavenet_amd.models import build_model, videopose_r50_cfg  # noqa: E402
from pavenet_amd.weights import init_random_weights  # noqa: E402

path = sys.argv[1].split('.')
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
T, B = int(os.environ.get('AB_T', 7)), int(os.environ.get('AB_B', 4))     # AB_T=3 AB_B=1: configs[1]
m = init_random_weights(build_model(videopose_r50_cfg(num_frames=T, max_per_img=20)), seed=0).cuda().eval()
set_gemm_mode('bf16x3')
img = torch.randn(B, T, 3, 800, 1344, device='cuda', generator=torch.Generator(device='cuda').manual_seed(1))
metas = [dict(batch_input_shape=(800, 1344), img_shape=(800, 1344, 3), scale_factor=(1., 1., 1., 1.))] * B
obj = m
for p in path[:-1]:
    obj = getattr(obj, p)


def run(n):
    with torch.no_grad():
        for _ in range(n):
            r = m.forward_device(img, metas)
            r['kpts'].cpu()


VALUES = (sys.argv[3], sys.argv[4]) if len(sys.argv) > 4 else (True, False)
for v in VALUES:
    setattr(obj, path[-1], v)
    run(3)
res = {v: [] for v in VALUES}
for rnd in range(3):
    for v in VALUES:
        setattr(obj, path[-1], v)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / steps * 1e3)
for v in VALUES:
    print(f'{sys.argv[1]} = {v}: ' + ', '.join(f'{t:.2f}' for t in res[v]) + ' ms/step')
