"""Stability soak: N forward steps of the bench workload; reports the allocator's peak / current
bytes at several points (a leak or a growing pool shows as a drift) and that the result stays put.
python tools/soak.py [steps=300]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pavenet_amd  # noqa: E402,F401
import torch  # noqa: E402
from pavenet_amd.bricks import set_gemm_mode  # noqa: E402
from pavenet_amd.models import build_model, videopose_r50_cfg  # noqa: E402
from pavenet_amd.weights import init_random_weights  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
T, B = 7, 4
m = init_random_weights(build_model(videopose_r50_cfg(num_frames=T, max_per_img=20)), seed=0).cuda().eval()
set_gemm_mode('bf16x3')
img = torch.randn(B, T, 3, 800, 1344, device='cuda', generator=torch.Generator(device='cuda').manual_seed(1))
metas = [dict(batch_input_shape=(800, 1344), img_shape=(800, 1344, 3), scale_factor=(1., 1., 1., 1.))] * B
first = None
with torch.no_grad():
    for it in range(steps):
        res = m.forward_device(img, metas)
        k = res['kpts'].float().cpu()
        first = k if first is None else first
        if it in (5, steps // 2, steps - 1):
            torch.cuda.synchronize()
            print(f'step {it}: allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB, reserved '
                  f'{torch.cuda.memory_reserved() / 2**30:.2f} GiB, peak {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB, '
                  f'max|kpts - first| = {float((k - first).abs().max()):.2e}', flush=True)
print('soak ok')
