"""A/B in one process, interleaved: BASELINE configs[4] on one GPU (R-50, T = 15, one clip, 800x1344) in the fp16
operand mode with the FFN hidden stored as fp16 between its two launches (bricks.FP16_ACTIVATIONS) and as fp32.
python tools/fp16_act_ab.py [steps=10]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pavenet_amd  # noqa: E402,F401
import torch  # noqa: E402
from pavenet_amd import bricks  # noqa: E402
from pavenet_amd.models import build_model, videopose_r50_cfg  # noqa: E402
from pavenet_amd.weights import init_random_weights  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
T = 15
m = init_random_weights(build_model(videopose_r50_cfg(num_frames=T, max_per_img=20)), seed=0).cuda().eval()
bricks.set_gemm_mode('fp16')
img = torch.randn(1, T, 3, 800, 1344, device='cuda', generator=torch.Generator(device='cuda').manual_seed(1))
metas = [dict(batch_input_shape=(800, 1344), img_shape=(800, 1344, 3), scale_factor=(1., 1., 1., 1.))]


def run(n):
    with torch.no_grad():
        for _ in range(n):
            r = m.forward_device(img, metas)
            r['kpts'].cpu()
    return r['kpts'].clone()


out = {}
for v in (True, False):
    bricks.FP16_ACTIVATIONS = v
    out[v] = run(3)
print('same key points with fp16 and fp32 hidden activations:', bool(torch.equal(out[True], out[False])))
res = {True: [], False: []}
for rnd in range(3):
    for v in (True, False):
        bricks.FP16_ACTIVATIONS = v
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / steps * 1e3)
for v in (True, False):
    print(f'FFN hidden as {"fp16" if v else "fp32"}: ' + ', '.join(f'{t:.2f}' for t in res[v]) + ' ms/step')
