"""A/B in one process, interleaved: BASELINE configs[4] on one GPU (R-50, T = 15, one clip, 800x1344) in the fp16
operand mode with the shipped tile selection (wide tiles where they apply) against narrow tiles only (diag variant 8;
round 5 measured: 21.8 against 22.1 ms with fp16 activations, 23.1 against 24.3 without -- also with the fp16
narrow kernels bounded to 156 registers for three blocks per CU; not kept).   python tools/fp16_form_ab.py [steps=10]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pavenet_amd  # noqa: E402,F401
import torch  # noqa: E402
from pavenet_amd import bricks, native  # noqa: E402
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


CASES = (('shipped selection, fp16 activations', 0, True), ('narrow tiles only, fp16 activations', 8, True),
         ('shipped selection, fp32 activations', 0, False), ('narrow tiles only, fp32 activations', 8, False))
for _, v, act in CASES:
    bricks.FP16_ACTIVATIONS = act
    with native.diag_build(v):
        run(2)
res = {c[0]: [] for c in CASES}
for rnd in range(3):
    for name, v, act in CASES:
        bricks.FP16_ACTIVATIONS = act
        with native.diag_build(v):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(steps)
            torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) / steps * 1e3)
for name, _, _ in CASES:
    print(f'{name}: ' + ', '.join(f'{t:.2f}' for t in res[name]) + ' ms/step')
