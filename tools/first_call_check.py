"""First forward of a fresh model against the second and third, six workloads (one-clip T = 3, fp16 T = 15, 8 clips,
Swin-L, HRNet-w48, fp16 3 clips): every difference must be 0.   python tools/first_call_check.py"""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pavenet_amd
from pavenet_amd.bricks import set_gemm_mode
from pavenet_amd.models import build_model, videopose_r50_cfg, with_hrnet_w48, with_swin_l
from pavenet_amd.weights import init_random_weights
for name, T, B, mode, bb in (('r50 T3 B1', 3, 1, 'bf16x3', None), ('r50 T15 B1 fp16', 15, 1, 'fp16', None), ('r50 T7 B8', 7, 8, 'bf16x3', None),
                             ('swin T3', 3, 1, 'bf16x3', 'swin'), ('hrnet T7 B1', 7, 1, 'bf16x3', 'hrnet'), ('r50 T7 B3 fp16', 7, 3, 'fp16', None)):
    cfg = videopose_r50_cfg(num_frames=T, max_per_img=20)
    if bb == 'swin': cfg = with_swin_l(cfg, num_frames=T)
    if bb == 'hrnet': cfg = with_hrnet_w48(cfg)
    m = init_random_weights(build_model(cfg), seed=0).cuda().eval()
    set_gemm_mode(mode)
    img = torch.randn(B, T, 3, 800, 1344, device='cuda', generator=torch.Generator(device='cuda').manual_seed(1))
    metas = [dict(batch_input_shape=(800, 1344), img_shape=(800, 1344, 3), scale_factor=(1., 1., 1., 1.))] * B
    outs = []
    with torch.no_grad():
        for _ in range(3):
            outs.append(m.forward_device(img, metas)['kpts'].float().cpu())
    print(name, 'first vs second', float((outs[0] - outs[1]).abs().max()), 'second vs third', float((outs[1] - outs[2]).abs().max()), flush=True)
    del m, img
    torch.cuda.empty_cache()
