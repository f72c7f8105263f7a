"""Per-shape census of the split-GEMM / convolution launches of one bench step (T = 7 x 4 clips,
800x1344, --gemm bf16x3): HIP-event time, TFLOP/s and launch count per (entry point, M, K, N, form),
in launch order.   python tools/gemm_census.py [steps=3] [diag variant (0 = none)] [T=7] [clips=4] [r50 | hrnet_w48 | swin_l] [bf16x3 | fp16]
(the last column: algorithmic TB/s of the launch -- A + W + out (+ residual) once, fp32 unless the mode stores fp16)"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pavenet_amd  # noqa: E402,F401
import torch  # noqa: E402
from pavenet_amd import ops, tuning  # noqa: E402
from pavenet_amd.bricks import set_gemm_mode  # noqa: E402
from pavenet_amd.models import build_model, videopose_r50_cfg  # noqa: E402
from pavenet_amd.weights import init_random_weights  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    if len(sys.argv) > 2 and int(sys.argv[2]):   # kernel-form override (pave_diag_gemm_variant), e.g. 8 = no wide tiles
        from pavenet_amd import native
        native.use_diag_build(int(sys.argv[2]))
    T, B, H, W = 7, 4, 800, 1344
    if len(sys.argv) > 4:
        T, B = int(sys.argv[3]), int(sys.argv[4])
    cfg = videopose_r50_cfg(num_frames=T, max_per_img=20)
    if len(sys.argv) > 5 and sys.argv[5] == 'hrnet_w48':
        from pavenet_amd.models import with_hrnet_w48
        cfg = with_hrnet_w48(cfg)
    elif len(sys.argv) > 5 and sys.argv[5] == 'swin_l':
        from pavenet_amd.models import with_swin_l
        cfg = with_swin_l(cfg, num_frames=T)
    m = build_model(cfg)
    init_random_weights(m, seed=0)
    m = m.cuda().eval()
    set_gemm_mode(sys.argv[6] if len(sys.argv) > 6 else 'bf16x3')
    tuning.use_tuned_gemms()
    img = torch.randn(B, T, 3, H, W, device='cuda')
    metas = [dict(batch_input_shape=(H, W), img_shape=(H, W, 3), scale_factor=(1., 1., 1., 1.))] * B
    for _ in range(2):
        m.forward_device(img, metas)
    torch.cuda.synchronize()
    ops.KERNEL_EVENT_TAGS = ('gemm_bf16x3', 'gemm_bf16x3_ln', 'gemm_bf16x3_small', 'gemm_bf16x3_ln_small', 'conv3x3_split', 'conv1x1_strided',
                             'conv7x7_stem', 'bottleneck_chain', 'enc_tile', 'rows_gemm', 'stem_pool', 'groupnorm')
    ops.KERNEL_EVENTS, ops.KERNEL_EVENT_SHAPES = [], []
    for _ in range(steps):
        m.forward_device(img, metas)
    torch.cuda.synchronize()
    ev, shapes = ops.KERNEL_EVENTS, ops.KERNEL_EVENT_SHAPES
    ops.KERNEL_EVENTS = ops.KERNEL_EVENT_SHAPES = None
    agg = collections.OrderedDict()
    for (tag, s, e, fl), shp in zip(ev, shapes):
        d = agg.setdefault((tag, shp), [0, 0.0, fl])
        d[0] += 1
        d[1] += s.elapsed_time(e)
    tot = 0.0
    print(f'{"entry point":18s} {"shape":44s} {"n/step":>6s} {"ms each":>8s} {"ms/step":>8s} {"TF/s":>6s} {"TB/s":>6s}')
    for (tag, shp), (n, ms, fl) in agg.items():
        each = ms / n
        tot += ms / steps
        tf = f'{fl / each / 1e9:6.0f}' if fl else '     -'
        tb = '     -'
        if shp and len(shp) >= 3 and all(isinstance(v, int) for v in shp[:3]) and fl:
            M_, K_, N_ = shp[:3]
            kk = K_ // 9 if any('3x3' in str(v) for v in shp) else K_     # (3x3: the map is read once, not 9 times)
            byt = 4 * (M_ * kk + M_ * N_ * (2 if 'res' in shp else 1)) + 6 * K_ * N_
            tb = f'{byt / each / 1e9:6.2f}'
        print(f'{tag:18s} {str(shp):44s} {n / steps:6.1f} {each:8.3f} {ms / steps:8.3f} {tf} {tb}')
    print(f'total of the listed launches: {tot:.2f} ms/step')


if __name__ == '__main__':
    main()
