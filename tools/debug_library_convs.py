"""Which convolutions of a model still go to the library (F.conv2d) in the headline GEMM mode: one
forward of the bench batch with F.conv2d wrapped, shapes and call counts printed.
    python tools/debug_library_convs.py [r50|hrnet_w48]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pavenet_amd  # noqa: E402,F401
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from pavenet_amd import tuning  # noqa: E402
from pavenet_amd.bricks import set_gemm_mode  # noqa: E402
from pavenet_amd.models import build_model, videopose_r50_cfg, with_hrnet_w48  # noqa: E402
from pavenet_amd.weights import init_random_weights  # noqa: E402


def main():
    T, B, H, W = 7, 4, 800, 1344
    cfg = videopose_r50_cfg(num_frames=T, max_per_img=20)
    if len(sys.argv) > 1 and sys.argv[1] == 'hrnet_w48':
        cfg = with_hrnet_w48(cfg)
    m = build_model(cfg)
    init_random_weights(m, seed=0)
    m = m.cuda().eval()
    set_gemm_mode('bf16x3')
    tuning.use_tuned_gemms()
    img = torch.randn(B, T, 3, H, W, device='cuda')
    metas = [dict(batch_input_shape=(H, W), img_shape=(H, W, 3), scale_factor=(1., 1., 1., 1.))] * B
    m.forward_device(img, metas)
    seen = collections.Counter()
    orig = F.conv2d

    def conv2d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        seen[(tuple(x.shape), tuple(w.shape), str(stride), str(padding), groups,
              'cl' if x.is_contiguous(memory_format=torch.channels_last) else 'nchw')] += 1
        return orig(x, w, b, stride, padding, dilation, groups)
    F.conv2d = conv2d
    torch.nn.functional.conv2d = conv2d
    m.forward_device(img, metas)
    torch.cuda.synchronize()
    for k, n in sorted(seen.items(), key=lambda kv: -kv[1]):
        print(n, k)
    print('library convolutions per forward:', sum(seen.values()))


if __name__ == '__main__':
    main()
