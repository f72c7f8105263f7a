"""Where do the small layout copies / adds / clamps of a step come from?  torch.profiler with
Python stacks over one bench step; prints the product source lines that issue the most
aten::copy_ / aten::add / aten::clamp_min device launches.   python tools/copy_census.py"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pavenet_amd  # noqa: E402,F401
import torch  # noqa: E402
from pavenet_amd.bricks import set_gemm_mode  # noqa: E402
from pavenet_amd.models import build_model, videopose_r50_cfg  # noqa: E402
from pavenet_amd.weights import init_random_weights  # noqa: E402

T, B = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (7, 4)
m = init_random_weights(build_model(videopose_r50_cfg(num_frames=T, max_per_img=20)), seed=0).cuda().eval()
set_gemm_mode('bf16x3')
img = torch.randn(B, T, 3, 800, 1344, device='cuda')
metas = [dict(batch_input_shape=(800, 1344), img_shape=(800, 1344, 3), scale_factor=(1., 1., 1., 1.))] * B
import traceback  # noqa: E402

from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

LIB = ('addmm', 'mm', 'bmm', 'baddbmm', 'linear', 'matmul', 'convolution', 'cudnn_convolution', 'miopen_convolution',
       '_scaled_dot_product_flash_attention', '_scaled_dot_product_efficient_attention', 'native_layer_norm',
       'native_group_norm', '_softmax', 'topk', 'sort')      # `python tools/copy_census.py lib T B`: library kernels
WANT = ('copy_', 'clone', 'add', 'add_', 'relu', 'clamp_min', 'clamp_min_', 'cat', 'mul', 'sigmoid',
        '_to_copy', 'index', 'gather', 'expand_copy', 'repeat', 'stack')
# `python tools/copy_census.py all`: every aten op that launches something (all but the view / metadata ops)
ALL = len(sys.argv) > 1 and sys.argv[1] == 'all'
LIBONLY = len(sys.argv) > 1 and sys.argv[1] == 'lib'
VIEWS = ('view', 'reshape', '_unsafe_view', 'permute', 'expand', 'slice', 'select', 'as_strided', 'unsqueeze',
         'squeeze', 'transpose', 't', 'detach', 'alias', 'empty', 'empty_like', 'empty_strided', 'unflatten', 'flatten',
         'split', 'split_with_sizes', 'unbind', 'chunk', 'narrow', 'view_as', '_reshape_alias', 'lift_fresh', 'unfold',
         'diagonal', 'movedim', 'new_empty', 'new_empty_strided', 'is_same_size', 'sym_size', 'sym_stride', 'stride',
         'size', 'dim', 'numel', 'is_contiguous', 'storage_offset', 'new_zeros_like', '_local_scalar_dense')
cnt = collections.Counter()


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split('.')[0]
        if (ALL and name not in VIEWS) or (LIBONLY and name in LIB) or (not LIBONLY and name in WANT):
            fr = [f for f in traceback.extract_stack() if 'pavenet_amd/' in f.filename]
            if fr:
                f = fr[-1]
                cnt[(name, f'{os.path.basename(f.filename)}:{f.lineno} {f.line[:70]}')] += 1
        return func(*args, **(kwargs or {}))


with torch.no_grad():
    for _ in range(3):
        m.forward_device(img, metas)
    torch.cuda.synchronize()
    with Census():
        m.forward_device(img, metas)
    torch.cuda.synchronize()
for (name, src), n in cnt.most_common(60):
    print(f'{n:4d}  {name:12s} {src}')
