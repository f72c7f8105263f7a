"""GPU input pipeline: raw decoded frames -> the network's input tensor
and ``img_metas``, on the device, in one HIP launch per clip.

Restates the reference's test pipeline
(configs/_base_/datasets/posetrack17_video_keypoint.py:71-84): Resize(img_scale=(1333, 800),
keep_ratio=True) [mmcv.imrescale -> rescale_size + cv2 INTER_LINEAR on float32 frames, since
the loader uses to_float32=True], RandomFlip(off), Normalize(mean, std, to_rgb=True),
Pad(size_divisor), MulImageToTensor / stacking of the T frames
(mmdet/datasets/pipelines/formatting.py:502-546).
"""
import ctypes

import torch

from . import native
from .ops import _require, _stream_ptr

MEAN = (123.675, 116.28, 103.53)
STD = (58.395, 57.12, 57.375)


def rescale_size(old_size, scale):
    """mmcv.image.rescale_size for a (long_edge, short_edge) tuple scale: old_size = (w, h)."""
    w, h = old_size
    max_long_edge, max_short_edge = max(scale), min(scale)
    f = min(max_long_edge / max(h, w), max_short_edge / min(h, w))
    return int(w * float(f) + 0.5), int(h * float(f) + 0.5)


def plan_clip(H0, W0, img_scale=(1333, 800), size_divisor=1):
    """Host arithmetic of Resize(keep_ratio=True) + Pad(size_divisor) for an H0 x W0 source:
    -> (Hn, Wn, Hp, Wp, scale_factor) exactly as mmcv.rescale_size, mmdet's Resize._resize_img
    (w_scale = new_w / w, h_scale = new_h / h) and mmcv.impad_to_multiple compute them."""
    Wn, Hn = rescale_size((W0, H0), img_scale)
    d = max(int(size_divisor), 1)
    Hp, Wp = -(-Hn // d) * d, -(-Wn // d) * d
    ws, hs = Wn / W0, Hn / H0
    return Hn, Wn, Hp, Wp, (ws, hs, ws, hs)


def preprocess_clip(frames, img_scale=(1333, 800), size_divisor=1, mean=MEAN, std=STD,
                    to_rgb=True):
    """frames [T, H0, W0, 3] uint8 / float32 BGR on the device -> (img [1, T, 3, Hp, Wp] fp32,
    img_meta dict with ori_shape / img_shape / pad_shape / batch_input_shape / scale_factor)."""
    lib = native.load()
    _require(frames.is_cuda and frames.dim() == 4 and frames.shape[-1] == 3 and
             frames.is_contiguous(), 'preprocess_clip: frames must be a contiguous device '
             '[T, H, W, 3] tensor')
    _require(frames.dtype in (torch.uint8, torch.float32), 'preprocess_clip: uint8 or float32')
    T, H0, W0, _ = frames.shape
    Hn, Wn, Hp, Wp, scale_factor = plan_clip(H0, W0, img_scale, size_divisor)
    out = torch.empty((1, T, 3, Hp, Wp), dtype=torch.float32, device=frames.device)
    m = (ctypes.c_float * 3)(*mean)
    s = (ctypes.c_float * 3)(*std)
    with torch.cuda.device(frames.device):
        st = lib.pave_preprocess_frames(frames.data_ptr(), int(frames.dtype == torch.uint8),
                                        out.data_ptr(), T, H0, W0, Hn, Wn, Hp, Wp,
                                        ctypes.cast(m, ctypes.c_void_p),
                                        ctypes.cast(s, ctypes.c_void_p), int(bool(to_rgb)),
                                        _stream_ptr())
    native.check(st, 'preprocess_frames')
    meta = dict(ori_shape=(H0, W0, 3), img_shape=(Hn, Wn, 3), pad_shape=(Hp, Wp, 3),
                batch_input_shape=(Hp, Wp), scale_factor=scale_factor, flip=False)
    return out, meta
