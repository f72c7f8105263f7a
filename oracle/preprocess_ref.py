"""oracle/preprocess_ref.py -- TEST INFRASTRUCTURE.

NumPy restatement of the reference's test-time input pipeline for one clip
(configs/_base_/datasets/posetrack17_video_keypoint.py:71-84): mmcv.imrescale (rescale_size +
cv2.resize INTER_LINEAR on float32 images), mmcv.imnormalize (BGR->RGB, (x-mean)*(1/std)),
mmcv.impad_to_multiple (zeros, bottom/right), frames stacked CHW.

PARITY UNPINNED for the resize: cv2 is not installed in the build image and the reference holds
no fixture for it (mmcv/image/geometric.py:63-107 just calls cv2.resize), so the bilinear kernel
below restates OpenCV's float INTER_LINEAR in its published arithmetic order and could not be
checked against cv2 itself:
    inv_scale = dst / src (double);  scale = 1 / inv_scale (double)
    f = float((dx + 0.5) * scale - 0.5)  computed in double, then cast;  s = floor(f);  f -= s
    s < 0 -> s = 0, f = 0;   s >= src - 1 -> s = src - 1, f = 0
    horizontal pass: row[dx] = S[s] * (1 - f) + S[s + 1] * f          (float, one rounding per op)
    vertical pass:   D = row0 * (1 - fy) + row1 * fy                  (float, one rounding per op)
rescale_size / normalise / pad are plain arithmetic (pinned: tests/golden/pipeline_shapes.json).
"""
import numpy as np


def rescale_size(old_size, scale):
    w, h = old_size
    f = min(max(scale) / max(h, w), min(scale) / min(h, w))
    return int(w * float(f) + 0.5), int(h * float(f) + 0.5)


def resize_linear(img, new_wh):
    H0, W0 = img.shape[:2]
    Wn, Hn = new_wh
    img = img.astype(np.float32)

    def coords(n_dst, n_src):
        scale = 1.0 / (float(n_dst) / float(n_src))                       # double, as OpenCV
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        i0 = np.floor(f).astype(np.int64)
        frac = (f - i0.astype(np.float32)).astype(np.float32)
        frac[i0 < 0] = 0
        i0[i0 < 0] = 0
        frac[i0 >= n_src - 1] = 0
        i0[i0 >= n_src - 1] = n_src - 1
        return i0, np.minimum(i0 + 1, n_src - 1), frac

    x0, x1, fx = coords(Wn, W0)
    y0, y1, fy = coords(Hn, H0)
    fx = fx[None, :, None]
    fy = fy[:, None, None]
    one = np.float32(1)
    # horizontal pass on the two source rows of every output row, then the vertical pass; every
    # product / sum is its own float32 operation (numpy never fuses them)
    top = img[y0][:, x0] * (one - fx) + img[y0][:, x1] * fx
    bot = img[y1][:, x0] * (one - fx) + img[y1][:, x1] * fx
    return (top * (one - fy) + bot * fy).astype(np.float32)


def preprocess_clip(frames, img_scale=(1333, 800), size_divisor=1,
                    mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375), to_rgb=True):
    """frames [T, H0, W0, 3] BGR -> (img [1, T, 3, Hp, Wp] float32, meta)."""
    T, H0, W0, _ = frames.shape
    Wn, Hn = rescale_size((W0, H0), img_scale)
    d = max(int(size_divisor), 1)
    Hp, Wp = -(-Hn // d) * d, -(-Wn // d) * d
    out = np.zeros((1, T, 3, Hp, Wp), np.float32)
    mean = np.asarray(mean, np.float32)
    stdinv = (1.0 / np.asarray(std, np.float64)).astype(np.float32)
    for t in range(T):
        r = resize_linear(frames[t], (Wn, Hn))
        if to_rgb:
            r = r[..., ::-1]
        r = (r - mean) * stdinv
        out[0, t, :, :Hn, :Wn] = r.transpose(2, 0, 1)
    meta = dict(ori_shape=(H0, W0, 3), img_shape=(Hn, Wn, 3), pad_shape=(Hp, Wp, 3),
                batch_input_shape=(Hp, Wp), scale_factor=(Wn / W0, Hn / H0, Wn / W0, Hn / H0))
    return out, meta
