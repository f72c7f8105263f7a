"""XCD / L2-aware processing order of encoder tokens (host logic, cached per shape).

MI355X deals workgroups round-robin over its 8 XCDs, each with a private 4 MiB L2, and one
frame's projected value maps are 22.9 MB at 800x1344 -- far more than one L2.  The fused
kernel (pave_kernels.hip: xcd_remap) hands every XCD one contiguous run of logical blocks;
this module orders the units so that such a run is one horizontal image band across all
levels (a query samples around its own position on every level, so a band of queries reads
a band of every value map: ~1/8 of 22.9 MB, which fits the XCD's L2), and so that all XCDs
work on the same frame at the same time.
"""
import numpy as np
import torch

N_XCD = 8


def encoder_unit_order(levels, n_frames, mode='band'):
    """Permutation [n_frames * S] int32: position -> unit (= frame * S + token).

    levels: [(H, W), ...] in flattening order.  mode 'band': tokens of a frame sorted by
    normalised row centre (ties: level, column), cut into 8 equal bands; units laid out
    band-major, then frame, then in-band order.  mode 'patch': as 'band' but 8 x 4 pixel patches
    consecutive inside a band.  mode 'none': identity.
    """
    S = sum(int(h) * int(w) for h, w in levels)
    if mode == 'none':
        return torch.arange(n_frames * S, dtype=torch.int32)
    ys, lv, xs = [], [], []
    for l, (h, w) in enumerate(levels):
        h, w = int(h), int(w)
        r = np.repeat(np.arange(h), w)
        ys.append((r + 0.5) / h)
        xs.append(np.tile(np.arange(w), h))
        lv.append(np.full(h * w, l))
    ys, lv, xs = np.concatenate(ys), np.concatenate(lv), np.concatenate(xs)
    if mode == 'quad':
        # 2x2 pixel patches are consecutive, so the 4 waves of a workgroup gather overlapping
        # corner rows (a 2x2 patch with equal offsets touches 3x3 rows instead of 4x4)
        rows = np.concatenate([np.repeat(np.arange(int(h)), int(w)) for h, w in levels])
        py, qy = rows // 2, rows % 2
        pxx, qx = xs // 2, xs % 2
        ysq = np.concatenate([((np.repeat(np.arange(int(h)), int(w)) // 2) * 2 + 1.0) / int(h)
                              for h, w in levels])
        tok = np.lexsort((qx, qy, pxx, lv, ysq))
    elif mode == 'patch':
        # 8 x 4 pixel patches are consecutive: the 32 queries of a workgroup of the head-major
        # kernel (one head, 32 units) then share one small neighbourhood of value rows
        rows = np.concatenate([np.repeat(np.arange(int(h)), int(w)) for h, w in levels])
        hs = np.concatenate([np.full(int(h) * int(w), int(h)) for h, w in levels])
        ysp = ((rows // 4) * 4 + 2.0) / hs                    # patch-row centre
        tok = np.lexsort((xs % 8, rows % 4, xs // 8, lv, ysp))
    else:
        tok = np.lexsort((xs, lv, ys))  # primary key: y centre
    bounds = [(S * i) // N_XCD for i in range(N_XCD + 1)]
    out = []
    for b in range(N_XCD):
        band = tok[bounds[b]:bounds[b + 1]]
        for f in range(n_frames):
            out.append(f * S + band)
    order = np.concatenate(out).astype(np.int32)
    assert order.shape[0] == n_frames * S
    return torch.from_numpy(order)
