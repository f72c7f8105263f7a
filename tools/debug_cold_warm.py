"""Is the forward the same from the FIRST call of a process on, and does it read uninitialised memory?
Runs the bench batch four times on one model: run 1 (every cache cold) against run 2, then run 3 / 4
after the allocator's free memory was filled with NaN / 1e30.  Everything must be bit-identical.
    python tools/debug_cold_warm.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pavenet_amd, torch
from pavenet_amd import bricks, tuning
from pavenet_amd.models import build_model, videopose_r50_cfg
from pavenet_amd.weights import init_random_weights
T, B, N, H, W = 7, 4, 20, 800, 1344
m = build_model(videopose_r50_cfg(num_frames=T, max_per_img=N)); init_random_weights(m, seed=0)
m = m.cuda().eval()
g = torch.Generator(device='cuda').manual_seed(1234)
img = torch.randn(B, T, 3, H, W, device='cuda', generator=g)
metas = [dict(batch_input_shape=(H, W), img_shape=(H, W, 3), scale_factor=(1., 1., 1., 1.)) for _ in range(B)]
bricks.set_gemm_mode('bf16x3'); tuning.use_tuned_gemms()
def fwd():
    with torch.no_grad():
        feats = [t.clone() for t in m.neck(m.backbone(img))]
        res = m.forward_device(img, metas)
        return feats, {k: v.clone() for k, v in res.items() if torch.is_tensor(v)}
def poison(val):
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    blocks = []
    # fill the caching allocator's free blocks AND fresh memory with a pattern, then release to the pool
    for sz in (8 << 30, 4 << 30, 2 << 30, 1 << 30, 1 << 30, 512 << 20, 512 << 20, 256 << 20, 128 << 20, 64 << 20, 32 << 20, 16 << 20, 8 << 20, 4 << 20, 2 << 20, 1 << 20):
        try:
            blocks.append(torch.full((sz // 4,), val, dtype=torch.float32, device='cuda'))
        except RuntimeError:
            pass
    torch.cuda.synchronize()
    del blocks
def cmp(a, b, name):
    fa, ra = a; fb, rb = b
    print(name, 'neck feats equal', [bool(torch.equal(p, q)) for p, q in zip(fa, fb)],
          {k: (bool(torch.equal(ra[k], rb[k])), bool(torch.isnan(rb[k].float()).any())) for k in ra})
r1 = fwd(); r2 = fwd(); cmp(r1, r2, 'run1 vs run2')
poison(float('nan')); r3 = fwd(); cmp(r2, r3, 'run2 vs run3 (free memory poisoned with NaN)')
poison(1e30); r4 = fwd(); cmp(r2, r4, 'run2 vs run4 (free memory poisoned with 1e30)')
