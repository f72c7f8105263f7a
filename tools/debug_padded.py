"""Where a padded batch's values come from: the same two-clip padded batch (valid sizes 800x1333 and 750x1333 in an
800x1344 batch) through the fast path and the masked_fill formulation (deform_attn.PADDED_FAST_PATH), as a batch and
clip by clip -- encoder memory and final key points compared pairwise.   python tools/debug_padded.py [T=3]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pavenet_amd  # noqa: E402,F401
import torch  # noqa: E402
from pavenet_amd import bricks, deform_attn  # noqa: E402
from pavenet_amd.models import build_model, videopose_r50_cfg  # noqa: E402
from pavenet_amd.weights import init_random_weights  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 3
H, W = 800, 1344
m = init_random_weights(build_model(videopose_r50_cfg(num_frames=T, max_per_img=20)), seed=0).cuda().eval()
bricks.set_gemm_mode('bf16x3')
g = torch.Generator(device='cuda').manual_seed(5)
img = torch.randn(2, T, 3, H, W, device='cuda', generator=g)
if len(sys.argv) > 2:      # the padded full-size test's inputs: bench.py's clip 0 + seed 1234
    import bench

    class A:
        height, width = H, W
    img = torch.randn(2, T, 3, H, W, device='cuda', generator=torch.Generator(device='cuda').manual_seed(1234))
    img[0].copy_(bench.clip0_image(A, T)[0])
shapes = [(800, 1333, 3), (750, 1333, 3)]
metas = [dict(batch_input_shape=(H, W), img_shape=s, scale_factor=(1., 1., 1., 1.)) for s in shapes]


def run(imgs, mt, fast, sel=None):
    deform_attn.PADDED_FAST_PATH = fast
    with torch.no_grad():
        feat = m.extract_feat(imgs)
        kw = {} if sel is None else dict(force_topk_proposals=sel[0])
        outs = m.bbox_head(feat, mt, last_level_only=True, **kw)
        res = m.bbox_head.get_bboxes(outs, mt, force_score_topk=None if sel is None else sel[1])
        mem = outs['memory']
        tp = m.bbox_head.transformer.last_topk_proposals.clone()
    return mem.clone(), res['kpts'].clone(), tp, res['score_index'].clone()


mem_f, k_f, tp, si = run(img, metas, True)
sel = (tp, si)
mem_f, k_f, _, _ = run(img, metas, True, sel)
mem_s, k_s, _, _ = run(img, metas, False, sel)
print('batch fast vs slow: memory max abs diff', float((mem_f - mem_s).abs().max()), ' kpts', float((k_f - k_s).abs().max()))
for c in range(2):
    selc = (tp[c:c + 1], si[c:c + 1])
    mem1f, k1f, _, _ = run(img[c:c + 1], metas[c:c + 1], True, selc)
    mem1s, k1s, _, _ = run(img[c:c + 1], metas[c:c + 1], False, selc)
    # memory layout [S, B*T, C] (sequence first): clip c's frames are columns c*T .. c*T+T-1
    mb_f = mem_f[:, c * T:(c + 1) * T] if mem_f.shape[1] == 2 * T else mem_f[c * T:(c + 1) * T]
    mb_s = mem_s[:, c * T:(c + 1) * T] if mem_s.shape[1] == 2 * T else mem_s[c * T:(c + 1) * T]
    print(f'clip {c}: alone fast vs alone slow: memory {float((mem1f - mem1s).abs().max()):.3e} kpts {float((k1f - k1s).abs().max()):.3e}')
    print(f'clip {c}: in batch vs alone (fast): memory {float((mb_f - mem1f).abs().max()):.3e} kpts {float((k_f[c] - k1f[0]).abs().max()):.3e}')
    print(f'clip {c}: in batch vs alone (slow): memory {float((mb_s - mem1s).abs().max()):.3e} kpts {float((k_s[c] - k1s[0]).abs().max()):.3e}')

# the padded full-size test's construction: clip 0's selections from its ALONE run pinned inside the batch
_, k1, tp1, si1 = run(img[:1], metas[:1], True)
_, k1p, _, _ = run(img[:1], metas[:1], True, (tp1, si1))
tpb, sib = tp.clone(), si.clone()
tpb[0].copy_(tp1[0])
sib[0].copy_(si1[0])
_, kb, _, _ = run(img, metas, True, (tpb, sib))
print('clip 0 alone free vs alone pinned-to-itself:', float((k1 - k1p).abs().max()))
print('clip 0: batch with the ALONE run\'s selections vs alone:', float((kb[0] - k1p[0]).abs().max()),
      ' same proposal set as the batch free run:', bool(torch.equal(tp1[0].sort()[0], tp[0].sort()[0])),
      ' same order:', bool(torch.equal(tp1[0], tp[0])), ' score picks equal:', bool(torch.equal(si1[0], si[0])))
