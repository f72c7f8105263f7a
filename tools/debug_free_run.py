"""Free run (nothing pinned) of clip 0 inside the bench batch against clip 0 alone: do the two top-k selections
agree (set and order), where do the logits sit at the selection boundaries, how far are the final poses apart.
    python tools/debug_free_run.py [T=7] [clips=4]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pavenet_amd  # noqa: E402,F401
import torch  # noqa: E402
import bench  # noqa: E402
from pavenet_amd import bricks  # noqa: E402
from pavenet_amd.models import build_model, videopose_r50_cfg  # noqa: E402
from pavenet_amd.weights import init_random_weights  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 7
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
H, W = 800, 1344
m = init_random_weights(build_model(videopose_r50_cfg(num_frames=T, max_per_img=20)), seed=0).cuda().eval()
bricks.set_gemm_mode('bf16x3')


class A:
    height, width = H, W


img = torch.randn(B, T, 3, H, W, device='cuda', generator=torch.Generator(device='cuda').manual_seed(1234))
img[0].copy_(bench.clip0_image(A, T)[0])
metas = [dict(batch_input_shape=(H, W), img_shape=(H, W, 3), scale_factor=(1., 1., 1., 1.)) for _ in range(B)]
tr = m.bbox_head.transformer


def run(imgs, mt, **kw):
    with torch.no_grad():
        res = m.forward_device(imgs, mt, **kw)
    return (tr.last_topk_proposals.clone(), tr.last_enc_cls.clone(), res['score_index'].clone(), res['scores'].clone(),
            res['kpts'].clone(), res['keep'].clone())


pb, eb, sb, scb, kb, keepb = run(img, metas)
pa, ea, sa, sca, ka, keepa = run(img[:1], metas[:1])
print('proposal logits, batch vs alone: max |d|', float((eb[0] - ea[0]).abs().max()))
lg = ea[0, :, 0]
top = lg.topk(302)[0]
print('alone: proposal logit 299th / 300th / 301st / 302nd:', top[298:302].tolist(), ' distinct values in the top 300:',
      int(top[:300].unique().numel()))
print('proposals: same set', set(pb[0].tolist()) == set(pa[0].tolist()), ' same order', bool(torch.equal(pb[0], pa[0])),
      ' first position that differs', int((pb[0] != pa[0]).nonzero()[0]) if not torch.equal(pb[0], pa[0]) else None)
print('score picks: same set', set(sb[0].tolist()) == set(sa[0].tolist()), ' same order', bool(torch.equal(sb[0], sa[0])))
print('scores batch', [round(v, 6) for v in scb[0].tolist()])
print('scores alone', [round(v, 6) for v in sca[0].tolist()])
print('kept: batch', int(keepb[0].sum()), ' alone', int(keepa[0].sum()), ' final kpts max |d| (px)',
      float((kb[0, ..., :2] - ka[0, ..., :2]).abs().max()))
# with the alone run's selections pinned inside the batch
pb2 = pb.clone()
pb2[0].copy_(pa[0])
sb2 = sb.clone()
sb2[0].copy_(sa[0])
_, _, _, _, kb2, keepb2 = run(img, metas, force_topk_proposals=pb2, force_score_topk=sb2)
print('batch with the alone run\'s selections pinned: kept', int(keepb2[0].sum()), ' final kpts max |d| (px)',
      float((kb2[0, ..., :2] - ka[0, ..., :2]).abs().max()))
